#!/bin/bash
# the last commit checked once more: smoke, every GPU test, the at-scale tests, the default bench line and the driver's command
TAG=${1:-r6last}
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py 2>&1 | tail -8 > gpurun_out/gputests_$TAG.log
bash scripts/gpu_scale_tests.sh $TAG 2400 > /dev/null 2>&1
python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_steps20_warmup5_$TAG.json 2> gpurun_out/bench_steps20_warmup5_$TAG.err
tail -2 gpurun_out/smoke_$TAG.log; cat gpurun_out/gputests_$TAG.log; tail -3 gpurun_out/scale_$TAG.log | cut -c1-200
for f in bench_default bench_steps20_warmup5; do python -c "
import json; d=json.load(open('gpurun_out/${f}_$TAG.json')); print('$f', d['value'], d['ms_per_step'], d['parity_diffs'], d['records_sha1'][:8], (d['config'].get('pcie_inclusive') or {}).get('reads_per_s'))"; done
