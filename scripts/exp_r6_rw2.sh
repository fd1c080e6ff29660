#!/bin/bash
# k_rescue_windows after a change: parity tests, then the driver's command twice
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/exp_r6_rw_tests.log 2>&1
tail -2 gpurun_out/exp_r6_rw_tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-pass --no-bam-pass > gpurun_out/exp_r6_rw_bench$i.json 2> gpurun_out/exp_r6_rw_bench$i.err
python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_rw_bench$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], d["parity_diffs"], d["records_sha1"][:8], "rescue_windows shared", r["kernel_ms_per_step"]["rescue_windows"], "alone", r["single_stream"]["kernel_ms_per_step"]["rescue_windows"])
P
done
