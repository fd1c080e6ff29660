#!/bin/bash
# the BSW kernel without the carried candidate (112 registers instead of 153): parity, then the step
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/exp_r6_slim_tests.log 2>&1
tail -3 gpurun_out/exp_r6_slim_tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 > gpurun_out/exp_r6_slim_bench$i.json 2> gpurun_out/exp_r6_slim_bench$i.err
python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_slim_bench$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("parity_diffs"), d["roofline"])
P
done
