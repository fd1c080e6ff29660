#!/bin/bash
# k_rescue_windows after a change: parity, the sections' ticks (stamps build, one context, small genome), then the step
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/exp_r6_rw_tests.log 2>&1
tail -3 gpurun_out/exp_r6_rw_tests.log
bash scripts/exp_r6_rw_stamps.sh 2>&1 | grep -E "^stamp +([0-9]|1[0-2]):|^[0-9.]+ [0-9.]+$"
python bench.py --steps 20 --warmup 5 > gpurun_out/exp_r6_rw_bench.json 2> gpurun_out/exp_r6_rw_bench.err
python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_rw_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], d["config"].get("parity_diffs"), "rescue_windows shared", r["kernel_ms_per_step"]["rescue_windows"], "alone", r["single_stream"]["kernel_ms_per_step"]["rescue_windows"])
P
