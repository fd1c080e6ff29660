#!/bin/bash
# k_rescue_windows section by section (shader-clock ticks of the sampled waves' lane 0; -DISAAC_KERNEL_STAMPS build), one context
mkdir -p gpurun_out
export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_stamps.so
python bench.py --genome-bases 300000000 --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass > gpurun_out/exp_r6_rw_stamps.json 2> gpurun_out/exp_r6_rw_stamps.err
grep stamp gpurun_out/exp_r6_rw_stamps.err
python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_rw_stamps.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]["rescue_windows"])
P
