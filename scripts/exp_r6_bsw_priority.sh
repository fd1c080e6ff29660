#!/bin/bash
# the banded SW on a side stream of the lowest priority (ISAAC_GPU_BSW_SIDE_STREAM=1) against the context's own stream: the driver's command, twice each
mkdir -p gpurun_out
for v in 0 1 0 1; do
ISAAC_GPU_BSW_SIDE_STREAM=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-pass --no-bam-pass > gpurun_out/exp_r6_bsw_priority_$v.json 2> gpurun_out/exp_r6_bsw_priority_$v.err
python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_bsw_priority_$v.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernel_ms_per_step"]
print("side stream $v:", d["value"], d["ms_per_step"], d["parity_diffs"], "gapped", k["gapped_fragments"], "rescue_align", k["rescue_align"], "rescue_windows", k["rescue_windows"], "sums_wave", k["sums_wave"])
P
done
