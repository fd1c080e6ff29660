#!/bin/bash
# k_plan_rescue with the fragment stage's cluster order (default) against an order of its own (ISAAC_GPU_PLAN_REORDER=1): the driver's command, two runs each
mkdir -p gpurun_out
for v in 0 1 0 1; do
ISAAC_GPU_PLAN_REORDER=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-pass --no-bam-pass > gpurun_out/exp_r6_plan_order_$v.json 2> gpurun_out/exp_r6_plan_order_$v.err
python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_plan_order_$v.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernel_ms_per_step"]; a=d["roofline"]["single_stream"]["kernel_ms_per_step"]
print("reorder $v:", d["value"], d["ms_per_step"], d["parity_diffs"], d["records_sha1"][:8], "plan_rescue shared", k["plan_rescue"], "alone", a["plan_rescue"], "rescue_windows alone", a["rescue_windows"])
P
done
