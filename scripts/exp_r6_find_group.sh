#!/bin/bash
# lanes per cluster in k_find_matches: 8 (the product) against 16 and 4 (variant builds), the driver's command
mkdir -p gpurun_out
for v in "" find16 find4; do
  if [ -n "$v" ]; then export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; else unset ISAAC_GPU_LIBRARY; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass > gpurun_out/exp_r6_find_group_$v.json 2> gpurun_out/exp_r6_find_group_$v.err
  python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_find_group_$v.json").read().strip().splitlines()[-1])
print("variant[$v]", d["value"], d["ms_per_step"], d["parity_diffs"], d["records_sha1"][:8], "find", d["roofline"]["kernel_ms_per_step"]["find_matches"])
P
done
