#!/bin/bash
# k_find_matches at 7 (the product: 70 registers), 8 (64 registers, two spilled) and 4 waves per SIMD: how much of its time is waiting that more waves would cover
# build first:  for w in 8 4; do ISAAC_GPU_BUILD_TAG=find${w}w ISAAC_GPU_BUILD_FLAGS=-DISAAC_WAVES_FIND=$w python -m isaac_aligner_amd.build; done
mkdir -p gpurun_out
for v in "" find8w find4w; do
  if [ -n "$v" ]; then export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; else unset ISAAC_GPU_LIBRARY; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass > gpurun_out/exp_r6_find_waves_$v.json 2> gpurun_out/exp_r6_find_waves_$v.err
  python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_find_waves_$v.json").read().strip().splitlines()[-1])
print("variant[$v]", d["value"], d["ms_per_step"], d["records_sha1"][:8], "find", d["roofline"]["kernel_ms_per_step"]["find_matches"])
P
done
