#!/bin/bash
# k_select / k_plan_rescue with more registers per thread (fewer spills, less scratch traffic) under the default three contexts
for v in w2 w3; do
  if [ -n "$v" ]; then export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; else unset ISAAC_GPU_LIBRARY; fi
  python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-bam-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['single_stream']['kernel_ms_per_step']
print('variant[$v]', d['value'], d['ms_per_step'], 'single-stream select', k['select'], 'plan', k['plan_rescue'], 'single select phase', d['roofline']['single_stream']['select_ms_per_step'])"
done
