#!/bin/bash
# fewer banded-SW wavefronts per CU (more LDS asked for per workgroup than used): does the room they leave help the other contexts' kernels more than it costs?
for b in 0 16000 18000 20000 26000; do
  ISAAC_GPU_GAPPED_LDS=$b python bench.py --steps 20 --warmup 5 --no-pcie-pass --no-bam-pass --no-cli-pass --cpu-sample-pairs 1000 > gpurun_out/exp_r6_lds_$b.json 2> gpurun_out/exp_r6_lds_$b.err
  python -c "
import json; d=json.load(open('gpurun_out/exp_r6_lds_$b.json')); r=d['roofline']; print('lds $b:', d['value'], d['ms_per_step'], d['parity_diffs'], r['kernel_ms_per_step']['gapped_fragments'], r['single_stream']['kernel_ms_per_step']['gapped_fragments'], r['kernel_ms_per_step']['rescue_align'])"
done | tee gpurun_out/exp_r6_gapped_lds.log
