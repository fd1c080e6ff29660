#!/bin/bash
# workgroups of k_gapped_jobs (it strides over the problems): shorter-lived workgroups let the other contexts' kernels onto the CUs more often, longer-lived ones amortise the look-ahead
for g in 4096 8192 16384 32768; do
  ISAAC_GPU_GAPPED_GRID=$g python bench.py --steps 20 --warmup 5 --no-pcie-pass --no-bam-pass --no-cli-pass --cpu-sample-pairs 1000 > gpurun_out/exp_r6_grid_$g.json 2> gpurun_out/exp_r6_grid_$g.err
  python -c "
import json; d=json.load(open('gpurun_out/exp_r6_grid_$g.json')); r=d['roofline']; print('grid $g:', d['value'], d['ms_per_step'], d['parity_diffs'], r['kernel_ms_per_step']['gapped_fragments'], r['single_stream']['kernel_ms_per_step']['gapped_fragments'], r['single_stream']['select_ms_per_step'])"
done | tee gpurun_out/exp_r6_gapped_grid.log
