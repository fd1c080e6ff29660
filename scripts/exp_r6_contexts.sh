#!/bin/bash
# how many contexts share the GPU in the timed region (bench.py --contexts): the step time for 2, 3, 4, 5 on the round-6 kernels (banded SW at three waves per SIMD)
for n in ${CONTEXTS:-2 3 4 5}; do
  python bench.py --contexts $n --steps 20 --warmup 5 --no-pcie-pass --no-bam-pass --no-cli-pass --no-single-stream-pass --cpu-sample-pairs 1000 > gpurun_out/exp_r6_contexts_$n.json 2> gpurun_out/exp_r6_contexts_$n.err
  python -c "
import json; d=json.load(open('gpurun_out/exp_r6_contexts_$n.json')); print('contexts $n:', d['value'], d['ms_per_step'], d['parity_diffs'], d['config']['hbm_used_gb'])"
done | tee gpurun_out/exp_r6_contexts.log
