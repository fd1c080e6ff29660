#!/bin/bash
# device deflate on the bench's record stream with library variants: throughput, ratio, inflate check.  usage: VARIANTS="tagA tagB" scripts/exp_deflate.sh
for v in $VARIANTS; do
  if [ "$v" = default ]; then unset ISAAC_GPU_LIBRARY; else export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; fi
  python bench.py --contexts 1 --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-pcie-pass --no-single-stream-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['config']['bam_output']
print('variant[$v]', b['bgzf_deflate_GB_per_s'], 'GB/s ratio', b['bgzf_deflate_ratio'], 'ok', b['bgzf_deflate_inflates_to_records'], 'zlib1', b['bgzf_ratio'], b['bgzf_GB_per_s'])"
done
