#!/bin/bash
# A/B of library variants (ISAAC_GPU_BUILD_TAG builds) on the bench workload, one context: per-kernel ms per step, the step time and the records' checksum.
# usage: VARIANTS="default tagA tagB" [KEYS="rescue_align align_candidates"] scripts/exp_variants.sh
for v in $VARIANTS; do
  if [ "$v" = default ]; then unset ISAAC_GPU_LIBRARY; else export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; fi
  python bench.py --contexts 1 --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-pcie-pass --no-single-stream-pass --no-bam-pass --no-cli-pass 2>/dev/null | KEYS="$KEYS" python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
keys=os.environ.get('KEYS','').split() or sorted(k)
print('variant[$v]', d['ms_per_step'], 'ms/step', d['records_sha1'][:12], 'diffs', d.get('parity_diffs'), ' '.join('%s=%.3f' % (x, k[x]) for x in keys if x in k), 'sum=%.2f' % sum(k.values()))"
done
