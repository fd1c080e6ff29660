#!/bin/bash
# A/B of library variants built with ISAAC_GPU_BUILD_TAG=<tag> ISAAC_GPU_BUILD_FLAGS=-D...: one context, the default workload, the kernels named in KERNELS.
# usage: VARIANTS="tagA tagB" KERNELS="rescue_windows select" scripts/exp_variants.sh
for v in $VARIANTS; do
  export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so
  python bench.py --contexts ${CONTEXTS:-1} --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass ${BENCH_ARGS} 2>/dev/null | KERNELS="$KERNELS" python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
print('variant[$v]', d['value'], d['ms_per_step'], ' '.join('%s %s' % (n, k.get(n)) for n in os.environ['KERNELS'].split()), d.get('records_sha1'))"
done
