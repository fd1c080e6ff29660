#!/bin/bash
# Where k_rescue_windows spends its time: the product library against builds that leave out the enumeration of the candidates, the
# window scan, and the scan and the mate's k-mer table (wrong results, timing only), one context, 1 M pairs per step.
# Build first (here, no GPU needed):  for v in noenum noscan notable; do ISAAC_GPU_BUILD_TAG=rw_$v ISAAC_GPU_BUILD_FLAGS=-DISAAC_TIMING_RW_NO_${v^^} ...; done
for v in "" rw_noenum rw_noscan rw_notable; do
  if [ -n "$v" ]; then export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; else unset ISAAC_GPU_LIBRARY; fi
  python bench.py --genome-bases 300000000 --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
print('variant[$v]', d['ms_per_step'], 'rescue_windows', k['rescue_windows'], 'plan_rescue', k['plan_rescue'], 'rescue_align', k['rescue_align'], 'rescue_calls', d['counters']['rescue_calls'])"
done
