#!/bin/bash
# Round 5: candidates built in registers and stored once (k_rescue_align, k_align_candidates) against the in-place form: parity, then the kernels one context, then three
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r5t_gputests.log
cat gpurun_out/r5t_gputests.log
VARIANTS="default inplace default inplace" KEYS="rescue_align align_candidates indel_fragments finish_candidates" STEPS=6 bash scripts/exp_variants.sh 2>&1 | tee gpurun_out/exp_r5_cand_registers.log
for v in default inplace default inplace; do
  if [ "$v" = default ]; then unset ISAAC_GPU_LIBRARY; else export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; fi
  python bench.py --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three contexts [$v]:', d['value'], d['ms_per_step'], d.get('records_sha1')[:8], d.get('parity_diffs'))" | tee -a gpurun_out/exp_r5_cand_registers.log
done
