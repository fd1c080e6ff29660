#!/bin/bash
# Round 5: the lean fragment kernels' view out of scratch as well (k_build_fragments 80 -> 0 bytes, k_finish_candidates / k_finish_fragments 144 -> 72): all GPU tests but
# the at-scale ones, one-context kernel times, three contexts
timeout 1800 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py -x 2>&1 | tail -6 > gpurun_out/r5w_gputests.log
cat gpurun_out/r5w_gputests.log
VARIANTS="default inplace default" KEYS="build_fragments finish_candidates finish_fragments align_candidates sums_wave rescue_align" STEPS=6 bash scripts/exp_variants.sh 2>&1 | tee gpurun_out/exp_r5_lean_view.log
for v in default default; do
  python bench.py --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three contexts [$v]:', d['value'], d['ms_per_step'], d.get('records_sha1')[:8], d.get('parity_diffs'))" | tee -a gpurun_out/exp_r5_lean_view.log
done
