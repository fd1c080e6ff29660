#!/bin/bash
# Round 5: the rescue problem's candidate range read once in the summary walks (k_rescue_gapped_plan): parity, kernel times one context (r5_end: rescue_gapped_plan 0.887), three contexts
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r5_hoist_gputests.log
cat gpurun_out/r5_hoist_gputests.log
VARIANTS="default default" KEYS="rescue_gapped_plan plan_rescue sums_wave rescue_align select" STEPS=6 bash scripts/exp_variants.sh 2>&1 | tee gpurun_out/exp_r5_hoist.log
for v in default default; do
  python bench.py --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three contexts [$v]:', d['value'], d['ms_per_step'], d.get('records_sha1')[:8], d.get('parity_diffs'))" | tee -a gpurun_out/exp_r5_hoist.log
done
