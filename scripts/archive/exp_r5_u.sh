#!/bin/bash
# Round 5: the rescue record stored once by the sums kernels (finishRescueFlat) on top of the candidates in registers: parity, one-context kernel times against
# the in-place library of the last call, the counter passes for the written bytes
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r5u_gputests.log
cat gpurun_out/r5u_gputests.log
VARIANTS="default inplace default" KEYS="rescue_align align_candidates sums_wave sums_large sums_xl sums_huge select" STEPS=6 bash scripts/exp_variants.sh 2>&1 | tee gpurun_out/exp_r5_store_once.log
for v in default default; do
  python bench.py --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three contexts [$v]:', d['value'], d['ms_per_step'], d.get('records_sha1')[:8], d.get('parity_diffs'))" | tee -a gpurun_out/exp_r5_store_once.log
done
bash scripts/pmc_traffic.sh > gpurun_out/pmc_traffic_r5u.log 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/pmc_summary_r5u.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_summary_r5u.json'))
for k in ('k_rescue_align','k_align_candidates','k_cluster_sums16','k_cluster_sums','k_select'):
    v=d[k]; print(k, v['launches'], 'read %.2f write %.2f GB'%(v['hbm_read_bytes_per_launch']/1e9, v['hbm_write_bytes_per_launch']/1e9))
PY
