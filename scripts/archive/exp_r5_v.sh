#!/bin/bash
# Round 5: the cluster view out of scratch memory in the sums kernels and k_align_candidates (ClusterFragments::list): parity, one-context kernel times against
# the library of two calls ago, three contexts, the counter passes for the written bytes
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r5v_gputests.log
cat gpurun_out/r5v_gputests.log
VARIANTS="default inplace default" KEYS="rescue_align align_candidates sums_wave sums_large sums_xl sums_huge select" STEPS=6 bash scripts/exp_variants.sh 2>&1 | tee gpurun_out/exp_r5_view_out_of_scratch.log
for v in default default; do
  python bench.py --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three contexts [$v]:', d['value'], d['ms_per_step'], d.get('records_sha1')[:8], d.get('parity_diffs'))" | tee -a gpurun_out/exp_r5_view_out_of_scratch.log
done
bash scripts/pmc_traffic.sh > gpurun_out/pmc_traffic_r5v.log 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/pmc_summary_r5v.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_summary_r5v.json'))
for k in ('k_rescue_align','k_align_candidates','k_cluster_sums16','k_cluster_sums','k_cluster_sums_mid','k_select','k_finish_candidates','k_finish_fragments','k_build_fragments'):
    v=d[k]; print(k, v['launches'], 'read %.2f write %.2f GB'%(v['hbm_read_bytes_per_launch']/1e9, v['hbm_write_bytes_per_launch']/1e9))
PY
