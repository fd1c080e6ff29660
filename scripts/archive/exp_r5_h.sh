#!/bin/bash
# Round 5, eighth GPU call: isaac_gpu_share_reference (parity tests, isaac-align's further contexts), then the default bench line with its isaac-align leg
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r5h_gputests.log
cat gpurun_out/r5h_gputests.log
free -g | head -2; df -h /dev/shm | tail -1; nproc
python bench.py > gpurun_out/r5h_bench_default.json 2> gpurun_out/r5h_bench_default.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5h_bench_default.json') if l.startswith('{')][-1])
print("default:", d['value'], d['ms_per_step'], d.get('parity_diffs'), d.get('records_sha1'))
c=d['config']['cli_end_to_end']; print("cli:", c.get('reads_per_s'), c.get('reads_per_s_without_reference_load'), c.get('stages_s'), c.get('error'))
s=d['roofline']['single_stream']; print("single:", s['select_ms_per_step'], s['band_cell_updates_per_s'])
print("pcie:", {k: v for k, v in d['config'].items() if 'pcie' in k})
PY
