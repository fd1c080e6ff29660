#!/bin/bash
# Round 5, sixth GPU call: the banded Smith-Waterman kernel with its next problem fetched beside the current one (parity, then the bench's single-stream times), isaac-align again
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r5f_gputests.log
python bench.py > gpurun_out/r5f_bench_default.json 2> gpurun_out/r5f_bench_default.err
cat gpurun_out/r5f_gputests.log; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5f_bench_default.json') if l.startswith('{')][-1])
print("default:", d['value'], d['ms_per_step'], d.get('parity_diffs'), d.get('records_sha1'))
c=d['config']['cli_end_to_end']; print("cli:", c.get('reads_per_s'), c.get('reads_per_s_without_reference_load'), c.get('stages_s'), c.get('error'))
s=d['roofline']['single_stream']; print("single:", s['select_ms_per_step'], s['band_cell_updates_per_s']); print(s['kernel_ms_per_step'])
PY
