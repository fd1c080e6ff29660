#!/bin/bash
# Round 5, sixteenth GPU call: the BAM index on a thread of its own; test_cli, then the 100 M-pair run three times on one set of files
timeout 1500 python -m pytest tests/test_cli.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5p_gputests.log
cat gpurun_out/r5p_gputests.log
timeout 2400 python scripts/cli_headline.py --pairs 100000000 --lanes 4 --devices 0,0 --sample-tiles 3 \
  --extra-env "ISAAC_ALIGN_BUILD_AHEAD=8;ISAAC_ALIGN_STREAM_SELECTION=0" --out gpurun_out/r5_cli_headline_100M_f.json > gpurun_out/r5_cli_headline_100M_f.log 2>&1
echo rc $?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5_cli_headline_100M_f.json'))
keys=('reference_s','reference_table_s','load_and_find_s','select_and_bin_s','build_and_write_s','build_records_s','build_deflate_s','build_download_s','build_writer_wait_s','index_s','file_write_s','total_s')
print('first', d['wall_s'], d['sampled_parity_diffs'], d['reads_per_s'], d['reads_per_s_without_reference_load'], {k:d['timing'].get(k) for k in keys})
for e in d['extra_runs']: print(e['env'], e['wall_s'], e['reads_per_s'], e['reads_per_s_without_reference_load'], {k:e['timing'].get(k) for k in keys})
PY
