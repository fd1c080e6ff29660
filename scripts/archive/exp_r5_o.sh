#!/bin/bash
# Round 5, fifteenth GPU call: the build stage of the 100 M-pair run with its new pieces switched one at a time on one box and one set of files (the boxes differ by
# a second or two, and the first run on fresh files is the slowest): default twice, then blocks fetched before the next bin, four warmed buffers, eight bins ahead
timeout 2400 python scripts/cli_headline.py --pairs 100000000 --lanes 4 --devices 0,0 --sample-tiles 1 \
  --extra-env ";ISAAC_ALIGN_SYNC_DOWNLOADS=1;ISAAC_ALIGN_WARM_BUFFERS=4;ISAAC_ALIGN_BUILD_AHEAD=8;ISAAC_ALIGN_WARM_BUFFERS=0;" --out gpurun_out/r5_cli_headline_100M_e.json > gpurun_out/r5_cli_headline_100M_e.log 2>&1
echo rc $?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5_cli_headline_100M_e.json'))
keys=('reference_s','reference_table_s','load_and_find_s','select_and_bin_s','build_and_write_s','build_records_s','build_deflate_s','build_download_s','build_writer_wait_s','build_release_s','build_device_s','file_write_s','total_s')
print('first', d['wall_s'], d['sampled_parity_diffs'], {k:d['timing'].get(k) for k in keys})
for e in d['extra_runs']: print(e['env'], e['wall_s'], {k:e['timing'].get(k) for k in keys})
PY
