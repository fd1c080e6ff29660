#!/bin/bash
# Round 5, fourteenth GPU call: test_cli again (pool of page-locked buffers sized for the build stage), then the 100 M-pair run four times on the same files --
# streamed / not streamed / streamed / not streamed -- because the first run after the files are written is always the slowest whatever its switches
timeout 1500 python -m pytest tests/test_cli.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5n_gputests.log
cat gpurun_out/r5n_gputests.log
timeout 2400 python scripts/cli_headline.py --pairs 100000000 --lanes 4 --devices 0,0 --extra-env "ISAAC_ALIGN_STREAM_SELECTION=0;ISAAC_ALIGN_STREAM_SELECTION=1;ISAAC_ALIGN_STREAM_SELECTION=0" --out gpurun_out/r5_cli_headline_100M_d.json > gpurun_out/r5_cli_headline_100M_d.log 2>&1
echo rc $?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5_cli_headline_100M_d.json'))
print({k:v for k,v in d.items() if k not in('timing','stderr_tail','sampled_tiles','lane_statistics','extra_runs')})
keys=('reference_s','reference_table_s','load_and_find_s','select_and_bin_s','select_busy_s','selection_streamed','build_and_write_s','build_records_s','build_deflate_s','build_download_s','build_device_s','file_write_s','total_s')
print({k:d['timing'].get(k) for k in keys})
for e in d['extra_runs']: print(e['env'], e['wall_s'], e.get('reads_per_s_without_reference_load'), {k:e['timing'].get(k) for k in keys})
PY
