#!/bin/bash
# Round 5, thirteenth GPU call: a bin's blocks downloaded while the next bin is encoded (test_cli, the full-size CLI test), then the 100 M-pair run as the program
# now runs it by itself (selections streamed), and once more with three workers
timeout 1500 python -m pytest tests/test_cli.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5m_gputests.log
cat gpurun_out/r5m_gputests.log
timeout 2400 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -s -k "isaac_align_on_the_full_size" 2>&1 | tail -25 > gpurun_out/scale_r5m.log
tail -12 gpurun_out/scale_r5m.log
timeout 2400 python scripts/cli_headline.py --pairs 100000000 --lanes 4 --devices 0,0 --extra-env "|--devices 0,0,0;ISAAC_ALIGN_STREAM_SELECTION=0" --out gpurun_out/r5_cli_headline_100M_c.json > gpurun_out/r5_cli_headline_100M_c.log 2>&1
echo rc $?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5_cli_headline_100M_c.json'))
print({k:v for k,v in d.items() if k not in('timing','stderr_tail','sampled_tiles','lane_statistics','extra_runs')})
t=d['timing']; t.pop('bin_ranges',None); print(t)
for e in d['extra_runs']: print(e)
PY
