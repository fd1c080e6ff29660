#!/bin/bash
# Round 5: the whole GPU suite on the round's last commit, and the default bench line
python __graft_entry__.py smoke > gpurun_out/smoke_r5_head.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_r5_head.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py 2>&1 | tail -6 > gpurun_out/gputests_r5_head.log
timeout 2400 python -m pytest tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/scale_r5_head.log
python bench.py > gpurun_out/bench_default_r5_head.json 2> gpurun_out/bench_default_r5_head.err
tail -2 gpurun_out/smoke_r5_head.log; cat gpurun_out/gputests_r5_head.log gpurun_out/scale_r5_head.log
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_default_r5_head.json') if l.startswith('{')][-1])
c=d['config'].get('cli_end_to_end') or {}
print(d['value'], d['ms_per_step'], d.get('parity_diffs'), d.get('records_sha1','')[:8], 'cli', c.get('reads_per_s'), c.get('reads_per_s_without_reference_load'), c.get('error'))
PY
