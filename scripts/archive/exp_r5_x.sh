#!/bin/bash
# Round 5, the round-end set on the round's last code (candidates in registers, the cluster view out of scratch): tag r5_end
bash scripts/gpu_round_check.sh r5_end
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_steps20_warmup5_r5_end.json 2> gpurun_out/bench_steps20_warmup5_r5_end.err
python3 - <<'PY'
import json
for f in ('gpurun_out/bench_default_r5_end.json','gpurun_out/bench_steps20_warmup5_r5_end.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    c=d['config'].get('cli_end_to_end') or {}
    print(f, d['value'], d['ms_per_step'], d.get('parity_diffs'), d.get('records_sha1','')[:8], 'cli', c.get('reads_per_s'), c.get('reads_per_s_without_reference_load'), c.get('error'))
PY
