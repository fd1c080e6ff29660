#!/bin/bash
# Round 5, last call: 20 KB BGZF blocks -- the GPU tests (the at-scale BAM and CLI tests with them), then the default bench line and the driver's command once more: tag r5_zz
python __graft_entry__.py smoke > gpurun_out/smoke_r5_zz.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_r5_zz.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py 2>&1 | tail -6 > gpurun_out/gputests_r5_zz.log
timeout 2400 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "bam_stage or isaac_align" 2>&1 | tail -6 > gpurun_out/scale_r5_zz.log
python bench.py > gpurun_out/bench_default_r5_zz.json 2> gpurun_out/bench_default_r5_zz.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_steps20_warmup5_r5_zz.json 2> gpurun_out/bench_steps20_warmup5_r5_zz.err
tail -2 gpurun_out/smoke_r5_zz.log; cat gpurun_out/gputests_r5_zz.log; cat gpurun_out/scale_r5_zz.log
python3 - <<'PY'
import json
for f in ('gpurun_out/bench_default_r5_zz.json','gpurun_out/bench_steps20_warmup5_r5_zz.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    c=d['config'].get('cli_end_to_end') or {}; b=d['config'].get('bam_output') or {}
    print(f, d['value'], d['ms_per_step'], d.get('parity_diffs'), d.get('records_sha1','')[:8], 'cli', c.get('reads_per_s'), c.get('reads_per_s_without_reference_load'), c.get('error'), 'deflate', b.get('bgzf_deflate_GB_per_s'), b.get('bgzf_deflate_ratio'))
PY
