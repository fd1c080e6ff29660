#!/bin/bash
# Round 5, last call: smoke, the GPU suite without the at-scale tests, the default bench line -- on the last commit (k_rescue_align at six waves): tag r5_tail
python __graft_entry__.py smoke > gpurun_out/smoke_r5_tail.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_r5_tail.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py 2>&1 | tail -6 > gpurun_out/gputests_r5_tail.log
python bench.py > gpurun_out/bench_default_r5_tail.json 2> gpurun_out/bench_default_r5_tail.err
tail -2 gpurun_out/smoke_r5_tail.log; cat gpurun_out/gputests_r5_tail.log
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_default_r5_tail.json') if l.startswith('{')][-1]); c=d['config'].get('cli_end_to_end') or {}
print(d['value'], d['ms_per_step'], d.get('parity_diffs'), d.get('records_sha1','')[:8], c.get('reads_per_s_without_reference_load'), c.get('error'), d['roofline']['single_stream']['kernel_ms_per_step'].get('rescue_align'))"
