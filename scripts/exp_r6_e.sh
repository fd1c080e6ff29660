#!/bin/bash
# round 6, after the rescue summaries: the random-lines probe, the GPU parity tests that touch the rescue, the driver's bench command
bash scripts/exp_r6_random_lines.sh > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_launch.py -q -x -m gpu 2>&1 | tail -4 > gpurun_out/gputests_r6e.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r6e.json 2> gpurun_out/bench_r6e.err
cat gpurun_out/exp_r6_random_lines.log; cat gpurun_out/gputests_r6e.log
python -c "
import json; d=json.load(open('gpurun_out/bench_r6e.json')); r=d['roofline']; s=r['single_stream']['kernel_ms_per_step']; print(d['value'], d['ms_per_step'], d['parity_diffs'], d['records_sha1'], s['rescue_gapped_plan'], s['rescue_align'], s['gapped_fragments'], r['single_stream']['select_ms_per_step'], d['config']['pcie_inclusive']['reads_per_s'])"
