#!/bin/bash
# round 6, after the dense second iteration of k_find_matches: the parity tests (match sets at every read length, the at-scale match set), the driver's bench command
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -q -x -m gpu 2>&1 | tail -4 > gpurun_out/gputests_r6f.log
timeout 1500 python -m pytest tests/test_gpu_scale.py -q -x -m gpu -k "find or configuration_2" 2>&1 | tail -4 >> gpurun_out/gputests_r6f.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r6f.json 2> gpurun_out/bench_r6f.err
cat gpurun_out/gputests_r6f.log
python -c "
import json; d=json.load(open('gpurun_out/bench_r6f.json')); r=d['roofline']; s=r['single_stream']['kernel_ms_per_step']; print(d['value'], d['ms_per_step'], d['parity_diffs'], d['records_sha1'], r['kernel_ms_per_step']['find_matches'], d['counters']['probes'], d['counters']['matches'], d['config']['pcie_inclusive']['reads_per_s'])"
