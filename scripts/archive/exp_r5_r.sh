#!/bin/bash
# Round 5: the realigner behind its filter (BAM-stage and CLI tests); the bench's isaac-align leg (10 M pairs in five loads) with the selection streamed and not
timeout 1500 python -m pytest tests/test_bam.py tests/test_cli.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5r_gputests.log
cat gpurun_out/r5r_gputests.log
CLI_GENOME_BASES=3.1e9 CLI_PAIRS=1e7 CLI_READ_LENGTH=150 CLI_WORK=/dev/shm CLI_INFLATE=1 \
CLI_ARGS="--clusters-at-a-time 2000000;--clusters-at-a-time 2000000 ISAAC_ALIGN_STREAM_SELECTION=0;--clusters-at-a-time 2000000 ISAAC_ALIGN_STREAM_SELECTION=1;--clusters-at-a-time 2000000 ISAAC_ALIGN_STREAM_SELECTION=0;--clusters-at-a-time 2000000 ISAAC_ALIGN_STREAM_SELECTION=1;--clusters-at-a-time 2000000 ISAAC_ALIGN_ORDERLY_EXIT=1" \
timeout 1500 python scripts/cli_timing.py > gpurun_out/r5r_cli_timing.log 2>&1
grep -E "rc |cli_end_to_end|identical" gpurun_out/r5r_cli_timing.log | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('cli_end_to_end'):
        d = json.loads(l.split(' ', 1)[1]); print({k: d.get(k) for k in ('reference_s', 'load_and_find_s', 'select_and_bin_s', 'build_and_write_s', 'build_records_s', 'build_device_s', 'file_write_s', 'total_s', 'wall_s', 'reads_per_s_without_reference_load', 'selection_streamed')})
    else: print(l.strip()[:200])"
