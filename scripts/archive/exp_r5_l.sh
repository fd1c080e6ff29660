#!/bin/bash
# Round 5, twelfth GPU call: the table through two copy streams (against one, and with more copying threads), on the same files; test_cli with the selection streamed
# where a run has many loads; the at-scale tests with the round's program
CLI_GENOME_BASES=3.1e9 CLI_PAIRS=1e7 CLI_READ_LENGTH=150 CLI_WORK=/dev/shm \
CLI_ARGS=";;ISAAC_GPU_LOAD_STREAMS=1;ISAAC_GPU_LOAD_THREADS=16;ISAAC_GPU_LOAD_THREADS=12;--devices 0,0" \
timeout 1500 python scripts/cli_timing.py > gpurun_out/r5l_cli_timing.log 2>&1
grep -E "rc |cli_end_to_end" gpurun_out/r5l_cli_timing.log | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('cli_end_to_end'):
        d = json.loads(l.split(' ', 1)[1]); print({k: d[k] for k in ('reference_s', 'reference_table_s', 'load_and_find_s', 'select_and_bin_s', 'build_and_write_s', 'file_write_s', 'total_s', 'wall_s', 'reads_per_s_without_reference_load', 'selection_streamed')})
    else: print(l.strip()[:200])"
timeout 1500 python -m pytest tests/test_cli.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5l_gputests.log
cat gpurun_out/r5l_gputests.log
bash scripts/gpu_scale_tests.sh r5l 2400
