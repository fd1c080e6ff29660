#!/bin/bash
# Round 5, seventh GPU call: isaac-align after the flagged clusters went to host threads side by side and the page-locked buffers to the reference stage;
# where the 10M-pair run's time goes (finer timers), with the switches one at a time on the same files
timeout 900 python -m pytest tests/test_cli.py "tests/test_gpu_parity.py::test_flagged_clusters_are_resolved_on_the_host" -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5g_gputests.log
cat gpurun_out/r5g_gputests.log
CLI_GENOME_BASES=3.1e9 CLI_PAIRS=1e7 CLI_WORK=/dev/shm CLI_INFLATE=1 \
CLI_ARGS=";;ISAAC_ALIGN_TIMING_NO_RESOLUTION=1;--bin-records 20000000;ISAAC_ALIGN_STREAM_SELECTION=1" \
timeout 1500 python scripts/cli_timing.py > gpurun_out/r5g_cli_timing.log 2>&1
grep -E "rc |cli_end_to_end|identical|md5" gpurun_out/r5g_cli_timing.log | cut -c1-1800
