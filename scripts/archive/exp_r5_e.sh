#!/bin/bash
# Round 5, fifth GPU call: the CLI and BAM suites on the streaming isaac-align, the MAPQ resolution test, FETCH_SIZE calibration, the default bench line,
# the counter passes and the kernel trace, and BASELINE configuration 4
timeout 1500 python -m pytest tests/test_cli.py tests/test_bam.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -30 > gpurun_out/r5e_gputests.log
bash scripts/exp_fetch_calib.sh > /dev/null 2>&1
python bench.py > gpurun_out/r5e_bench_default.json 2> gpurun_out/r5e_bench_default.err
bash scripts/pmc_traffic.sh > gpurun_out/r5e_pmc_traffic.log 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/r5e_pmc_summary.json
bash scripts/prof_trace.sh r5e > gpurun_out/r5e_prof_trace.log 2>&1
python bench.py --read-length 250 --indel-read-fraction 0.05 --indel-max 10 --steps 4 --no-cli-pass > gpurun_out/r5e_config4_bench.json 2> gpurun_out/r5e_config4_bench.err
cat gpurun_out/r5e_gputests.log; cat gpurun_out/fetch_calib.log; tail -c 1200 gpurun_out/r5e_bench_default.json; tail -3 gpurun_out/r5e_pmc_traffic.log; tail -c 600 gpurun_out/r5e_config4_bench.json
