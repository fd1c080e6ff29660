#!/bin/bash
# Round 5, fourth GPU call: the whole GPU suite (the at-scale tests apart), the default bench line, and the headline configuration through isaac-align (100 M pairs)
python __graft_entry__.py smoke > gpurun_out/r5d_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r5d_smoke.log
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py 2>&1 | tail -40 > gpurun_out/r5d_gputests.log
python bench.py > gpurun_out/r5d_bench_default.json 2> gpurun_out/r5d_bench_default.err
timeout 1500 python scripts/cli_headline.py --pairs ${HEADLINE_PAIRS:-100000000} --lanes 4 --devices 0,0 --out gpurun_out/r5d_cli_headline.json > gpurun_out/r5d_cli_headline.log 2>&1
tail -2 gpurun_out/r5d_smoke.log; cat gpurun_out/r5d_gputests.log; tail -c 1500 gpurun_out/r5d_bench_default.json; tail -3 gpurun_out/r5d_bench_default.err; tail -c 3000 gpurun_out/r5d_cli_headline.log
