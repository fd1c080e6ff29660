#!/bin/bash
# Round 5, third GPU call: the whole GPU suite (the at-scale tests apart) on the interleaved table, the prefetched clipper scans and the new isaac-align; the default
# bench line; the lookup with and without the whole-slice fetch (library variant find0: always the bisection)
python __graft_entry__.py smoke > gpurun_out/r5c_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r5c_smoke.log
timeout 1800 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py -x 2>&1 | tail -25 > gpurun_out/r5c_gputests.log
VARIANTS="default find0" KEYS="find_matches select plan_rescue rescue_windows" STEPS=4 bash scripts/exp_variants.sh > gpurun_out/r5c_exp_find.log 2>&1
python bench.py > gpurun_out/r5c_bench_default.json 2> gpurun_out/r5c_bench_default.err
tail -2 gpurun_out/r5c_smoke.log; cat gpurun_out/r5c_gputests.log; cat gpurun_out/r5c_exp_find.log; tail -c 3000 gpurun_out/r5c_bench_default.json; tail -5 gpurun_out/r5c_bench_default.err
