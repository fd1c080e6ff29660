#!/bin/bash
# Round-end GPU check in one call: smoke, the GPU parity suite (the at-scale tests apart, with their output kept), the default bench line, a
# kernel-trace profile of a short bench run and the HBM-traffic counter passes; the summaries land in gpurun_out/ (copy what is to be kept
# into profiles/)
TAG=${1:-r3}
python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py 2>&1 | tail -8 > gpurun_out/gputests_$TAG.log
bash scripts/gpu_scale_tests.sh $TAG 2400 > /dev/null 2>&1
python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err
bash scripts/prof_trace.sh $TAG > gpurun_out/prof_trace_$TAG.log 2>&1
bash scripts/pmc_traffic.sh > gpurun_out/pmc_traffic_$TAG.log 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/pmc_summary_$TAG.json
tail -2 gpurun_out/smoke_$TAG.log; cat gpurun_out/gputests_$TAG.log; tail -12 gpurun_out/scale_$TAG.log; tail -c 1500 gpurun_out/bench_default_$TAG.json; tail -3 gpurun_out/pmc_traffic_$TAG.log
