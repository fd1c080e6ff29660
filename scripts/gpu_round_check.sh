#!/bin/bash
# Round-end GPU check: smoke, the GPU parity suite, the default bench line, and a kernel-trace profile of a short bench run.
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/t.log
python bench.py > gpurun_out/bench_default.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_i -- python3 $R/bench.py --steps 4 --no-cpu-baseline > $R/gpurun_out/bench_rocprof.log 2>&1
cd $R
tail -3 gpurun_out/smoke.log; cat gpurun_out/t.log; tail -1 gpurun_out/bench_default.log; tail -1 gpurun_out/bench_rocprof.log
find gpurun_out/prof_i -name "*kernel_stats.csv" | head
