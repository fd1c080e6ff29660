#!/bin/bash
# kernel-trace profile of a short default bench run: gpurun_out/prof_<tag>/.../*kernel_stats.csv and the bench line of that run
TAG=${1:-r2}
R=$PWD
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8      # bench.py sets it for itself, but under rocprofv3 the runtime is up before Python starts
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass > $R/gpurun_out/bench_under_rocprof_$TAG.json 2> $R/gpurun_out/bench_under_rocprof_$TAG.err
cd $R
f=$(ls -t $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv") | head -1)
python3 scripts/trace_gaps.py $(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1) 20 gpurun_out/trace_timed_$TAG.csv > gpurun_out/trace_gaps_$TAG.txt 2>&1
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete      # tens of MB; the stats file is what is kept
echo "stats: $f"; head -32 "$f" | cut -c1-200
