#!/bin/bash
# experiment helper: bench lines with and without an environment knob (KNOB="NAME=value")
# NOT to be wrapped under rocprofv3: it starts the program through `env`, a hop that re-executes after the profiler has initialised the GPU
# (put the program itself after `--`, as scripts/prof_trace.sh and scripts/pmc_*.sh do).
B="python bench.py --steps 4 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_total"]; print(d["value"], {n: k[n] for n in sys.argv[1:]})'
KEYS="${KEYS:-select select_heavy plan_rescue}"
echo "== default"; $B 2>&1 | tail -1 | python -c "$P" $KEYS
for k in $KNOBS; do echo "== $k"; env $k $B 2>&1 | tail -1 | python -c "$P" $KEYS; done
