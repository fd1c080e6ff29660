#!/bin/bash
# experiment helper: bench lines with and without an environment knob (KNOB="NAME=value")
B="python bench.py --steps 4 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_total"]; print(d["value"], {n: k[n] for n in sys.argv[1:]})'
KEYS="${KEYS:-select select_heavy plan_rescue}"
echo "== default"; $B 2>&1 | tail -1 | python -c "$P" $KEYS
for k in $KNOBS; do echo "== $k"; env $k $B 2>&1 | tail -1 | python -c "$P" $KEYS; done
