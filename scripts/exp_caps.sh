#!/bin/bash
# light-capacity sweep for the select stage (experiment helper)
B="python bench.py --steps 4 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_total"]; print(d["value"], "select", k["select"], "heavy", k["select_heavy"], "plan", k["plan_rescue"], d["roofline"]["heavy_clusters"])'
echo "== default"; $B 2>&1 | tail -1 | python -c "$P"
for caps in 32,256,384,128,128,12,768 16,128,384,96,96,12,768 12,96,384,64,64,8,512 8,64,384,48,48,8,384; do echo "== $caps"; ISAAC_GPU_LIGHT_CAPS=$caps $B 2>&1 | tail -1 | python -c "$P"; done
