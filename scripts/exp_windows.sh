#!/bin/bash
# experiment helper: which part of k_rescue_windows costs the time
B="python bench.py --steps 2 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms_total"]["rescue_windows"])'
for e in 0 1 2 3 4 7; do echo "== exp $e"; ISAAC_GPU_EXP=$e $B 2>&1 | tail -1 | python -c "$P"; done
