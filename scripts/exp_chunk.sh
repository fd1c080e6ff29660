#!/bin/bash
# chunk-size sweep (experiment helper)
B="python bench.py --steps 4 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms_total"], d["roofline"]["heavy_clusters"])'
for c in 131072 262144 524288; do echo "== chunk $c"; ISAAC_GPU_CHUNK_CLUSTERS=$c $B 2>&1 | tail -1 | python -c "$P"; done
