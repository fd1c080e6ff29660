#!/bin/bash
# light-capacity sweep for the select stage (experiment helper)
B="python bench.py --steps 4 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline"
echo "== default"; $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_total'], d['roofline']['heavy_clusters'])"
echo "== A 256"; ISAAC_GPU_LIGHT_CAPS=256,1024,384,1024,1024,32,1024 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_total'], d['roofline']['heavy_clusters'])"
echo "== B 1000"; ISAAC_GPU_CHUNK_CLUSTERS=65536 ISAAC_GPU_LIGHT_CAPS=1000,4096,384,4096,4096,64,2048 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_total'], d['roofline']['heavy_clusters'])"
echo "== C 1000 big"; ISAAC_GPU_CHUNK_CLUSTERS=65536 ISAAC_GPU_LIGHT_CAPS=1000,8192,384,16384,16384,256,8192 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_total'], d['roofline']['heavy_clusters'])"
