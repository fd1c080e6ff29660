#!/bin/bash
# experiment helper: bench lines of library variants built with different -D flags (isaac_aligner_amd/libisaac_gpu_<tag>.so)
B="python bench.py --steps 4 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_total"]; print(d["value"], {n: k[n] for n in sys.argv[1:]})'
KEYS="${KEYS:-rescue_windows select find_matches}"
for v in "" $VARIANTS; do echo "== lib$v"; ISAAC_GPU_LIBRARY=isaac_aligner_amd/libisaac_gpu$v.so $B 2>&1 | tail -1 | python -c "$P" $KEYS; done
