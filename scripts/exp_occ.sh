#!/bin/bash
# occupancy-target sweep for k_select / k_build_fragments (experiment helper; variants built with -DISAAC_SELECT_WAVES=..)
B="python bench.py --steps 4 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_total"]; print(d["value"], "select", k["select"], "heavy", k["select_heavy"], "build", k["build_fragments"], "plan", k["plan_rescue"])'
for v in "" _w2 _w4 _w8; do echo "== lib$v"; ISAAC_GPU_LIBRARY=isaac_aligner_amd/libisaac_gpu$v.so $B 2>&1 | tail -1 | python -c "$P"; done
