#!/bin/bash
# GPU check used while iterating: parity tests, then a 4 x 500K-pair bench line (kernel totals)
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/t.log
python bench.py --steps 4 --warmup 1 --pairs-per-step 500000 --no-cpu-baseline > gpurun_out/b.log 2>&1
tail -1 gpurun_out/b.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms_total"], d["roofline"]["heavy_clusters"])'
cat gpurun_out/t.log
