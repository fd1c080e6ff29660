#!/bin/bash
# Round 5, the round-end set (scripts/gpu_round_check.sh) plus the driver's own bench command, and the default line with smaller chunks
bash scripts/gpu_round_check.sh r5_final
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_steps20_warmup5_r5_final.json 2> gpurun_out/bench_steps20_warmup5_r5_final.err
tail -c 600 gpurun_out/bench_steps20_warmup5_r5_final.json | head -c 300; echo
for chunk in 524288 786432; do
  ISAAC_GPU_CHUNK_CLUSTERS=$chunk python bench.py --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $chunk:', d['value'], d['ms_per_step'], d.get('records_sha1'))" | tee -a gpurun_out/exp_r5_chunk_contexts.log
done
