#!/bin/bash
# Round 5: device deflate with 16 KB / 20 KB of input per BGZF block instead of 24 KB (more blocks in flight per CU), on the bench's BAM pass
for v in default d20 d16 d16h10 default; do
  if [ "$v" = default ]; then unset ISAAC_GPU_LIBRARY; else export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; fi
  python bench.py --steps 4 --warmup 1 --no-cli-pass --no-cpu-baseline --no-pcie-pass --no-single-stream-pass 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['config']['bam_output']
print('deflate [$v]:', b.get('bgzf_deflate_GB_per_s'), 'GB/s, ratio', b.get('bgzf_deflate_ratio'), 'ms', b.get('bgzf_deflate_ms'), 'inflates', b.get('bgzf_deflate_inflates_to_records'))" | tee -a gpurun_out/exp_r5_deflate_block.log
done
