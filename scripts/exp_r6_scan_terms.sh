#!/bin/bash
# table terms of the ungapped scan asked for one / two / four at a time (-DISAAC_SCAN_TERMS), k_rescue_align at six or five waves per SIMD: the driver's command
mkdir -p gpurun_out
for v in "" terms1 terms4w5 terms2w5; do
  if [ -n "$v" ]; then export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; else unset ISAAC_GPU_LIBRARY; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-pass --no-bam-pass > gpurun_out/exp_r6_scan_terms_$v.json 2> gpurun_out/exp_r6_scan_terms_$v.err
  python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_scan_terms_$v.json").read().strip().splitlines()[-1])
a=d["roofline"]["single_stream"]["kernel_ms_per_step"]
print("variant[$v]", d["value"], d["ms_per_step"], d["records_sha1"][:8], "alone: rescue_align", a["rescue_align"], "align_candidates", a["align_candidates"], "rescan", a["gapped_fragments_rescan"], "indel", a["indel_fragments"])
P
done
