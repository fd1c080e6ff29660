for v in nodp; do
  if [ -n "$v" ]; then export ISAAC_GPU_LIBRARY=$PWD/isaac_aligner_amd/libisaac_gpu_$v.so; else unset ISAAC_GPU_LIBRARY; fi
  python bench.py --genome-bases 300000000 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-bam-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
print('variant[$v]', d['ms_per_step'], 'gapped_fragments', k['gapped_fragments'], 'gapped_rescue', k['gapped_rescue'], 'bsw_jobs', d['counters']['bsw_jobs'], d['counters']['rescue_bsw'])"
done
