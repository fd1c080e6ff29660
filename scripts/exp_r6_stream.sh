#!/bin/bash
# the lookups beside the selections (round 6) against the two phases apart (--two-phases, rounds 1-5); then the one-rank RCCL path of the same
python bench.py --steps 20 --warmup 5 --no-bam-pass --no-cli-pass --cpu-sample-pairs 1000 > gpurun_out/exp_r6_stream_on.json 2> gpurun_out/exp_r6_stream_on.err
python bench.py --steps 20 --warmup 5 --no-bam-pass --no-cli-pass --cpu-sample-pairs 1000 --two-phases --no-pcie-pass --no-single-stream-pass > gpurun_out/exp_r6_stream_off.json 2> gpurun_out/exp_r6_stream_off.err
for n in on off; do python -c "
import json; d=json.load(open('gpurun_out/exp_r6_stream_$n.json')); r=d['roofline']; print('$n:', d['value'], d['ms_per_step'], d['parity_diffs'], d['records_sha1'], d['config'].get('selections_began_after_lookup_of_step'), d['config'].get('pcie_inclusive',{}).get('reads_per_s'), r['kernel_ms_per_step']['find_matches'])"; done | tee gpurun_out/exp_r6_stream.log
timeout 900 python -m pytest tests/test_bench_launch.py -q -m gpu 2>&1 | tail -3 | tee -a gpurun_out/exp_r6_stream.log
