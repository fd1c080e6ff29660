#!/bin/bash
# the round's closing measurements in one call: the round check (smoke, GPU tests, at-scale tests, default bench line, kernel trace, counter passes), the driver's
# bench command, configuration 4 (2 x 250 with indel reads), and 100 M pairs through isaac-align with sampled tiles checked against the oracle
TAG=${1:-r6z}
bash scripts/gpu_round_check.sh $TAG > gpurun_out/round_check_$TAG.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_steps20_warmup5_$TAG.json 2> gpurun_out/bench_steps20_warmup5_$TAG.err
python bench.py --read-length 250 --indel-read-fraction 0.05 --indel-max 10 --steps 4 --no-cli-pass > gpurun_out/config4_bench_$TAG.json 2> gpurun_out/config4_bench_$TAG.err
timeout 1500 python scripts/cli_headline.py --pairs 100000000 --lanes 4 --devices 0,0 --out gpurun_out/cli_headline_100M_$TAG.json > gpurun_out/cli_headline_100M_$TAG.log 2>&1
tail -12 gpurun_out/round_check_$TAG.log | cut -c1-300
for f in bench_default bench_steps20_warmup5 config4_bench; do python -c "
import json; d=json.load(open('gpurun_out/${f}_$TAG.json')); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], d['parity_diffs'], d['records_sha1'][:8], (d['config'].get('pcie_inclusive') or {}).get('reads_per_s'), (d['config'].get('cli_end_to_end') or {}).get('reads_per_s'), (d['config'].get('cli_end_to_end') or {}).get('reads_per_s_without_reference_load'))"; done
python -c "
import json; d=json.load(open('gpurun_out/cli_headline_100M_$TAG.json')); print({k: v for k, v in d.items() if k in ('pairs','rc','wall_s','reads_per_s','reads_per_s_without_reference_load','peak_device_gb','peak_host_gb','sampled_parity_diffs')})"
