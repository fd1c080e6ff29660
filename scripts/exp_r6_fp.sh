#!/bin/bash
# fingerprints in the seed lookup's directory: parity tests, then the driver's command
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/exp_r6_fp_tests.log 2>&1
tail -3 gpurun_out/exp_r6_fp_tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-pass --no-bam-pass > gpurun_out/exp_r6_fp_bench$i.json 2> gpurun_out/exp_r6_fp_bench$i.err
python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_fp_bench$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], d["parity_diffs"], d["records_sha1"][:8], "find", r["kernel_ms_per_step"]["find_matches"], r["seed_lookup"])
P
done
