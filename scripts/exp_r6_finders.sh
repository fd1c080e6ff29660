#!/bin/bash
# lookup contexts of the timed region (bench.py --finders): one (a host wait between the lookups), two and three side by side
mkdir -p gpurun_out
for n in 1 2 3 2; do
python bench.py --finders $n --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-cli-pass --no-single-stream-pass > gpurun_out/exp_r6_finders_$n.json 2> gpurun_out/exp_r6_finders_$n.err
python - <<P
import json
d=json.loads(open("gpurun_out/exp_r6_finders_$n.json").read().strip().splitlines()[-1])
print("finders $n:", d["value"], d["ms_per_step"], d["parity_diffs"], d["records_sha1"][:8], "lookup phase", d["config"]["lookup_phase_ms_per_step"], "find", d["roofline"]["kernel_ms_per_step"]["find_matches"], d["config"]["hbm_used_gb"])
P
done
