#!/bin/bash
# VALU / SALU / memory instruction counts of every kernel of a short bench run (one rocprofv3 --pmc pass), next to kernel times
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_all -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-cli-pass > $R/gpurun_out/pmc_all.log 2>&1
cd $R
f=$(find gpurun_out/pmc_all -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, re
tot = collections.defaultdict(collections.Counter); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if not m: continue
    tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); disp[m.group(1)].add(r["Dispatch_Id"])
print("kernel launches  VALU/launch  valu_ms  gpu_ms  lane_util  SALU  FLAT  LDS  waves")
for k in sorted(tot):
    n = len(disp[k]); c = tot[k]
    valu = c["SQ_INSTS_VALU"] / n
    print("%-22s %3d %12.0f %8.3f %8.3f %6.2f %12.0f %10.0f %10.0f %8.0f" % (k, n, valu, valu * 4 / 1024 / 2.4e6, c["GRBM_GUI_ACTIVE"] / n / 8 / 2.4e6,
          c["SQ_THREAD_CYCLES_VALU"] / max(1.0, c["SQ_ACTIVE_INST_VALU"] * 64 ), c["SQ_INSTS_SALU"] / n, c["SQ_INSTS_FLAT"] / n, c["SQ_INSTS_LDS"] / n, c["SQ_WAVES"] / n))
PY
