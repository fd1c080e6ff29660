#!/bin/bash
# SQ counter passes for one kernel (KERNEL=regex), each pass its own rocprofv3 run (no tracing combined with --pmc)
KERNEL=${KERNEL:-k_rescue_windows}
R=$PWD
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-include-regex "$KERNEL" --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie-pass > $R/gpurun_out/pmc_$name.log 2>&1
  echo "pass $name rc=$?"
}
run ka SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run kb SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run kc GRBM_GUI_ACTIVE SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_VMEM
run kd TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum
cd $R
for n in ka kb kc kd; do f=$(find gpurun_out/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
tot = collections.Counter(); disp = set()
for r in csv.DictReader(open(sys.argv[1])):
    tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
print(len(disp), "dispatches", {k: round(v / len(disp)) for k, v in tot.items()})
PY
done
