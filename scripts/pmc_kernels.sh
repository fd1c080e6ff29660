#!/bin/bash
# SQ / LDS / cache counter passes for the kernels matching KERNEL (regex), each pass its own rocprofv3 run (no tracing combined with
# --pmc); per-kernel averages per dispatch are written to gpurun_out/pmc_kernels_<TAG>.json
KERNEL=${KERNEL:-"k_rescue_windows|k_gapped_jobs|k_cluster_sums|k_select|k_plan_rescue|k_rescue_gapped_plan|k_rescue_align"}
TAG=${TAG:-r2}
R=$PWD
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rm -rf $R/gpurun_out/pmc_${TAG}_$name
  rocprofv3 --pmc "$@" --kernel-include-regex "$KERNEL" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$name -- python3 $R/${PROGRAM:-bench.py} ${PROGRAM_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-bam-pass} $BENCH_ARGS > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1
  echo "pass $name rc=$?"
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run c GRBM_GUI_ACTIVE SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_VMEM SQ_LDS_IDX_ACTIVE
run d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum
cd $R
python3 - "$TAG" <<'PY'
import csv, sys, collections, glob, json
tag = sys.argv[1]
out = collections.defaultdict(dict)
for n in "abcd":
    for f in glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, n), recursive=True):
        tot = collections.defaultdict(collections.Counter); disp = collections.defaultdict(set); meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0].replace("isaac::", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
            meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_block_bytes": int(r["LDS_Block_Size"]), "scratch_bytes": int(r["Scratch_Size"]), "workgroup": int(r["Workgroup_Size"])}
        for k in tot:
            out[k].update(meta[k]); out[k]["dispatches_" + n] = len(disp[k])
            out[k].update({c: round(v / len(disp[k])) for c, v in tot[k].items()})
json.dump(out, open("gpurun_out/pmc_kernels_%s.json" % tag, "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
