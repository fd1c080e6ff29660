#!/bin/bash
# HBM traffic, SQ activity and LDS bank conflicts per kernel: four separate rocprofv3 --pmc passes (never combined with tracing) over a short bench
# run with the default 1 M pairs per launch, reduced by scripts/pmc_summary.py into gpurun_out/pmc_summary.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8      # bench.py sets it for itself, but under rocprofv3 the runtime is up before Python starts
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-bam-pass --no-single-stream-pass --no-cli-pass"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_j_fetch -- $B > $R/gpurun_out/pmc_j_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_j_write -- $B > $R/gpurun_out/pmc_j_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc_j_sq -- $B > $R/gpurun_out/pmc_j_sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_j_lds -- $B > $R/gpurun_out/pmc_j_lds.log 2>&1; echo "lds rc=$?"
cd $R
python3 scripts/pmc_summary.py gpurun_out/pmc_summary.json workload=gpurun_out/pmc_j_sq.log fetch=$(find gpurun_out/pmc_j_fetch -name "*counter_collection.csv" | head -1) write=$(find gpurun_out/pmc_j_write -name "*counter_collection.csv" | head -1) sq=$(find gpurun_out/pmc_j_sq -name "*counter_collection.csv" | head -1) lds=$(find gpurun_out/pmc_j_lds -name "*counter_collection.csv" | head -1)

rm -rf gpurun_out/pmc_j_fetch gpurun_out/pmc_j_write gpurun_out/pmc_j_sq gpurun_out/pmc_j_lds      # the per-dispatch CSVs: tens of MB
