#!/bin/bash
# FETCH_SIZE against known byte counts for the library's access patterns (scripts/probes/fetch_calib.hip): gpurun_out/fetch_calib.log
R=$PWD
hipcc -O3 --offload-arch=gfx950 -o /tmp/fetch_calib scripts/probes/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetch_calib -- /tmp/fetch_calib > $R/gpurun_out/fetch_calib.out 2>&1
cd $R
python3 - <<'PY' > gpurun_out/fetch_calib.log
import csv, glob, re
expected = {"k_stream16": 8 << 30, "k_record64": 64 << 26, "k_line16": 64 << 26, "k_gather8": 64 << 26}
print(open("gpurun_out/fetch_calib.out").read().strip().splitlines()[-1])
for f in glob.glob("gpurun_out/fetch_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if m and r["Counter_Name"] == "FETCH_SIZE":
            kb = float(r["Counter_Value"]); e = expected[m.group(1)]
            print("%-12s FETCH_SIZE %12.0f KB = %6.3f x the 64-byte lines touched (%6.3f x with the guide's doubling)" % (m.group(1), kb, kb * 1024 / e, 2 * kb * 1024 / e))
PY
cat gpurun_out/fetch_calib.log
rm -rf gpurun_out/fetch_calib
