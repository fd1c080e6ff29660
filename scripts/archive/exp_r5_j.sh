#!/bin/bash
# Round 5, tenth GPU call: what memory the box's container may use (the 400 M-pair attempt lost its box while 264 GB of FASTQ text were written to /dev/shm);
# isaac-align with a loader context per reading thread and read, and the output file's blocks asked for ahead of the writes: tests, then the bench line
for f in /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory.high /sys/fs/cgroup/memory/memory.limit_in_bytes /sys/fs/cgroup/memory.current; do [ -r $f ] && echo "$f: $(cat $f)"; done
grep -E "MemTotal|MemAvailable|Shmem:" /proc/meminfo; ulimit -a | grep -E "memory|file size"
timeout 1500 python -m pytest tests/test_cli.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5j_gputests.log
cat gpurun_out/r5j_gputests.log
python bench.py > gpurun_out/r5j_bench_default.json 2> gpurun_out/r5j_bench_default.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5j_bench_default.json') if l.startswith('{')][-1])
print("default:", d['value'], d['ms_per_step'], d.get('parity_diffs'), d.get('records_sha1'))
c=d['config']['cli_end_to_end']; print("cli:", c.get('reads_per_s'), c.get('reads_per_s_without_reference_load'), c.get('stages_s'), c.get('error'))
PY
