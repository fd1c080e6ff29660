#!/bin/bash
# Round 5, eleventh GPU call: the spill files' scenario with the rest of test_cli; what disks the box has (the container may use 300 GiB of memory, /dev/shm included:
# the 400 M-pair attempt died of its 264 GB of FASTQ text there); the 100 M-pair headline run again with the round's last program, and once more with streamed selections
df -h / /tmp /dev/shm 2>/dev/null; ( cd /tmp && timeout 60 dd if=/dev/zero of=/tmp/_probe bs=1M count=4096 oflag=direct 2>&1 | tail -1; rm -f /tmp/_probe )
timeout 1500 python -m pytest tests/test_cli.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5k_gputests.log
cat gpurun_out/r5k_gputests.log
timeout 2400 python scripts/cli_headline.py --pairs 100000000 --lanes 4 --devices 0,0 --extra-env "ISAAC_ALIGN_STREAM_SELECTION=1" --out gpurun_out/r5_cli_headline_100M_b.json > gpurun_out/r5_cli_headline_100M_b.log 2>&1
echo rc $?
tail -c 5000 gpurun_out/r5_cli_headline_100M_b.log
