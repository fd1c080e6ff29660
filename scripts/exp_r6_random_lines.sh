#!/bin/bash
# the measured ceiling of the seed lookup: random 64-byte lines per second out of a table-sized buffer (scripts/probes/random_lines.hip) -> gpurun_out/exp_r6_random_lines.log
hipcc -O3 --offload-arch=gfx950 -o /tmp/random_lines scripts/probes/random_lines.hip || exit 1
timeout 600 /tmp/random_lines 47 | tee gpurun_out/exp_r6_random_lines.log
