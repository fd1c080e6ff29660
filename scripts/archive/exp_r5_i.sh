#!/bin/bash
# Round 5, ninth GPU call: the headline configuration at its real size -- 400 M pairs of 2x150 (30x of a 3.1 Gbp genome) through bin/isaac-align, sampled tiles against the oracle;
# then the same files with the selections started while lanes are still read
timeout 3300 python scripts/cli_headline.py --pairs 400000000 --lanes 8 --devices 0,0 --extra-env "ISAAC_ALIGN_STREAM_SELECTION=1" --out gpurun_out/r5_cli_headline_400M.json > gpurun_out/r5_cli_headline_400M.log 2>&1
echo rc $?
tail -c 6000 gpurun_out/r5_cli_headline_400M.log
