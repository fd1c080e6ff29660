#!/bin/bash
# Round 5, second GPU call: the parity suite with the register-resident clippers, the general bins (tests/test_bam.py, tests/test_cli.py), then k_select's sections
# (library variants noclip / notemplate leave a section out: timing only) against the product and against the unstaged form (s0)
python __graft_entry__.py smoke > gpurun_out/r5b_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r5b_smoke.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bam.py tests/test_cli.py tests/test_gpu_boundary.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r5b_gputests.log
VARIANTS="default s0 noclip notemplate" KEYS="select plan_rescue sums_wave rescue_gapped_plan" STEPS=4 bash scripts/exp_variants.sh > gpurun_out/r5b_exp_select.log 2>&1
tail -2 gpurun_out/r5b_smoke.log; cat gpurun_out/r5b_gputests.log; cat gpurun_out/r5b_exp_select.log
