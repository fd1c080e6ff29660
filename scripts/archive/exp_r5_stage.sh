#!/bin/bash
# Round 5, first GPU call: smoke + the parity suite on the staged k_select / k_plan_rescue, then the A/B of the stage sizes (library variants s0 / s4 / s8
# built with ISAAC_GPU_BUILD_TAG), one context
python __graft_entry__.py smoke > gpurun_out/r5a_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r5a_smoke.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5a_gputests.log
VARIANTS="default s0 s4 s8" KEYS="select plan_rescue sums_wave rescue_gapped_plan" STEPS=4 bash scripts/exp_variants.sh > gpurun_out/r5a_exp_stage.log 2>&1
tail -2 gpurun_out/r5a_smoke.log; cat gpurun_out/r5a_gputests.log; cat gpurun_out/r5a_exp_stage.log
