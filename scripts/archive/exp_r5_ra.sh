#!/bin/bash
# Round 5: k_rescue_align held to six waves per SIMD (80 registers) now that its candidate lives in registers (87 registers, five waves)
VARIANTS="default ra6 default ra6" KEYS="rescue_align align_candidates" STEPS=6 bash scripts/exp_variants.sh 2>&1 | tee gpurun_out/exp_r5_rescue_align_waves.log
