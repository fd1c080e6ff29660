#!/bin/bash
# the at-scale GPU tests (tests/test_gpu_scale.py) with their output kept: gpurun_out/scale_<TAG>.log
TAG=${1:-r3}
timeout ${2:-2000} python -m pytest tests/test_gpu_scale.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/scale_$TAG.log
cat gpurun_out/scale_$TAG.log
