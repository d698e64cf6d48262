#!/bin/bash
# the whole GPU suite at the current sources -> gpurun_out/r6/gpu_tests.log
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6/gpu_tests.log
cat gpurun_out/r6/gpu_tests.log
