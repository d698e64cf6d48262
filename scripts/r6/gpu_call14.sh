#!/bin/bash
# igemm ping-pong experiment: timing A/B on the model's shapes + the conv parity tests against the experimental library
mkdir -p gpurun_out/r6
{
bash scripts/r6/gpu_igemm_ab.sh ab/libivid_x0.so ab/libivid_x1.so ab/libivid_x0.so ab/libivid_x1.so
echo "== parity tests with x1"
IVID_HIP_LIB=$PWD/ab/libivid_x1.so timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv" 2>&1 | tail -5
} > gpurun_out/r6/call14_igemm_pp.log 2>&1
