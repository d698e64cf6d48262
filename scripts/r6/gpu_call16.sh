#!/bin/bash
# igemm epilogue forms: per-layer timing on the model's residual launches (layer harness) + conv parity tests against the new build
mkdir -p gpurun_out/r6
{
for lib in ab/libivid_x0.so ab/libivid_x1.so ab/libivid_x0.so ab/libivid_x1.so; do
  echo "== $lib"
  IVID_HIP_LIB=$PWD/$lib python scripts/r6/igemm_res_bench.py
done
echo "== parity tests with x1"
IVID_HIP_LIB=$PWD/ab/libivid_x1.so timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv" 2>&1 | tail -5
} > gpurun_out/r6/call16_igemm_epi.log 2>&1
