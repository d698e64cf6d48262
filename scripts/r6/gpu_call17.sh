#!/bin/bash
# stem im2col: thread per 16-byte piece vs thread per pixel
mkdir -p gpurun_out/r6
{
for lib in ab/libivid_base.so ab/libivid_x1.so ab/libivid_base.so ab/libivid_x1.so; do
  echo "== $lib"; IVID_HIP_LIB=$PWD/$lib python scripts/r6/stem_bench.py 2>/dev/null
done
echo "== op tests with x1"
IVID_HIP_LIB=$PWD/ab/libivid_x1.so timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "stem or im2col or nchw" 2>&1 | tail -3
} > gpurun_out/r6/call17_stem.log 2>&1
