#!/bin/bash
# A/B of two library builds on ONE box (box-to-box variance is several %): ab/lib_a.so vs the in-tree library.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
  for v in a b; do
    if [ $v = a ]; then export IVID_HIP_LIB=$PWD/ab/lib_a.so; else unset IVID_HIP_LIB; fi
    echo "== $v (rep $rep)"
    RES=${RES:-0} CFGS=${CFGS:--1,2} SHAPES_ONLY=${SHAPES_ONLY:-0,1,8} REPS=10 python scripts/conv_bench.py 2>&1 | grep -v amdgpu
  done
done
