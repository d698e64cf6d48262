#!/bin/bash
# Development aid: ablation timings of the fused conv kernel (library must be built with -DIVID_DEV_ABLATE).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for ab in ${ABS:-0 1 2 3 4 7 8 15}; do
  echo "== ablate $ab"
  IVID_FUSED_ABLATE=$ab CFGS=-1 SHAPES_ONLY=${SHAPES_ONLY:-0,1,6} REPS=10 timeout 120 python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/ablate.log
