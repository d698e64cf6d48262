#!/bin/bash
# Run on the GPU box through gpurun: parity tests with full logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
make -C oracle -s
for t in ${TESTS:-ops warp pipeline unet}; do
  timeout 1500 python -m pytest tests/test_${t}_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -${TAIL:-60} > gpurun_out/${t}.log
  echo "$t exit: $?" >> gpurun_out/${t}.log
  tail -4 gpurun_out/${t}.log
done
