#!/bin/bash
# Run on the GPU box through gpurun: parity tests with full logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout 180 -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/ops.log
echo "ops exit: $?" >> gpurun_out/ops.log
timeout 1500 python -m pytest tests/test_unet_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/unet.log
echo "unet exit: $?" >> gpurun_out/unet.log
tail -5 gpurun_out/ops.log; tail -5 gpurun_out/unet.log
