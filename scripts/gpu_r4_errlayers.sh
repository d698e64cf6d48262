#!/bin/bash
# Round 4: per-site error budget of the fp16 roundings on the representative forward set (torch on the GPU; test tooling).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python tests/tools/error_layers.py --model large --out gpurun_out/error_layers_large.json > gpurun_out/error_layers_large.log 2>&1
tail -3 gpurun_out/error_layers_large.log
timeout 600 python tests/tools/error_layers.py --model small --out gpurun_out/error_layers_small.json > gpurun_out/error_layers_small.log 2>&1
tail -3 gpurun_out/error_layers_small.log
