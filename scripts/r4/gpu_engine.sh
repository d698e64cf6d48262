#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_cpu.py -x -q > gpurun_out/engine_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/engine_tests.log
timeout 600 python scripts/r4/engine_fullsize.py 8 > gpurun_out/engine_fullsize.json 2> gpurun_out/engine_fullsize.err; echo "fullsize exit $?"; cat gpurun_out/engine_fullsize.json; tail -3 gpurun_out/engine_fullsize.err
