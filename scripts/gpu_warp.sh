#!/bin/bash
# One box: warp parity tests, the warp micro-benchmark (smooth + white-noise depth) and a short c4 run with warp timers.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
make -C oracle -s
timeout 900 python -m pytest tests/test_warp_gpu.py tests/test_pipeline_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -15
SKIP_PIPELINE=1 timeout 600 python scripts/pipeline_bench.py > gpurun_out/warp_bench.json 2> gpurun_out/warp_bench.err
echo "warp bench exit $?"; cat gpurun_out/warp_bench.json; tail -3 gpurun_out/warp_bench.err
timeout 600 python bench.py --config c4 --c3-steps-uncond 20 --c3-steps-cond 5 > gpurun_out/bench_c4_short.json 2> gpurun_out/bench_c4_short.err
echo "c4 exit $?"; cat gpurun_out/bench_c4_short.json; tail -3 gpurun_out/bench_c4_short.err
