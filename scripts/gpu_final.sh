#!/bin/bash
# Round-end validation on the GPU box: full parity suite, smoke(), the default bench line (+ rocprofv3 kernel stats of the
# same command) and BASELINE configs 4/5 end to end (one batch of 32 samples, 27 views each, SR on every view).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash scripts/gpu_tests.sh
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/gpu_bench.sh
timeout 1200 python bench.py --config c5 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
echo "c5 exit $?"; cat gpurun_out/bench_c5.json; tail -3 gpurun_out/bench_c5.err
