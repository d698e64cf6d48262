#!/bin/bash
# Round-end validation on the GPU box: full parity suite, smoke(), the default bench line (+ rocprofv3 kernel stats of the
# same command), the PMC traffic passes, the other models, and (FULL=1) BASELINE configs 3 and 5 end to end.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash scripts/gpu_tests.sh
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5
bash scripts/gpu_bench.sh
bash scripts/gpu_pmc_bench.sh | tail -12
bash scripts/gpu_models.sh
if [ "${FULL:-0}" = "1" ]; then
  timeout 900 python bench.py --config c3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
  echo "c3 exit $?"; cat gpurun_out/bench_c3.json; tail -3 gpurun_out/bench_c3.err
  timeout 1200 python bench.py --config c5 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
  echo "c5 exit $?"; cat gpurun_out/bench_c5.json; tail -3 gpurun_out/bench_c5.err
fi
