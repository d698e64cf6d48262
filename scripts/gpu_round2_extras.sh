#!/bin/bash
# Round-2 evidence that is not part of the default bench line: clean rocprof kernel stats + PMC traffic of the bf16 headline,
# per-kernel breakdown of the parity mode (bf16x3), BASELINE config 3 end to end, the small-128 / SR-256 models.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
echo "prof exit $?"; head -12 gpurun_out/prof/bench_kernel_stats.csv | cut -c1-200
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
bash scripts/gpu_pmc_bench.sh | tail -8
IVID_BENCH_LAYERS=gpurun_out/layers_x3.json timeout 600 python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > gpurun_out/bench_x3.json 2> gpurun_out/bench_x3.err
echo "x3 exit $?"; python -c "
import json; r=json.load(open('gpurun_out/bench_x3.json')); print(r['value'], r['ms_per_step'], r['kernel_time_ms_per_forward'])"
timeout 900 python bench.py --config c3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
echo "c3 exit $?"; cat gpurun_out/bench_c3.json; tail -3 gpurun_out/bench_c3.err
