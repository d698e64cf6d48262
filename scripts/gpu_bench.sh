#!/bin/bash
# Run on the GPU box through gpurun: bench line + rocprofv3 kernel stats of the same command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
IVID_BENCH_LAYERS=gpurun_out/layers.json timeout 900 python bench.py --steps ${STEPS:-10} --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ "${PROFILE:-1}" = "1" ]; then
  rm -rf gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
  echo "prof exit $?"; tail -3 gpurun_out/prof.err
  find gpurun_out/prof -name "*stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep only the summaries (traces are large)
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
