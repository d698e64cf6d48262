#!/bin/bash
# PMC counter passes for the conv kernel (separate rocprofv3 runs: counters only, no tracing besides --kernel-trace).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA|GRBM)_[A-Z0-9_a-z]+" | sort -u > gpurun_out/pmc/counters_available.txt
wc -l gpurun_out/pmc/counters_available.txt
run() { # name, counters...
  name=$1; shift
  rm -rf gpurun_out/pmc/$name
  SHAPES_ONLY="${SHAPES_ONLY:-1,3}" CFGS="${CFGS:--1}" REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -o p -- python scripts/conv_bench.py > gpurun_out/pmc/$name.log 2>&1
  echo "$name exit $?"
  f=$(find gpurun_out/pmc/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/pmc_summary.py "$f" > gpurun_out/pmc/$name.summary.txt && cat gpurun_out/pmc/$name.summary.txt
  find gpurun_out/pmc/$name -name "*.csv" -size +5M -delete
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
