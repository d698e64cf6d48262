#!/bin/bash
# HBM traffic of the conv kernel family in the BENCH workload: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters
# only + --kernel-trace) over `python bench.py`, summarised to gpurun_out/pmc_bench/traffic.json (copy to profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc_bench
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_bench/$c
  IVID_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_bench/$c -o p -- \
    python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/pmc_bench/$c.log 2>&1
  echo "$c exit $?"
done
python scripts/pmc_traffic.py gpurun_out/pmc_bench > gpurun_out/pmc_bench/traffic.json && cat gpurun_out/pmc_bench/traffic.json
find gpurun_out/pmc_bench -name "*.csv" -size +5M -delete
