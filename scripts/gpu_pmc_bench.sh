#!/bin/bash
# HBM traffic of the conv kernel family in the BENCH workload: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters
# only + --kernel-trace, one counter per pass as MI355X_MICROARCH.md prescribes) over `python bench.py --precision $PREC`,
# summarised to gpurun_out/pmc_bench_$PREC/traffic.json (copy to profiles/r03_pmc_traffic_$PREC.json).
#   IVID_COMMIT=$(git rev-parse --short HEAD) PREC=fp16c bash scripts/gpu_pmc_bench.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
PREC=${PREC:-fp16c}
D=gpurun_out/pmc_bench_$PREC
mkdir -p $D
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $D/$c
  IVID_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$c -o p -- \
    python bench.py --precision $PREC --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > $D/$c.log 2>&1
  echo "$c exit $?"
done
python scripts/pmc_traffic.py $D $PREC > $D/traffic.json && cat $D/traffic.json
find $D -name "*.csv" -size +5M -delete
