#!/bin/bash
# Round 3 profiles: rocprofv3 --kernel-trace --stats of the bench command in every benched precision mode (no bs-1 launches
# mixed in: --no-parity-mode keeps the golden forwards out), and the two PMC passes of the headline mode.
#   IVID_COMMIT=$(git rev-parse --short HEAD) bash scripts/gpu_r3_profile.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof3
export TMPDIR=/tmp
for p in ${MODES:-fp16c bf16 fp16 bf16x3}; do
  rm -rf gpurun_out/prof3/$p
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3/$p -o bench -- \
    python bench.py --precision $p --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/prof3/bench_$p.json 2> gpurun_out/prof3/$p.err
  echo "prof $p exit $?"
  f=$(find gpurun_out/prof3/$p -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/prof3/kernel_stats_$p.csv && head -8 "$f"
  find gpurun_out/prof3/$p -name "*.csv" -size +2M -delete
done
if [ "${PMC:-1}" = "1" ]; then PREC=${PMC_PREC:-fp16c} bash scripts/gpu_pmc_bench.sh; fi
