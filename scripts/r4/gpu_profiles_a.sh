#!/bin/bash
# Round 4: PMC (MFMA utilisation + traffic) of the headline mode, small-128 / SR-256 benches in the headline mode, rocprof stats.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
IVID_COMMIT=${IVID_COMMIT:-unknown} PREC=fp16s bash scripts/r4/gpu_pmc_mfma.sh > gpurun_out/pmc_r4.log 2>&1
tail -5 gpurun_out/pmc_r4.log
for m in small sr256; do
  for p in fp16s fp16cx fp16; do
    IVID_BENCH_LAYERS=gpurun_out/layers_${m}_$p.json timeout 600 python bench.py --model $m --precision $p --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode $( [ $m = sr256 ] && echo --batch 16 ) > gpurun_out/bench_${m}_$p.json 2> gpurun_out/bench_${m}_$p.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_${m}_$p.json").read().strip().splitlines()[-1])
    print("$m $p", d["value"], d["ms_per_step"], d["mfma_roofline_frac_whole_step"], d.get("kernel_time_ms_per_forward"))
except Exception as e:
    print("$m $p failed", e); print(open("gpurun_out/bench_${m}_$p.err").read()[-1500:])
PY
  done
done
rm -rf gpurun_out/stats_fp16s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_fp16s -o p -- python bench.py --precision fp16s --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/stats_fp16s.log 2>&1
f=$(find gpurun_out/stats_fp16s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_fp16s.csv && head -12 gpurun_out/kernel_stats_fp16s.csv | cut -c1-160
find gpurun_out/stats_fp16s -name "*.csv" -size +5M -delete
