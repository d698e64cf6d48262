#!/bin/bash
# Bench lines (headline + per-layer tables) of the other BASELINE models: small-128 (configs 1), SR-256 (config 5).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for m in ${MODELS:-small sr256}; do
  b=64; [ "$m" = "sr256" ] && b=${SR_BATCH:-16}
  IVID_BENCH_LAYERS=gpurun_out/layers_${m}.json timeout 900 python bench.py --model $m --batch $b --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/bench_${m}.json 2> gpurun_out/bench_${m}.err
  echo "== $m exit $?"; tail -2 gpurun_out/bench_${m}.err
  python - "$m" <<'PY'
import json, sys
r = json.load(open("gpurun_out/bench_%s.json" % sys.argv[1]))
print(r["value"], r["ms_per_step"], r["mfma_roofline_frac_whole_step"], r.get("kernel_time_ms_per_forward"), r.get("parity_mode"))
PY
done
