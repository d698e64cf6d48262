#!/bin/bash
# A/B of environment-variable tuning hooks on ONE box: bench.py headline per setting in ENVS ("NAME=VAL" or "-").
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for e in ${ENVS:--}; do
  tag=$(echo "$e" | tr '=' '_')
  ( [ "$e" != "-" ] && export "$e"
    IVID_BENCH_LAYERS=gpurun_out/layers_${tag}.json timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-parity-mode ${BENCH_ARGS:-} > gpurun_out/env_${tag}.json 2> gpurun_out/env_${tag}.err )
  echo "== $tag exit $?"
  python - "$tag" <<'PY'
import json, sys
r = json.load(open("gpurun_out/env_%s.json" % sys.argv[1]))
print(r["value"], r["ms_per_step"], r["mfma_roofline_frac_whole_step"], r.get("kernel_time_ms_per_forward"))
PY
done
