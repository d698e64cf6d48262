#!/bin/bash
# A/B on ONE box: bench.py (headline only) for each library in LIBS (paths relative to the repo root; "-" = the product lib).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in 1 ${REPS:-}; do
for lib in ${LIBS:-- ab/libivid_oldfused.so}; do
  tag=$(basename "$lib" .so)
  [ "$lib" = "-" ] && unset IVID_HIP_LIB || export IVID_HIP_LIB="$PWD/$lib"
  IVID_BENCH_LAYERS=gpurun_out/layers_${tag}.json timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-parity-mode ${BENCH_ARGS:-} > gpurun_out/ab_${tag}.json 2> gpurun_out/ab_${tag}.err
  echo "== $tag rep $rep exit $?"
  python - "$tag" <<'PY'
import json, sys
r = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
print(r["value"], r["ms_per_step"], r["mfma_roofline_frac_whole_step"], r.get("kernel_time_ms_per_forward"))
PY
done
done
