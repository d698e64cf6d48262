#!/bin/bash
# A/B on ONE box: bench.py (fp16s headline only) for each library in LIBS ("-" = the product lib), REPS interleaved repetitions.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in ${REPS:-1 2}; do
for lib in ${LIBS:-- ab/libivid_scalarxf.so}; do
  tag=$(basename "$lib" .so)
  [ "$lib" = "-" ] && unset IVID_HIP_LIB || export IVID_HIP_LIB="$PWD/$lib"
  timeout 600 python bench.py --precision ${PREC:-fp16s} --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-parity-mode ${BENCH_ARGS:-} > gpurun_out/ab_${tag}.json 2> gpurun_out/ab_${tag}.err
  python - "$tag" "$rep" <<'PY'
import json, sys
try:
    r = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    k = r.get("kernel_time_ms_per_forward", {})
    print(sys.argv[1], "rep", sys.argv[2], r["value"], r["ms_per_step"], "fused", k.get("conv3x3_fused_kernel"), "igemm", k.get("conv_igemm_kernel"))
except Exception as e:
    print(sys.argv[1], "failed", e); print(open("gpurun_out/ab_%s.err" % sys.argv[1]).read()[-800:])
PY
done
done
