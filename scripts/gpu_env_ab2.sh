#!/bin/bash
# Same-box A/B of plan / kernel switches: every VARIANT is "tag:ENV=VAL,ENV=VAL" (empty env list = the product defaults);
# optional test selections first.  Prints the bench headline + per-kernel times per variant and model.
#   OPS_K / UNET_K : pytest -k expressions for tests/test_ops_gpu.py / tests/test_unet_gpu.py ("" = skip)
#   VARIANTS       : e.g. "base:IVID_NO_IGEMM_SKIP=1 new:"
#   MODELS         : large small sr256
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle -s
if [ -n "${OPS_K}" ]; then
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "${OPS_K}" 2>&1 | tail -25 > gpurun_out/ab_ops.log
  tail -8 gpurun_out/ab_ops.log
fi
if [ -n "${UNET_K}" ]; then
  timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "${UNET_K}" 2>&1 | tail -25 > gpurun_out/ab_unet.log
  tail -8 gpurun_out/ab_unet.log
fi
for m in ${MODELS:-large}; do
  b=64; [ "$m" = "sr256" ] && b=16
  for v in ${VARIANTS:-new:}; do
    tag=${v%%:*}; envs=${v#*:}
    (
      IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
      IVID_BENCH_LAYERS=gpurun_out/layers_${m}_${tag}.json timeout 400 python bench.py --model $m --batch $b --steps ${STEPS:-10} --warmup 2 \
        --no-cpu-baseline --no-parity-mode ${BENCH_ARGS:-} > gpurun_out/ab_${m}_${tag}.json 2> gpurun_out/ab_${m}_${tag}.err
      echo "== $m $tag exit $?"
    )
    python - "$m" "$tag" <<'PY'
import json, sys
try:
    r = json.load(open("gpurun_out/ab_%s_%s.json" % (sys.argv[1], sys.argv[2])))
    print(r["value"], r["ms_per_step"], r["mfma_roofline_frac_whole_step"], r.get("kernel_time_ms_per_forward"))
except Exception as e:
    print("no result:", e); print(open("gpurun_out/ab_%s_%s.err" % (sys.argv[1], sys.argv[2])).read()[-1500:])
PY
  done
done
