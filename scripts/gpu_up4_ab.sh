#!/bin/bash
# Round 2, second session: validation + A/B of (1) ivid_conv3x3_up (phase-decomposed upsample + conv, 4/9 of the MACs)
# for different source-size thresholds and (2) the two-workgroups-per-CU output head.  One box, same clocks for every variant.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle -s
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider \
  -k "conv3x3_up or test_conv2d or out_head" 2>&1 | tail -15 > gpurun_out/ab_ops.log
echo "ops exit: $?"; tail -6 gpurun_out/ab_ops.log
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider \
  -k "mini or small128_forward or large128_forward or full_size or edge" 2>&1 | tail -15 > gpurun_out/ab_unet.log
echo "unet exit: $?"; tail -6 gpurun_out/ab_unet.log
for up in ${UPS:-0 8 16 32 64}; do
  IVID_UP4_MAX_SIDE=$up IVID_BENCH_LAYERS=gpurun_out/layers_up${up}.json timeout 400 python bench.py --steps ${STEPS:-10} --warmup 2 \
    --no-cpu-baseline --no-parity-mode > gpurun_out/ab_up${up}.json 2> gpurun_out/ab_up${up}.err
  echo "== large up4<=${up} exit $?"
  python - "$up" <<'PY'
import json, sys
r = json.load(open("gpurun_out/ab_up%s.json" % sys.argv[1]))
print(r["value"], r["ms_per_step"], r["mfma_roofline_frac_whole_step"], r.get("kernel_time_ms_per_forward"))
PY
done
for m in small sr256; do
  b=64; [ "$m" = "sr256" ] && b=16
  for up in ${UPS_SMALL:-0 32 128}; do
    IVID_UP4_MAX_SIDE=$up IVID_BENCH_LAYERS=gpurun_out/layers_${m}_up${up}.json timeout 400 python bench.py --model $m --batch $b --steps 5 \
      --warmup 2 --no-cpu-baseline --no-parity-mode > gpurun_out/ab_${m}_up${up}.json 2> gpurun_out/ab_${m}_up${up}.err
    echo "== $m up4<=${up} exit $?"
    python - "$m" "$up" <<'PY'
import json, sys
r = json.load(open("gpurun_out/ab_%s_up%s.json" % (sys.argv[1], sys.argv[2])))
print(r["value"], r["ms_per_step"], r["mfma_roofline_frac_whole_step"], r.get("kernel_time_ms_per_forward"))
PY
  done
done
