#!/bin/bash
# Round 3, first GPU call: compensated-storage ops, UNet precision modes incl. the config-2 chain, timing of the modes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 900 python -m pytest tests/test_comp_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/comp.log
echo "comp exit: $?" >> gpurun_out/comp.log; tail -5 gpurun_out/comp.log
timeout 1500 python -m pytest tests/test_unet_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -s 2>&1 | tail -60 > gpurun_out/unet.log
echo "unet exit: $?" >> gpurun_out/unet.log; tail -8 gpurun_out/unet.log
for p in fp16c fp16 bf16; do
  IVID_BENCH_LAYERS=gpurun_out/layers_$p.json timeout 600 python bench.py --steps 10 --warmup 3 --precision $p --no-cpu-baseline --no-parity-mode > gpurun_out/bench_$p.json 2> gpurun_out/bench_$p.err
  echo "bench $p exit $?"; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$p.json").read().strip().splitlines()[-1])
    print("$p", d["value"], d["ms_per_step"], d.get("rel_l2_vs_reference"), [(k["kernel"], k["achieved"], round(k["avg_launch_ms"]*k["launches_per_forward"],2)) for k in d.get("roofline_kernels", [])])
except Exception as e:
    print("no line", e); print(open("gpurun_out/bench_$p.err").read()[-1500:])
PY
done
