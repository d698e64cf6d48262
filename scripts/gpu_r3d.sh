#!/bin/bash
# Round 3: fp16c with input lo planes in the fused kernels: op tests, UNet tests, bench of fp16c vs fp16 on one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 900 python -m pytest tests/test_comp_gpu.py tests/test_unet_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/comp_unet.log
tail -6 gpurun_out/comp_unet.log
for p in fp16c fp16cx fp16; do
  IVID_BENCH_LAYERS=gpurun_out/layers_$p.json timeout 600 python bench.py --precision $p --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > gpurun_out/bench_$p.json 2> gpurun_out/bench_$p.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$p.json").read().strip().splitlines()[-1])
    print("$p", d["value"], d["ms_per_step"], d["kernel_time_ms_per_forward"])
except Exception as e:
    print("$p failed", e); print(open("gpurun_out/bench_$p.err").read()[-1500:])
PY
done
python - <<'PY'
import json
r = json.load(open("gpurun_out/parity_report.json"))
for k in sorted(r):
    if "fp16c" in k and (k.startswith("unet/") or k.startswith("chain/")): print(k, r[k])
PY
