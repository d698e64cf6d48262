#!/bin/bash
# Round 3: compensated-storage op tests (incl. the 128-wide fused kernel) + small / SR model timings per precision mode.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
make -C oracle -s
timeout 900 python -m pytest tests/test_comp_gpu.py tests/test_unet_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/comp_unet.log
tail -5 gpurun_out/comp_unet.log
for m in small sr256; do for p in fp16c fp16; do
  b=64; [ $m = sr256 ] && b=16
  timeout 600 python bench.py --model $m --batch $b --precision $p --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > gpurun_out/bench_${m}_$p.json 2> gpurun_out/bench_${m}_$p.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_${m}_$p.json").read().strip().splitlines()[-1])
    print("$m $p", d["value"], d["ms_per_step"], d["mfma_roofline_frac_whole_step"], d["kernel_time_ms_per_forward"])
except Exception as e:
    print("$m $p failed", e); print(open("gpurun_out/bench_${m}_$p.err").read()[-1200:])
PY
done; done
