#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
make -C oracle -s
timeout 1500 python -m pytest tests/test_comp_gpu.py tests/test_unet_gpu.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "o16 or fp16s or forward_set or mini_forward or batch_invariant" 2>&1 | tail -8 > gpurun_out/tests_c.log
tail -8 gpurun_out/tests_c.log
for e in 0 1; do
  IVID_NO_ISLAND_O16=$e IVID_BENCH_LAYERS=gpurun_out/layers_o16_$e.json timeout 600 python bench.py --precision fp16s --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > gpurun_out/bench_o16_$e.json 2> gpurun_out/bench_o16_$e.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_o16_$e.json").read().strip().splitlines()[-1])
    print("IVID_NO_ISLAND_O16=$e", d["value"], d["ms_per_step"], d.get("kernel_time_ms_per_forward"))
except Exception as ex:
    print("failed", ex); print(open("gpurun_out/bench_o16_$e.err").read()[-1500:])
PY
done
