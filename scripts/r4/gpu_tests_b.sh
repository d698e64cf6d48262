#!/bin/bash
# Round 4: chain tests in the 16-bit modes, gn_apply_p, pooled-residual A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
make -C oracle -s
timeout 1500 python -m pytest tests/test_comp_gpu.py tests/test_unet_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "gn_apply_p or 16bit_modes or scene_fixture or down or forward_set or mini_forward" 2>&1 | tail -15 > gpurun_out/tests_b.log
tail -15 gpurun_out/tests_b.log
for e in 0 1; do
  IVID_NO_POOL_RES=$e IVID_BENCH_LAYERS=gpurun_out/layers_pool$e.json timeout 600 python bench.py --precision fp16s --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > gpurun_out/bench_pool$e.json 2> gpurun_out/bench_pool$e.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_pool$e.json").read().strip().splitlines()[-1])
    print("IVID_NO_POOL_RES=$e", d["value"], d["ms_per_step"], d.get("kernel_time_ms_per_forward"))
except Exception as ex:
    print("failed", ex); print(open("gpurun_out/bench_pool$e.err").read()[-1500:])
PY
done
