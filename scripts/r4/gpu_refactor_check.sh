#!/bin/bash
# Round 4: the one-source fused kernel -- bit identity against the previous commit's library, op / unet tests, perf A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
make -C oracle -s
IVID_HIP_LIB=$PWD/ab/libivid_head.so python scripts/r4/ab_bits.py head 2>&1 | grep -v amdgpu.ids | tail -3
python scripts/r4/ab_bits.py new 2>&1 | grep -v amdgpu.ids | tail -3
python - <<'PY'
import json
a, b = json.load(open("gpurun_out/ab_bits_head.json")), json.load(open("gpurun_out/ab_bits_new.json"))
bad = [k for k in a if a[k] != b.get(k)]
print("bit-identical cases: %d / %d" % (len(a) - len(bad), len(a)), "DIFFER:", bad)
PY
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_comp_gpu.py tests/test_unet_gpu.py -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/tests_refactor.log
tail -6 gpurun_out/tests_refactor.log
for m in large small sr256; do
  timeout 600 python bench.py --model $m --precision fp16s --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode $( [ $m = sr256 ] && echo --batch 16 ) > gpurun_out/bench_rf_$m.json 2> gpurun_out/bench_rf_$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_rf_$m.json").read().strip().splitlines()[-1])
    print("$m fp16s", d["value"], d["ms_per_step"], d.get("kernel_time_ms_per_forward"))
except Exception as ex:
    print("failed", ex); print(open("gpurun_out/bench_rf_$m.err").read()[-1500:])
PY
done
python scripts/r4/fwd_set_modes.py small fp16s bf16x3 2>&1 | grep -v amdgpu.ids
