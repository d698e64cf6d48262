#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
make -C oracle -s
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "forward_set" 2>&1 | tail -3
timeout 600 python bench.py --model sr256 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_sr256.json 2> gpurun_out/bench_sr256.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_sr256.json").read().strip().splitlines()[-1])
print("sr256", d["precision_mode"], d["value"], d["ms_per_step"], d["mfma_roofline_frac_whole_step"], d.get("forward_rel_l2_max_over_set"), d["headline_selection"]["within_tolerance"])
PY
timeout 900 python bench.py --config c3 --c3-steps-uncond 50 --c3-steps-cond 10 > gpurun_out/bench_c3_short.json 2> gpurun_out/bench_c3_short.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_c3_short.json").read().strip().splitlines()[-1])
print("c3 short", d["precision_mode"], d["value"], json.dumps(d["precision_selection"])[:700])
PY
tail -3 gpurun_out/bench_c3_short.err
