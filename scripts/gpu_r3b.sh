#!/bin/bash
# Round 3: full GPU suite + the default bench line (auto headline selection) + per-layer tables of two modes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ${PYTEST_X:--x} 2>&1 | tail -40 > gpurun_out/all_gpu.log
tail -6 gpurun_out/all_gpu.log
if [ "${BENCH:-1}" = "1" ]; then
IVID_BENCH_LAYERS=gpurun_out/layers_fp16c.json timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["precision_mode"], d["dtype"], d["value"], d["ms_per_step"], d["mfma_roofline_frac_whole_step"], d.get("rel_l2_vs_reference"))
print(d["kernel_time_ms_per_forward"])
PY
tail -3 gpurun_out/bench_default.err
fi
