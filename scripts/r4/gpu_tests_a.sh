#!/bin/bash
# Round 4: new op / unet tests + default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 1500 python -m pytest tests/test_comp_gpu.py tests/test_unet_gpu.py -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/comp_unet.log
tail -25 gpurun_out/comp_unet.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"; tail -c 3000 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference")})
print(json.dumps(d["headline_selection"]["within_tolerance"]))
print(json.dumps(d["parity"]))
for m in d.get("other_modes", []) + [d.get("parity_mode", {})]:
    print(m.get("precision_mode"), m.get("value"), m.get("ms_per_step"), m.get("within_tolerance"), json.dumps(m.get("parity")))
PY
