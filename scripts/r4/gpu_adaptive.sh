#!/bin/bash
# The adaptive precision mode (opt-in "fp16sa"): its GPU tests (every forward-set row in the mode its timestep selects, incl. the
# mid-t sets) and the bench line with fp16s / fp16cs timed beside it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
timeout 700 python -m pytest tests/test_adaptive_gpu.py -q -s -p no:cacheprovider > gpurun_out/tests_adaptive.log 2>&1; echo "tests exit $?"
grep -E "forward set|passed|failed|Error|assert" gpurun_out/tests_adaptive.log | head -30
cp gpurun_out/parity_report.json gpurun_out/parity_report_adaptive.json 2>/dev/null
timeout 500 python bench.py --precision fp16sa --extra-precisions fp16s,fp16cs --no-cpu-baseline --steps 20 > gpurun_out/bench_fp16sa.json 2> gpurun_out/bench_fp16sa.err; echo "bench exit $?"; tail -3 gpurun_out/bench_fp16sa.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_fp16sa.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step", "adaptive")})
print(d.get("parity"))
print([(m["precision_mode"], m["value"], m["ms_per_step"], m["within_tolerance"], m["parity"]["fwd_set_max"]) for m in d["other_modes"]], d["parity_mode"]["value"])
PY
