#!/bin/bash
# Final re-check at HEAD: the whole GPU suite, smoke(), the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step")})
print("roofline", d["roofline"]["frac"], d["roofline"]["mfma_util"]["mfma_busy_frac"], d["roofline"]["traffic_source"], d["roofline"]["mfma_util"]["source"])
print([(m["precision_mode"], m["value"], m["within_tolerance"]) for m in d["other_modes"]], d["parity_mode"]["value"])
PY
