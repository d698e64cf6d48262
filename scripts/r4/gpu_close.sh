#!/bin/bash
# Round close at HEAD (after the engine-file loader went into csrc/program.hip): the whole GPU suite, the PMC passes of the
# headline mode (the profiles carry the hash of csrc/), smoke(), the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/tests_close.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/tests_close.log
IVID_COMMIT=$1 PREC=fp16s bash scripts/r4/gpu_pmc_mfma.sh > gpurun_out/pmc_close.log 2>&1; echo "pmc exit $?"; grep -c . gpurun_out/pmc_close.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cp gpurun_out/pmc_r4_fp16s/mfma.json profiles/r04_pmc_mfma_fp16s.json; cp gpurun_out/pmc_r4_fp16s/traffic.json profiles/r04_pmc_traffic_fp16s.json
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step")})
print("roofline", d["roofline"]["frac"], d["roofline"]["mfma_util"]["mfma_busy_frac"], d["roofline"]["traffic_source"], d["roofline"]["mfma_util"]["source"])
print([(m["precision_mode"], m["value"], m["within_tolerance"]) for m in d["other_modes"]], d["parity_mode"]["value"])
PY
