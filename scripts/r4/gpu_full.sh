#!/bin/bash
# Round 4 validation on the GPU box: full parity suite, smoke(), default bench line (+ rocprofv3 kernel stats of the same command),
# and (FULL=1) BASELINE configs 3 and 5 end to end.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/tests_full.log
tail -12 gpurun_out/tests_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -9
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step")})
print("roofline", json.dumps(d.get("roofline"))[:900])
print("cpu", d.get("cpu_baseline"))
PY
if [ "${FULL:-0}" = "1" ]; then
  timeout 900 python bench.py --config c3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
  echo "c3 exit $?"; head -c 1500 gpurun_out/bench_c3.json; echo; tail -3 gpurun_out/bench_c3.err
  timeout 1500 python bench.py --config c5 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
  echo "c5 exit $?"; head -c 1500 gpurun_out/bench_c5.json; echo; tail -3 gpurun_out/bench_c5.err
fi
