#!/bin/bash
# Round-4 validation on the GPU box: full parity suite, smoke(), the default bench line (+ rocprofv3 kernel stats of the same
# command), the PMC passes (MFMA utilisation, traffic), the other models, and (FULL=1) BASELINE configs 3 / 4 / 5 end to end.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/tests_full.log
tail -4 gpurun_out/tests_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step")})
print("roofline frac", d["roofline"]["frac"], d["roofline"]["achieved"], "kernel ms", d.get("kernel_time_ms_per_forward"))
for m in d.get("other_modes", []) + [d.get("parity_mode", {})]:
    print(m.get("precision_mode"), m.get("value"), m.get("ms_per_step"), m.get("within_tolerance"), m.get("parity", {}).get("fwd_set_max"))
PY
IVID_COMMIT=${IVID_COMMIT:-unknown} PREC=fp16s bash scripts/r4/gpu_pmc_mfma.sh > gpurun_out/pmc_r4.log 2>&1; tail -3 gpurun_out/pmc_r4.log
rm -rf gpurun_out/stats_fp16s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_fp16s -o p -- python bench.py --precision fp16s --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/bench_profiled_fp16s.json 2> gpurun_out/stats_fp16s.log
f=$(find gpurun_out/stats_fp16s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_fp16s.csv && head -6 gpurun_out/kernel_stats_fp16s.csv | cut -c1-170
find gpurun_out/stats_fp16s -name "*.csv" -size +5M -delete
IVID_BENCH_LAYERS=gpurun_out/layers_fp16s.json timeout 600 python bench.py --precision fp16s --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > gpurun_out/bench_fp16s_layers.json 2>/dev/null
for m in small sr256; do
  timeout 600 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline $( [ $m = sr256 ] && echo --batch 16 ) > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$m.json").read().strip().splitlines()[-1])
    print("$m", d["precision_mode"], d["value"], d["ms_per_step"], d["mfma_roofline_frac_whole_step"], d.get("forward_rel_l2_max_over_set"), [(o["precision_mode"], o["value"]) for o in d.get("other_modes", [])])
except Exception as e:
    print("$m failed", e); print(open("gpurun_out/bench_$m.err").read()[-1500:])
PY
done
if [ "${FULL:-0}" = "1" ]; then
  timeout 900 python bench.py --config c3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
  echo "c3 exit $?"; head -c 400 gpurun_out/bench_c3.json; echo
  timeout 1500 python bench.py --config c5 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
  echo "c5 exit $?"; head -c 400 gpurun_out/bench_c5.json; echo
  python - <<'PY'
import json
for c in ("c3", "c5"):
    d = json.loads(open("gpurun_out/bench_%s.json" % c).read().strip().splitlines()[-1])
    print(c, d["value"], d["seconds_per_batch"], d.get("sr_seconds_per_batch"), d.get("config4_samples_per_s_same_run"), d.get("unet_forward_ms"))
PY
fi
