#!/bin/bash
# Round-5 closing validation on the GPU box: the full parity suite, smoke(), the default bench line (what the driver runs), the same
# workload over the WHOLE 50-step schedule, then the PMC passes and rocprofv3 kernel statistics of the headline ladder.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/tests_full.log
tail -12 gpurun_out/tests_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -11
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"; tail -2 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic", "avg_launch_ms")}, d["roofline"].get("mfma_util"))
print("parity", {k: v for k, v in d["parity"].items() if not isinstance(v, dict)})
print("strict", d.get("strict_both_metrics"))
for m in d.get("other_modes", []) + [d.get("parity_mode", {})]:
    print(m.get("precision_mode"), m.get("value"), m.get("ms_per_step"), m.get("within_tolerance"))
PY
timeout 600 python bench.py --precision fp16sa3 --steps 50 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-breakdown > gpurun_out/bench_fp16sa3_50steps.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/bench_fp16sa3_50steps.json').read().strip().splitlines()[-1]); print('50 steps (the whole schedule):', d['value'], d['ms_per_step'], d['adaptive']['tiers'])"
# (config 4 and config 5 end to end were run by separate calls of `python bench.py --config c4|c5`: profiles/r05_bench_c4_fp16sa.json,
#  r05_bench_c5_fp16sa_sr_bf16.json)
# PMC passes + kernel statistics of the headline ladder at the closing sources (profiles/r05_pmc_*_fp16sa3.json, r05_kernel_stats_fp16sa3.csv)
IVID_COMMIT=${IVID_COMMIT:-unknown} PREC=fp16sa3 bash scripts/r5/gpu_pmc.sh > gpurun_out/pmc_r5.log 2>&1; tail -3 gpurun_out/pmc_r5.log
rm -rf gpurun_out/stats_fp16sa3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_fp16sa3 -o p -- python bench.py --precision fp16sa3 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/bench_profiled_fp16sa3.json 2> gpurun_out/stats_fp16sa3.log
f=$(find gpurun_out/stats_fp16sa3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_fp16sa3.csv && head -4 gpurun_out/kernel_stats_fp16sa3.csv | cut -c1-160
find gpurun_out/stats_fp16sa3 -name "*.csv" -size +3M -delete
