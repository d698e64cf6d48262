#!/bin/bash
# Round-6 evidence at the current sources: smoke(), the default bench line (the driver's command), the same workload over the whole
# 50-step schedule, PMC passes + rocprofv3 kernel statistics of the headline ladder, per-launch tables of its tiers, the other models.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6p
mkdir -p $O
export TMPDIR=/tmp
make -C oracle -s
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 | tee $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "bench exit $?"; tail -2 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6p/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic", "avg_launch_ms")}, d["roofline"].get("mfma_util"))
print("strict", d.get("strict_both_metrics"))
print("c3_sanity", d.get("c3_sanity"))
print("cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "kind")})
for m in d.get("other_modes", []) + [d.get("parity_mode", {})]:
    print(m.get("precision_mode"), m.get("value"), m.get("ms_per_step"), m.get("within_tolerance"))
PY
timeout 600 python bench.py --precision fp16sa3 --steps 50 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-breakdown --no-c3-sanity > $O/bench_fp16sa3_50steps.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_fp16sa3_50steps.json').read().strip().splitlines()[-1]); print('50 steps (the whole schedule):', d['value'], d['ms_per_step'], d['adaptive']['tiers'])"
for tier in fp16s fp16cs fp16cx; do
  IVID_BENCH_LAYERS=$O/layers_$tier.json timeout 300 python bench.py --precision $tier --steps 3 --warmup 2 --no-cpu-baseline --no-parity-mode --no-c3-sanity > /dev/null 2>&1
done
ls -la $O | head -20
if [ -z "$SKIP_PMC" ]; then
D=$O/pmc; mkdir -p $D
run() { name=$1; shift; rm -rf $D/$name
  IVID_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D/$name -o p -- \
    python bench.py --precision fp16sa3 --steps 10 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode --no-c3-sanity > $D/$name.log 2>&1
  echo "$name exit $?"; }
run SQ SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
python scripts/r4/pmc_mfma_summary.py $D fp16sa3 | sed "s/--steps 1 --warmup 1/--steps 10 --warmup 1/" > $O/pmc_mfma_fp16sa3.json && head -40 $O/pmc_mfma_fp16sa3.json
python scripts/pmc_traffic.py $D fp16sa3 | sed "s/--steps 1 --warmup 1/--steps 10 --warmup 1/" > $O/pmc_traffic_fp16sa3.json && head -c 1200 $O/pmc_traffic_fp16sa3.json
find $D -name "*.csv" -size +3M -delete
fi
rm -rf $O/stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python bench.py --precision fp16sa3 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode --no-c3-sanity > $O/bench_profiled_fp16sa3.json 2> $O/stats.log
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_fp16sa3.csv && head -6 $O/kernel_stats_fp16sa3.csv | cut -c1-170
find $O/stats -name "*.csv" -size +3M -delete
for m in small sr256; do
  timeout 600 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline $( [ $m = sr256 ] && echo --batch 16 ) > $O/bench_$m.json 2> $O/bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$m.json").read().strip().splitlines()[-1])
    print("$m", d["precision_mode"], d["value"], d["ms_per_step"], d["mfma_roofline_frac_whole_step"], d.get("forward_rel_l2_max_over_set"), d.get("strict_both_metrics", {}).get("value"))
except Exception as e:
    print("$m failed", e); print(open("$O/bench_$m.err").read()[-1500:])
PY
done
