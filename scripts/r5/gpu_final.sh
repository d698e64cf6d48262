#!/bin/bash
# Round-5 closing validation on the GPU box: the full parity suite, smoke(), the default bench line (what the driver runs), the same
# workload over the WHOLE 50-step schedule, config 4 end to end, and a shortened config-5 run (code path of the SR deviation record).
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
timeout 900 python bench.py --config c4 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err
echo "c4 exit $?"; python -c "
import json
d=json.loads(open('gpurun_out/bench_c4.json').read().strip().splitlines()[-1]); print('c4', d['value'], d['seconds_per_batch'], d['precision_mode'])"
timeout 900 python bench.py --config c5 --c3-steps-uncond 20 --c3-steps-cond 4 > gpurun_out/bench_c5_short.json 2> gpurun_out/bench_c5_short.err
echo "c5 (shortened) exit $?"; python -c "
import json
d=json.loads(open('gpurun_out/bench_c5_short.json').read().strip().splitlines()[-1]); print('c5 short', d['value'], d.get('sr_forward_set_deviation'))" || tail -5 gpurun_out/bench_c5_short.err
