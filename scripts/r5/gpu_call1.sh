#!/bin/bash
# Round-5 first GPU call: the tiered adaptive modes + the new synthetic checkpoints.  (a) the adaptive / engine / use_fp16 tests,
# (b) the per-timestep deviation table of the mode ladder on the new checkpoints and (fp16c / fp16cx) on the mid-t sets: what the
# tier thresholds are read from, (c) the default bench line (headline rule over fp16sa3 / fp16sa / fp16s), (d) per-launch tables.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.json
timeout 900 python -m pytest tests/test_adaptive_gpu.py tests/test_engine_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/r5_tests_a.log
tail -15 gpurun_out/r5_tests_a.log
timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "use_fp16 or out_of_range or empty" 2>&1 | tail -5
timeout 900 python scripts/r5/fwd_set_modes.py gpurun_out/r5_modes_seeds.json large128_s11,large128_tr12,small128_s21,small128_s22,small128_s23,small128_tr24 fp16,fp16c,fp16cx,fp16cs,fp16s 2>&1 | tail -40
timeout 900 python scripts/r5/fwd_set_modes.py gpurun_out/r5_modes_main.json large128,large128_mid,small128,small128_mid,largecond128,largecond128_mid,sr256,sr256_mid fp16c,fp16cx,fp16cs,fp16s 2>&1 | tail -40
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step")})
print("selection", d.get("headline_selection", {}).get("within_tolerance"))
print("parity", d.get("parity"))
print("adaptive", d.get("adaptive"))
print("roofline frac", d["roofline"]["frac"], d["roofline"]["achieved"], "kernel ms", d.get("kernel_time_ms_per_forward"))
for m in d.get("other_modes", []) + [d.get("parity_mode", {})]:
    print(m.get("precision_mode"), m.get("value"), m.get("ms_per_step"), m.get("within_tolerance"), m.get("parity", {}).get("fwd_set_max"), m.get("parity", {}).get("other_checkpoints_max"))
PY
for p in fp16s fp16cs fp16cx; do
  IVID_BENCH_LAYERS=gpurun_out/r5_layers_$p.json timeout 300 python bench.py --precision $p --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > gpurun_out/r5_bench_$p.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/r5_bench_$p.json').read().strip().splitlines()[-1]); print('$p', d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_forward'))"
done
