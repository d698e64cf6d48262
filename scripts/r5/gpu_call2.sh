#!/bin/bash
# Round-5 second GPU call: the changed GPU tests (adaptive ladders incl. fp16sx / fp16sa3 and the new checkpoints, max-norm metric,
# teacher-forced strength-3 eps, config-4 ragged plans), the default bench line, its PMC passes and kernel statistics.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.json
make -C oracle -s
timeout 1500 python -m pytest tests/test_adaptive_gpu.py tests/test_unet_gpu.py "tests/test_pipeline_gpu.py::test_config4_rank_shard_at_full_size_batch_32_then_the_ragged_batch_of_2" -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r5_tests_b.log
tail -25 gpurun_out/r5_tests_b.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "chain_rel_l2_vs_reference", "mfma_roofline_frac_whole_step")})
print("selection", d.get("headline_selection", {}).get("within_tolerance"), d.get("headline_selection", {}).get("both_metrics_within_tolerance"))
print("parity", d.get("parity"))
print("strict", d.get("strict_both_metrics"))
print("adaptive", d.get("adaptive"))
print("roofline frac", d["roofline"]["frac"], d["roofline"]["achieved"], "kernel ms", d.get("kernel_time_ms_per_forward"))
print("cpu", d.get("cpu_baseline"))
for m in d.get("other_modes", []) + [d.get("parity_mode", {})]:
    print(m.get("precision_mode"), m.get("value"), m.get("ms_per_step"), m.get("within_tolerance"), m.get("parity", {}).get("fwd_set_max"), m.get("parity", {}).get("fwd_set_max_rel"), m.get("parity", {}).get("other_checkpoints_max"))
PY
IVID_COMMIT=${IVID_COMMIT:-unknown} PREC=fp16sa3 bash scripts/r5/gpu_pmc.sh > gpurun_out/pmc_r5.log 2>&1; tail -5 gpurun_out/pmc_r5.log
rm -rf gpurun_out/stats_fp16sa3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_fp16sa3 -o p -- python bench.py --precision fp16sa3 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/bench_profiled_fp16sa3.json 2> gpurun_out/stats_fp16sa3.log
f=$(find gpurun_out/stats_fp16sa3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_fp16sa3.csv && head -8 gpurun_out/kernel_stats_fp16sa3.csv | cut -c1-200
find gpurun_out/stats_fp16sa3 -name "*.csv" -size +3M -delete
