#!/bin/bash
# Round-5 third GPU call: the Winograd core-loop microbenchmark, the config-4 ragged-plan test, the other models and the
# end-to-end configs in the new default mode.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 scripts/micro/wino_core 2>&1 | tee gpurun_out/r5_wino_core.txt
timeout 600 python -m pytest "tests/test_pipeline_gpu.py::test_config4_rank_shard_at_full_size_batch_32_then_the_ragged_batch_of_2" -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -5
for m in small sr256; do
  timeout 600 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline $( [ $m = sr256 ] && echo --batch 16 ) > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$m.json").read().strip().splitlines()[-1])
    print("$m", d["precision_mode"], d["value"], d["ms_per_step"], d["mfma_roofline_frac_whole_step"], d.get("forward_rel_l2_max_over_set"), d.get("headline_selection", {}).get("within_tolerance"), [(o["precision_mode"], o["value"], o.get("within_tolerance")) for o in d.get("other_modes", [])])
except Exception as e:
    print("$m failed", e); print(open("gpurun_out/bench_$m.err").read()[-1500:])
PY
done
timeout 900 python bench.py --config c3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
echo "c3 exit $?"; head -c 300 gpurun_out/bench_c3.json; echo
timeout 1500 python bench.py --config c5 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
echo "c5 exit $?"; head -c 300 gpurun_out/bench_c5.json; echo
python - <<'PY'
import json
for c in ("c3", "c5"):
    try:
        d = json.loads(open("gpurun_out/bench_%s.json" % c).read().strip().splitlines()[-1])
        print(c, d["value"], d.get("seconds_per_batch"), d.get("sr_seconds_per_batch"), d.get("config4_samples_per_s_same_run"), d.get("unet_forward_ms"), d.get("precision_mode"))
    except Exception as e:
        print(c, "failed", e); print(open("gpurun_out/bench_%s.err" % c).read()[-1500:])
PY
