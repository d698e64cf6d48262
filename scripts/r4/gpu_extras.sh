#!/bin/bash
# Round-4 extras: BASELINE config 4 end to end (one batch), rocprofv3 kernel statistics of the small-128 / SR-256 benches in the headline mode.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in small sr256; do
  rm -rf gpurun_out/stats_$m
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$m -o p -- python bench.py --model $m --precision fp16s --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode $( [ $m = sr256 ] && echo --batch 16 ) > gpurun_out/bench_profiled_$m.json 2> gpurun_out/stats_$m.log
  f=$(find gpurun_out/stats_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_${m}_fp16s.csv && head -5 gpurun_out/kernel_stats_${m}_fp16s.csv | cut -c1-150
  find gpurun_out/stats_$m -name "*.csv" -size +5M -delete
done
timeout 1200 python bench.py --config c4 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err
echo "c4 exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_c4.json").read().strip().splitlines()[-1])
print("c4", d["precision_mode"], d["value"], d["seconds_per_batch"], d.get("unet_forward_ms"), d["precision_selection"]["picked"])
PY
