#!/bin/bash
# rocprofv3 kernel stats of the small-128 bench per setting in ENVS (true kernel durations of the tiny kernels: HIP-event
# timing of an eager launch is floored by the launch latency).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for e in ${ENVS:--}; do
  tag=$(echo "$e" | tr '=' '_')
  rm -rf gpurun_out/prof_$tag
  ( [ "$e" != "-" ] && export "$e"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o p -- python bench.py --model ${MODEL:-small} --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err )
  echo "== $tag"; python -c "
import json; r=json.load(open('gpurun_out/prof_$tag.json')); print(r['value'], r['ms_per_step'])"
  grep -i "finalize\|gn_apply\|Name" gpurun_out/prof_$tag/p_kernel_stats.csv | cut -c1-60,200-400 | cut -d, -f1-4 | head -8
  find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
done
