#!/bin/bash
# Round 3: BASELINE configs 3 and 5 end to end in the headline precision (fp16c), one batch of 32.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for c in ${CONFIGS:-c3 c5}; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  echo "$c exit $?"; tail -c 1800 gpurun_out/bench_$c.json; tail -2 gpurun_out/bench_$c.err
done
