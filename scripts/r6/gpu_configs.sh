#!/bin/bash
# BASELINE configs 3, 4 and 5 end to end in the use_fp16 default (fp16sx + the guidance tier); lines -> gpurun_out/r6/bench_c{3,4,5}.json
mkdir -p gpurun_out/r6
for c in c3 c4 c5; do
  timeout 1500 python bench.py --config $c > gpurun_out/r6/bench_$c.json 2> gpurun_out/r6/bench_$c.err
  echo "$c exit $?"; head -c 400 gpurun_out/r6/bench_$c.json; echo; tail -2 gpurun_out/r6/bench_$c.err
done
