#!/bin/bash
# igemm A/B on the model's layer shapes: for lib in "$@": conv_bench with tile auto (cfg 0) ... prints ms per shape
for lib in "$@"; do
  echo "== $lib"
  IVID_HIP_LIB=$PWD/$lib DTYPE=bf16 CFGS=${CFGS:-0} SHAPES_ONLY=${SHAPES_ONLY:-3,7,8,9,10,11,12,13,14,15,16,17} REPS=10 python scripts/conv_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%4d %5d->%-5d taps %d cfg %d  %.4f ms  %7.1f TF' % (r['h'], r['cin'], r['cout'], r['taps'], r['cfg'], r['ms'], r['tflops']))"
done
