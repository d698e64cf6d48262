#!/bin/bash
# generic A/B runner: OUT=<name> FORMS=... LAYERS=... bash scripts/r6/gpu_ab.sh lib1.so lib2.so ...
mkdir -p gpurun_out/r6
python scripts/r6/fused_ab.py "$@" > gpurun_out/r6/${OUT:-ab}.jsonl 2>&1
python - <<EOF
import json
for l in open("gpurun_out/r6/${OUT:-ab}.jsonl"):
    try: r = json.loads(l)
    except Exception: print(l.rstrip()[:300]); continue
    if "error" in r: print(r); continue
    print(f'{r["lib"][8:-3]:12s} {r["layer"]:9s} {r["form"]:8s} ms {r["ms"]:.4f} TF {r["tflops"]:7.1f}  step {r.get("us_per_kstep")} pro {r.get("prologue_us")} epi {r.get("epilogue_us")} wg {r.get("wg_us")}  sum {r["sum"]:.6g} {r["stats_sum"]:.6g}')
EOF
