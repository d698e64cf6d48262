#!/bin/bash
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "groupnorm or gn_" 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6/call11_bench.json 2> gpurun_out/r6/call11_bench.err; tail -2 gpurun_out/r6/call11_bench.err
python - <<EOF
import json
d=json.loads(open("gpurun_out/r6/call11_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","precision_mode","mfma_roofline_frac_whole_step")})
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_ms")})
print("strict", d.get("strict_both_metrics",{}).get("value"))
print("kt", d.get("kernel_time_ms_per_forward"))
print("others", [(m["precision_mode"], m["value"]) for m in d.get("other_modes",[])])
EOF
