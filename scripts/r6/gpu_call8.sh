#!/bin/bash
mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_adaptive_gpu.py -x -q -m gpu -k "teacher_forced or use_fp16 or adaptive" 2>&1 | tail -6
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/call8_bench_default.json 2> gpurun_out/r6/call8_bench_default.err; tail -3 gpurun_out/r6/call8_bench_default.err
python - <<EOF
import json
d=json.loads(open("gpurun_out/r6/call8_bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","precision_mode","mfma_roofline_frac_whole_step")})
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_ms","whole_step_frac_of_the_strict_mode")})
print("strict", d.get("strict_both_metrics"))
print("c3_sanity", d.get("c3_sanity"))
print("kt", d.get("kernel_time_ms_per_forward"))
print("others", [(m["precision_mode"], m["value"], m.get("within_tolerance")) for m in d.get("other_modes",[])])
print("parity", {k:v for k,v in d["parity"].items() if not isinstance(v,dict)})
EOF
