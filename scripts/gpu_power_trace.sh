#!/bin/bash
# Sample package power and shader clock (rocm-smi) while the bench runs: evidence for the power-limited regime.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( for i in $(seq 1 120); do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power \(W\)|sclk|Max Graphics" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_trace.txt &
SM=$!
python bench.py --steps 200 --warmup 2 --no-cpu-baseline --no-kernel-breakdown > gpurun_out/power_bench.json 2>/dev/null
kill $SM 2>/dev/null
python - <<'PY'
import re
rows = [l for l in open("gpurun_out/power_trace.txt") if "Power" in l]
pw = [float(m.group(1)) for l in rows for m in [re.search(r"Current Socket Graphics Package Power \(W\): ([0-9.]+)", l)] if m]
ck = [int(m.group(1)) for l in rows for m in [re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)] if m]
mx = [m.group(1) for l in rows for m in [re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", l)] if m]
print("samples", len(pw), "power W: min %.0f median %.0f max %.0f" % (min(pw), sorted(pw)[len(pw)//2], max(pw)) if pw else "no power")
print("sclk MHz: min %d median %d max %d" % (min(ck), sorted(ck)[len(ck)//2], max(ck)) if ck else "no sclk")
print("power cap W:", mx[:1])
PY
cat gpurun_out/power_bench.json | cut -c1-200
