#!/bin/bash
# After the stem's fp16-twin epilogue: op + unet tests, bench, PMC passes at the final kernel sources.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle -s
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_comp_gpu.py tests/test_unet_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -3
IVID_COMMIT=${IVID_COMMIT:-unknown} PREC=fp16s bash scripts/r4/gpu_pmc_mfma.sh > gpurun_out/pmc_r4.log 2>&1; tail -2 gpurun_out/pmc_r4.log
for e in 1 0; do
IVID_NO_ISLAND_O16=$e timeout 600 python bench.py --precision fp16s --steps 20 --warmup 2 --no-cpu-baseline --no-parity-mode > gpurun_out/bench_o16b_$e.json 2> gpurun_out/bench_o16b_$e.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_o16b_$e.json").read().strip().splitlines()[-1])
print("NO_ISLAND_O16=$e", d["value"], d["ms_per_step"], d.get("kernel_time_ms_per_forward"))
PY
done
