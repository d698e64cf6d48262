#!/bin/bash
# Last GPU call of round 4: the tests added after the final validation, then the PMC passes at the final kernel sources.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle -s
timeout 900 python -m pytest tests/test_comp_gpu.py tests/test_unet_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gn_partial_c or unfused_statistics or full_size or batch_invariant or plan_cache" 2>&1 | tail -6
IVID_COMMIT=${IVID_COMMIT:-unknown} PREC=fp16s bash scripts/r4/gpu_pmc_mfma.sh > gpurun_out/pmc_r4.log 2>&1; tail -3 gpurun_out/pmc_r4.log
timeout 600 python bench.py --steps 20 --warmup 2 > gpurun_out/bench_default_final.json 2> gpurun_out/bench_default_final.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_final.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "precision_mode", "forward_rel_l2_max_over_set", "mfma_roofline_frac_whole_step")})
print(d["roofline"]["mfma_util"], d["roofline"]["traffic_source"])
PY
