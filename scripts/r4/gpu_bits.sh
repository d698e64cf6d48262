#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
IVID_HIP_LIB=$PWD/ab/libivid_head.so python scripts/r4/ab_bits.py head 2>&1 | grep -v amdgpu.ids | tail -2
python scripts/r4/ab_bits.py new 2>&1 | grep -v amdgpu.ids | tail -2
python - <<'PY'
import json
a, b = json.load(open("gpurun_out/ab_bits_head.json")), json.load(open("gpurun_out/ab_bits_new.json"))
bad = [k for k in a if a[k] != b.get(k)]
print("bit-identical cases: %d / %d" % (len(a) - len(bad), len(a)), "DIFFER:", bad)
PY
timeout 300 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -p no:cacheprovider -k "out_of_range" 2>&1 | tail -2
