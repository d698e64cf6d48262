"""Deviation of every precision mode from the live reference on the representative forward set (tests/golden/*_fwd_set.npz),
measured on the GPU.   python scripts/r4/fwd_set_modes.py [large128|small128|largecond128|sr256] [modes...]  -> gpurun_out/fwd_set_<model>.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as C  # noqa: E402
from ivid_amd.diffusion.backbones import AdmUnet2d  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "large128"
model = {"large": "large128", "small": "small128"}.get(model, model)
modes = sys.argv[2:] or ["fp32", "bf16x3", "fp16s", "fp16cx", "fp16c", "fp16", "bf16"]
args, seed = C.FWD_SETS[model][:2]
m = AdmUnet2d(**args, precision=modes[0])
m.load_state_dict(C.synth_weights(args, seed), strict=True)
m = m.cuda().eval()
out = {}
for p in modes:
    m.set_precision(p)
    rows = C.fwd_set_deviation(m, model)
    out[p] = dict(max=max(rows.values()), argmax=max(rows, key=rows.get), min=min(rows.values()), rows=rows)
    print(model, p, "max %.3e (%s) min %.3e" % (out[p]["max"], out[p]["argmax"], out[p]["min"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"fwd_set_{model}.json"), "w"), indent=1)
