"""Deviation of every precision mode from the live reference on the representative forward set (tests/golden/*_fwd_set.npz),
measured on the GPU.   python scripts/r4/fwd_set_modes.py [large|small] [modes...]  -> gpurun_out/fwd_set_<model>.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as C  # noqa: E402
from ivid_amd.diffusion.backbones import AdmUnet2d  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "large"
modes = sys.argv[2:] or ["fp32", "bf16x3", "fp16cx", "fp16c", "fp16", "bf16"]
args, seed, gname = (C.LARGE128, 4, "large128_fwd_set") if model == "large" else (C.SMALL128, 3, "small128_fwd_set")
g = C.load_golden(gname)
ins = C.fwd_set_inputs(args["in_channels"], args["image_size"])
x = torch.cat([i[1] for i in ins]).cuda()
t = torch.tensor([i[2] for i in ins]).cuda()
has_cls = args["num_classes"] is not None
cls = torch.tensor([i[3] for i in ins]).cuda() if has_cls else None
m = AdmUnet2d(**args, precision=modes[0])
m.load_state_dict(C.synth_weights(args, seed), strict=True)
m = m.cuda().eval()
out = {}
for p in modes:
    m.set_precision(p)
    if has_cls:
        ec, eu = m.forward_cfg(x, t, cls)
        ec, eu = ec.cpu(), eu.cpu()
    else:
        ec, eu = None, m(x, t, None).cpu()
    rows = {}
    for i, (key, _, _, _) in enumerate(ins):
        if ec is not None:
            rows[key + "_c"] = C.rel_l2(ec[i], g[key + "_c"])
        rows[key + "_u"] = C.rel_l2(eu[i], g[key + "_u"])
    out[p] = dict(max=max(rows.values()), argmax=max(rows, key=rows.get), min=min(rows.values()), rows=rows)
    print(model, p, "max %.3e (%s) min %.3e" % (out[p]["max"], out[p]["argmax"], out[p]["min"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"fwd_set_{model}.json"), "w"), indent=1)
