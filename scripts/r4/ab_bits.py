"""Bit-identity of the refactored fused kernel against the library built from the previous commit (ab/libivid_head.so): the
outputs of whole forwards are hashed under each library (IVID_HIP_LIB) for every (model, mode) whose launch plan is the same
under both.   python scripts/r4/ab_bits.py <tag>  -> gpurun_out/ab_bits_<tag>.json"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as C  # noqa: E402
from ivid_amd.diffusion.backbones import AdmUnet2d  # noqa: E402

CASES = [("large", C.LARGE128, 4, 2, ["fp32", "bf16x3", "fp16s", "fp16cx", "fp16c", "fp16", "bf16"]),
         ("small", C.SMALL128, 3, 2, ["fp32", "fp16cx", "fp16c", "fp16", "bf16"]),
         ("sr256", C.SR256, 6, 1, ["fp16cx", "fp16", "bf16"]),
         ("mini_cond", C.MINI_COND, 2, 3, ["fp32", "fp16cx", "fp16c", "fp16", "bf16"])]
out = {}
for name, args, seed, bs, modes in CASES:
    S = args["image_size"]
    x = C.seeded_randn(900 + seed, bs, args["in_channels"], S, S).cuda()
    t = torch.tensor([999, 20, 500][:bs]).cuda()
    has_cls = args["num_classes"] is not None
    cls = (torch.tensor([7, 416, 3][:bs]) % args["num_classes"]).cuda() if has_cls else None
    m = AdmUnet2d(**args, precision=modes[0])
    m.load_state_dict(C.synth_weights(args, seed))
    m = m.cuda().eval()
    for p in modes:
        m.set_precision(p)
        if has_cls:
            ec, eu = m.forward_cfg(x, t, cls)
            y = torch.cat([ec, eu]).cpu()
        else:
            y = m(x, t, None).cpu()
        out[f"{name}/{p}"] = hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16]
        print(name, p, out[f"{name}/{p}"], float(y.abs().mean()), flush=True)
    del m
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ab_bits_%s.json" % sys.argv[1]), "w"), indent=1)
