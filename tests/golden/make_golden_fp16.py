"""The reference's OWN fp16 torso (`use_fp16: true`, adm.py:508-516 / backbones/utils.py:6-13) on the forward cases of
make_golden.py, run on the host in the build container: tests/golden/fwd_fp16_ref.npz.  Five of the six shipped configs
select this mode; the product's `fp16` precision is compared with these outputs (and, as before, with the fp32 ones)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class EasyDict(dict):
    def __getattr__(self, k):
        return self[k]
    __setattr__ = dict.__setitem__


_m = types.ModuleType("easydict")
_m.EasyDict = EasyDict
sys.modules["easydict"] = _m
sys.path.insert(0, "/root/reference")
import diffusion.backbones as rb  # noqa: E402
import common as C  # noqa: E402

torch.set_num_threads(os.cpu_count())
out = {}
with torch.no_grad():
    for name, args, seed, batch, t, classes in (("mini_fwd", C.MINI, 0, 2, 37, [3, -1]), ("mini_cond_fwd", C.MINI_COND, 2, 2, 0, [9, 0]),
                                                ("small128_fwd", C.SMALL128, 3, 1, 500, None), ("large128_fwd", C.LARGE128, 4, 1, 999, [7])):
        m = rb.AdmUnet2d(**dict(args, use_fp16=True)).eval()
        m.load_state_dict(C.synth_weights(args, seed), strict=True)
        S = args["image_size"]
        x = C.seeded_randn(100 + seed, batch, args["in_channels"], S, S)
        tt = torch.full((batch,), t, dtype=torch.long)
        cls = torch.tensor(classes, dtype=torch.long) if classes is not None else None
        eps = m(x, tt, cls)
        g = np.load(os.path.join(HERE, name + ".npz"))
        out[name] = eps.numpy().astype(np.float32)
        print(f"{name}: reference fp16 torso vs its fp32 output rel-L2 {C.rel_l2(eps, g['eps']):.3e}")
np.savez_compressed(os.path.join(HERE, "fwd_fp16_ref.npz"), **out)
print("wrote fwd_fp16_ref.npz")
