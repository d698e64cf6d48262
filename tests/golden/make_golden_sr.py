"""Golden fixture of the super-resolution framework from the LIVE reference: rf.SuperResCFG + rs.DdimSampler
(/root/reference/diffusion/frameworks/sr_cfg.py:23-60, the call of trainers/superres.py:120-124) on the 64-px mini SR model.

Run in the build container only:  python tests/golden/make_golden_sr.py   -> tests/golden/mini_superres.npz
(kept apart from make_golden.py so that the large fixtures need not be regenerated; same shims, same seeding rules)."""
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
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v
    def __setitem__(self, k, v):
        super().__setitem__(k, EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v)
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
    __setattr__ = __setitem__


_m = types.ModuleType("easydict"); _m.EasyDict = EasyDict; sys.modules["easydict"] = _m
sys.path.insert(0, "/root/reference")
import diffusion.backbones as rb  # noqa: E402
import diffusion.frameworks as rf  # noqa: E402
import diffusion.samplers as rs  # noqa: E402

import common as C  # noqa: E402
from oracle import adm_oracle, sampler_oracle  # noqa: E402
import torch.nn.functional as F  # noqa: E402

args = C.MINI_SR
m = rb.AdmUnet2d(**args).eval()
sd = C.synth_weights(args, 7)
m.load_state_dict(sd, strict=True)
fw = rf.SuperResCFG(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
smp = rs.DdimSampler(fw)
low = C.seeded_randn(62, 2, 4, 32, 32).clamp(-1, 1)
cls = torch.tensor([4, 9])
with torch.no_grad():
    # one framework call (make_cond_inputs + both guidance branches) ...
    x = C.seeded_randn(63, 2, 4, 64, 64)
    t = torch.full((2,), 500, dtype=torch.long)
    eps = fw.model_inference(x, t, low, classes=cls, strength=3.0)
    ci = fw.make_cond_inputs(x, low)
    # ... and the sampling chain; x_T is the first draw of the seeded CPU generator (ddim.py:150)
    torch.manual_seed(3)
    ref = smp.sample(2, classes=cls, steps=4, strength=3.0, verbose=False, y=low)
    # oracle restatement against it
    um = lambda a, b, c: adm_oracle.unet_forward(sd, args, a, b, c)

    def eps_fn(x_t, tt):
        c = torch.cat([x_t, F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=False)], dim=1)
        return 4.0 * um(c, tt, cls) - 3.0 * um(c, tt, None)
    torch.manual_seed(3)
    orc = sampler_oracle.ddim_sample(eps_fn, torch.randn(2, 4, 64, 64), 4, fw.betas)["samples"]
err = C.rel_l2(orc, ref.samples)
print("oracle vs reference SuperResCFG chain:", err)
assert err < 1e-5
np.savez_compressed(os.path.join(HERE, "mini_superres.npz"), low=low.numpy(), classes=cls.numpy(), x=x.numpy(),
                    cond_inputs=ci.numpy(), eps=eps.numpy(), samples=ref.samples.numpy(), x0_first=ref.pred_x_0[0].numpy())
print("wrote mini_superres.npz")
