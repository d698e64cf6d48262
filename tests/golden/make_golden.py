"""Generate the golden fixtures from the LIVE reference (/root/reference) and pin the oracle to it.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
For every case it (1) builds the reference model from backbone args, (2) loads the deterministic
synthetic checkpoint (oracle/synth.py — a reference-format bare state_dict, strict load), (3) runs
the reference on seeded inputs, (4) checks the oracle restatement against it, and (5) stores the
reference outputs as small .npz fixtures + manifest.json with the measured oracle-vs-reference errors.
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# ---- easydict shim (the only missing import on the reference's sampler path) ----
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

torch.set_num_threads(os.cpu_count())
manifest = {}


def ref_model(args, seed):
    m = rb.AdmUnet2d(**args).eval()
    sd = C.synth_weights(args, seed)
    m.load_state_dict(sd, strict=True)
    return m, sd


def record(name, arrays, errs, note):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: np.asarray(v) for k, v in arrays.items()})
    manifest[name] = dict(note=note, oracle_vs_reference=errs)
    print(f"{name}: {errs}", flush=True)


@torch.no_grad()
def forward_case(name, args, seed, batch, t, classes, note):
    m, sd = ref_model(args, seed)
    S = args["image_size"]
    x = C.seeded_randn(100 + seed, batch, args["in_channels"], S, S)
    tt = torch.full((batch,), t, dtype=torch.long)
    cls = torch.tensor(classes, dtype=torch.long) if classes is not None else None
    t0 = time.time(); ref = m(x, tt, cls); dt = time.time() - t0
    orc = adm_oracle.unet_forward(sd, args, x, tt, cls)
    errs = dict(rel_l2=C.rel_l2(orc, ref), max_rel=C.max_rel(orc, ref), ref_seconds=round(dt, 2))
    arrays = dict(eps=ref.numpy(), t=np.int64(t), x_checksum=np.float64(x.double().sum()))
    if classes is not None:
        arrays["classes"] = np.asarray(classes, dtype=np.int64)
        ref_u = m(x, tt, None)
        errs["rel_l2_uncond"] = C.rel_l2(adm_oracle.unet_forward(sd, args, x, tt, None), ref_u)
        arrays["eps_uncond"] = ref_u.numpy()
    if S <= 32:
        arrays["x"] = x.numpy()
    record(name, arrays, errs, note)


forward_case("mini_fwd", C.MINI, 0, 2, 37, [3, -1], "mini UNet, classes [3, null], t=37")
forward_case("mini_unclass_fwd", C.MINI_UNCLASS, 1, 2, 999, None, "mini UNet without class embedding, t=999")
forward_case("mini_cond_fwd", C.MINI_COND, 2, 2, 0, [9, 0], "mini 10-channel (inpaint) UNet, t=0")
forward_case("small128_fwd", C.SMALL128, 3, 1, 500, None, "rgbd_singlecategory_adm_128_small backbone (fp32), bs 1")
forward_case("large128_fwd", C.LARGE128, 4, 1, 999, [7], "rgbd_imagenet_adm_128_large_cfg backbone, bs 1, both CFG branches")


# ---------------- sampler chains ----------------
@torch.no_grad()
def ddim_cfg_case():
    args = C.MINI
    m, sd = ref_model(args, 0)
    fw = rf.ClassifierFreeGuidance(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    smp = rs.DdimSampler(fw)
    x_T = C.seeded_randn(11, 2, 4, 32, 32)
    cls = torch.tensor([1, 5])
    torch.manual_seed(5)
    ref = smp.sample(2, noise=x_T, classes=cls, steps=5, strength=0.5, eta=0.5, verbose=False)
    torch.manual_seed(5)
    eps = lambda x, t: sampler_oracle.cfg_eps(lambda a, b, c: adm_oracle.unet_forward(sd, args, a, b, c), x, t, cls, 0.5)
    orc = sampler_oracle.ddim_sample(eps, x_T, 5, fw.betas, eta=0.5)
    errs = dict(samples=C.rel_l2(orc["samples"], ref.samples), x0_first=C.rel_l2(orc["pred_x_0"][0], ref.pred_x_0[0]))
    record("mini_ddim_cfg", dict(x_T=x_T.numpy(), classes=cls.numpy(), samples=ref.samples.numpy(),
                                 x0_first=ref.pred_x_0[0].numpy(), x0_last=ref.pred_x_0[-1].numpy()), errs,
           "ClassifierFreeGuidance + DdimSampler, 5 steps, strength 0.5, eta 0.5, torch.manual_seed(5) noise stream")


@torch.no_grad()
def ddim_inpaint_case():
    args = C.MINI_COND
    m, sd = ref_model(args, 2)
    fw = rf.InpaintCFG(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
    smp = rs.DdimSampler(fw)
    B, S = 2, 32
    x_T = C.seeded_randn(21, B, 4, S, S)
    y = C.seeded_randn(22, B, 4, S, S).clamp(-1, 1)
    mask = (C.seeded_randn(23, B, 1, S, S) > 0).float()
    mask_rgb = mask * (C.seeded_randn(24, B, 1, S, S) > -0.5).float()
    convex = C.seeded_randn(25, B, 1, S, S).clamp(-1, 1)
    cls = torch.tensor([2, 8])
    kw = dict(y=y, mask=mask, mask_rgb=mask_rgb, replace_rgb=(0.1, y[:, :3], mask_rgb),
              replace_depth=(0.2, y[:, 3:], mask), constrain_depth=(0.5, convex))
    torch.manual_seed(7)
    ref = smp.sample(B, noise=x_T, classes=cls, steps=4, strength=3.0, verbose=False, **kw)
    torch.manual_seed(7)
    um = lambda a, b, c: adm_oracle.unet_forward(sd, args, a, b, c)
    eps = lambda x, t: sampler_oracle.inpaint_cfg_eps(um, x, t, y, mask, cls, 3.0, mask_rgb)
    orc = sampler_oracle.ddim_sample(eps, x_T, 4, fw.betas, replace_rgb=kw["replace_rgb"],
                                     replace_depth=kw["replace_depth"], constrain_depth=kw["constrain_depth"])
    errs = dict(samples=C.rel_l2(orc["samples"], ref.samples), x0_first=C.rel_l2(orc["pred_x_0"][0], ref.pred_x_0[0]))
    record("mini_ddim_inpaint", dict(x_T=x_T.numpy(), y=y.numpy(), mask=mask.numpy(), mask_rgb=mask_rgb.numpy(),
                                     convex=convex.numpy(), classes=cls.numpy(), samples=ref.samples.numpy(),
                                     x0_first=ref.pred_x_0[0].numpy(), x0_last=ref.pred_x_0[-1].numpy()), errs,
           "InpaintCFG + DdimSampler 4 steps, strength 3.0, replace_rgb 0.1 / replace_depth 0.2 / constrain 0.5 "
           "(inference/sample.py:106-119), torch.manual_seed(7) noise stream")


@torch.no_grad()
def ddpm_case():
    args = C.MINI_UNCLASS
    m, sd = ref_model(args, 1)
    fw = rf.GaussianDiffusion(m, timesteps=100, beta_schedule="linear")
    smp = rs.DdpmSampler(fw)
    x_T = C.seeded_randn(31, 2, 4, 32, 32)
    torch.manual_seed(9)
    ref = smp.sample(2, noise=x_T, verbose=False)
    torch.manual_seed(9)
    eps = lambda x, t: adm_oracle.unet_forward(sd, args, x, t, None)
    orc = sampler_oracle.ddpm_sample(eps, x_T, fw.betas)
    errs = dict(samples=C.rel_l2(orc["samples"], ref.samples), x0_first=C.rel_l2(orc["pred_x_0"][0], ref.pred_x_0[0]))
    record("mini_ddpm", dict(x_T=x_T.numpy(), samples=ref.samples.numpy(), x0_first=ref.pred_x_0[0].numpy()), errs,
           "GaussianDiffusion(timesteps=100) + DdpmSampler (100 ancestral steps), torch.manual_seed(9) noise stream")


@torch.no_grad()
def c1_case():
    args = C.SMALL128
    m, sd = ref_model(args, 3)
    fw = rf.GaussianDiffusion(m, timesteps=1000, beta_schedule="linear")
    smp = rs.DdimSampler(fw)
    x_T = C.seeded_randn(123, 2, 4, 128, 128)
    torch.manual_seed(1)
    t0 = time.time()
    ref = smp.sample(2, noise=x_T, steps=10, verbose=False)
    dt = time.time() - t0
    torch.manual_seed(1)
    eps = lambda x, t: adm_oracle.unet_forward(sd, args, x, t, None)
    orc = sampler_oracle.ddim_sample(eps, x_T, 10, fw.betas)
    errs = dict(samples=C.rel_l2(orc["samples"], ref.samples), x0_first=C.rel_l2(orc["pred_x_0"][0], ref.pred_x_0[0]),
                ref_seconds=round(dt, 2))
    record("small128_ddim10", dict(samples=ref.samples.numpy(), x0_first=ref.pred_x_0[0].numpy(),
                                   x_checksum=np.float64(x_T.double().sum())), errs,
           "BASELINE config 1 at bs 2: rgbd_singlecategory_adm_128_small (fp32) + 10-step DDIM, eta 0")


ddim_cfg_case()
ddim_inpaint_case()
ddpm_case()
c1_case()

manifest["_env"] = dict(torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads())
with open(os.path.join(HERE, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1, sort_keys=True)
print("wrote", os.path.join(HERE, "manifest.json"))
