"""Chain goldens for the sampler settings of BASELINE configs 3 / 4 (round 4), from the LIVE reference (/root/reference),
build container only:   python tests/golden/make_golden_chains16.py
  smallcfg_ddpm250_cfg3     DdpmSampler (ancestral sampling, ddpm.py:104-160) + ClassifierFreeGuidance strength 3.0
                            (inference/sample.py:44-47,79: the unconditional first view) on a class-conditional variant of the
                            small-128 backbone with a 250-timestep framework (250 x 2 CPU forwards), bs 1, recorded x_T, the
                            ancestral noise from torch's seeded CPU generator
  mini128cond_inpaint50     InpaintCFG strength 3.0 + DdimSampler 50 steps with replace_rgb 0.1 / replace_depth 0.2 /
                            constrain_depth 0.5 (inference/sample.py:99-122) on the CONDITIONING OF THE SCENE FIXTURE
                            (sample_all_scene_ref.npz: 86-89 % mask coverage), the mini 10-channel model at 128^2, bs 2
The 16-bit modes of the product are compared against these on the GPU (tests/test_unet_gpu.py) -> profiles/r04_chain_parity.json.
The oracle chain is checked against the reference on the spot."""
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
man_path = os.path.join(HERE, "manifest.json")
man = json.load(open(man_path))


@torch.no_grad()
def ddpm_cfg3():
    args = C.SMALL128_CFG
    m = rb.AdmUnet2d(**args).eval()
    sd = C.synth_weights(args, 5)
    m.load_state_dict(sd, strict=True)
    fw = rf.ClassifierFreeGuidance(m, timesteps=250, beta_schedule="linear", p_uncond=0.1)
    smp = rs.DdpmSampler(fw)
    x_T = C.seeded_randn(311, 1, 4, 128, 128)
    cls = torch.tensor([3])
    torch.manual_seed(13)
    t0 = time.time()
    ref = smp.sample(1, noise=x_T, classes=cls, strength=3.0, verbose=False)
    dt = time.time() - t0
    um = lambda a, b, c: adm_oracle.unet_forward(sd, args, a, b, c)
    torch.manual_seed(13)
    orc = sampler_oracle.ddpm_sample(lambda x, t: sampler_oracle.cfg_eps(um, x, t, cls, 3.0), x_T, fw.betas)
    errs = dict(samples=C.rel_l2(orc["samples"], ref.samples), x0_first=C.rel_l2(orc["pred_x_0"][0], ref.pred_x_0[0]),
                ref_seconds=round(dt, 1))
    np.savez_compressed(os.path.join(HERE, "smallcfg_ddpm250_cfg3.npz"), samples=ref.samples.numpy(), x0_first=ref.pred_x_0[0].numpy(),
                        x0_mid=ref.pred_x_0[125].numpy(), x0_last=ref.pred_x_0[-1].numpy(), classes=cls.numpy(),
                        x_checksum=np.float64(x_T.double().sum()))
    man["smallcfg_ddpm250_cfg3"] = dict(
        note="DdpmSampler (250 ancestral steps of a 250-timestep framework) + ClassifierFreeGuidance strength 3.0, class-conditional "
             "small-128 backbone (tests/common.SMALL128_CFG), bs 1, x_T = seeded_randn(311), torch.manual_seed(13) noise stream",
        oracle_vs_reference=errs)
    print("smallcfg_ddpm250_cfg3", errs, "|samples| max", float(ref.samples.abs().max()), flush=True)


@torch.no_grad()
def inpaint50():
    args = C.MINI128_COND
    m = rb.AdmUnet2d(**args).eval()
    sd = C.synth_weights(args, 2)
    m.load_state_dict(sd, strict=True)
    fw = rf.InpaintCFG(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
    smp = rs.DdimSampler(fw)
    g = C.load_golden("sample_all_scene_ref")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    color, depth = T(g["cond_color"]), T(g["cond_depth"])                         # [2,3,S,S], [2,1,S,S] in [-1, 1]
    mask = T(g["cond_mask"]).permute(0, 3, 1, 2)
    mask_rgb = T(g["cond_mask_rgb"]).permute(0, 3, 1, 2)
    convex = T(g["cond_depth_convex"]).permute(0, 3, 1, 2) * 2 - 1
    y = torch.cat([color, depth], 1)
    x_T = C.seeded_randn(411, 2, 4, 128, 128)
    cls = torch.tensor([2, 8])
    kw = dict(y=y, mask=mask, mask_rgb=mask_rgb, replace_rgb=(0.1, color, mask_rgb), replace_depth=(0.2, depth, mask),
              constrain_depth=(0.5, convex))
    torch.manual_seed(17)
    t0 = time.time()
    ref = smp.sample(2, noise=x_T, classes=cls, steps=50, strength=3.0, verbose=False, **kw)
    dt = time.time() - t0
    um = lambda a, b, c: adm_oracle.unet_forward(sd, args, a, b, c)
    eps = lambda x, t: sampler_oracle.inpaint_cfg_eps(um, x, t, y, mask, cls, 3.0, mask_rgb)
    torch.manual_seed(17)
    orc = sampler_oracle.ddim_sample(eps, x_T, 50, fw.betas, replace_rgb=kw["replace_rgb"], replace_depth=kw["replace_depth"],
                                     constrain_depth=kw["constrain_depth"])
    errs = dict(samples=C.rel_l2(orc["samples"], ref.samples), x0_first=C.rel_l2(orc["pred_x_0"][0], ref.pred_x_0[0]),
                ref_seconds=round(dt, 1), mask_coverage=float(mask.mean()), mask_rgb_coverage=float(mask_rgb.mean()))
    np.savez_compressed(os.path.join(HERE, "mini128cond_inpaint50.npz"), samples=ref.samples.numpy(), x0_first=ref.pred_x_0[0].numpy(),
                        x0_mid=ref.pred_x_0[25].numpy(), x0_last=ref.pred_x_0[-1].numpy(), classes=cls.numpy(),
                        x_checksum=np.float64(x_T.double().sum()))
    man["mini128cond_inpaint50"] = dict(
        note="InpaintCFG strength 3.0 + DdimSampler 50 steps, replace_rgb 0.1 / replace_depth 0.2 / constrain_depth 0.5 on the scene "
             "fixture's conditioning (sample_all_scene_ref.npz cond_*), mini 10-channel model at 128^2, bs 2, x_T = seeded_randn(411)",
        oracle_vs_reference=errs)
    print("mini128cond_inpaint50", errs, "|samples| max", float(ref.samples.abs().max()), flush=True)


which = sys.argv[1:] or ["inpaint", "ddpm"]
if "inpaint" in which:
    inpaint50()
if "ddpm" in which:
    ddpm_cfg3()
json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
