"""Teacher-forced per-step fixtures at GUIDANCE STRENGTH 3.0 (round 5), from the LIVE reference (/root/reference), build container only:
    python tests/golden/make_golden_steps_s3.py
Configs 3 / 4 / 5 of BASELINE.json sample with strength 3.0 (inference/sample.py:79,117): a guided eps = 4 eps_c - 3 eps_u carries up
to 7 x a forward's deviation, and a chain's final samples cannot say what a single step does.  The two strength-3 chains of
make_golden_chains16.py are re-run with a spy: at a few steps the tensor the reference fed its BACKBONE (for InpaintCFG: the
10-channel conditional input with that step's fresh hole noise, inpaint_cfg.py:24-49,80-83) and the guided eps it got back
(classifier_free_guidance.py:39-42) are recorded.  The spy is inert: the chain's samples must equal the committed golden bit for bit.
  smallcfg_ddpm250_cfg3_steps   DdpmSampler 250 + ClassifierFreeGuidance 3.0 (class-conditional small-128), steps t = 249, 189, 124, 59, 9, 0
  mini128cond_inpaint50_steps   InpaintCFG 3.0 + DdimSampler 50 on the scene fixture's conditioning, sampler steps 0, 20, 45 (t = 999, 599, 99)
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

torch.set_num_threads(os.cpu_count())
man_path = os.path.join(HERE, "manifest.json")
man = json.load(open(man_path))


def spy_on(fw, m, steps):
    """Record, at the model_inference calls listed in `steps`, the first tensor handed to the backbone and the guided eps returned."""
    rec, calls, inner, state = {}, [0], fw.model_inference, {"want": False}

    def pre(_mod, inputs):
        if state["want"]:
            rec[f"in_step{calls[0]}"] = inputs[0].numpy().copy()
            rec[f"t_step{calls[0]}"] = np.int64(int(inputs[1][0]))
            state["want"] = False
    h = m.register_forward_pre_hook(pre)

    def wrapped(*a, **kw):
        state["want"] = calls[0] in steps
        out = inner(*a, **kw)
        if calls[0] in steps:
            rec[f"eps_step{calls[0]}"] = out.numpy().copy()
        calls[0] += 1
        return out
    fw.model_inference = wrapped
    return rec, (lambda: (h.remove(), setattr(fw, "model_inference", inner)))


@torch.no_grad()
def ddpm_cfg3():
    args = C.SMALL128_CFG
    m = rb.AdmUnet2d(**args).eval()
    m.load_state_dict(C.synth_weights(args, 5), strict=True)
    fw = rf.ClassifierFreeGuidance(m, timesteps=250, beta_schedule="linear", p_uncond=0.1)
    smp = rs.DdpmSampler(fw)
    x_T, cls = C.seeded_randn(311, 1, 4, 128, 128), torch.tensor([3])
    rec, undo = spy_on(fw, m, {0, 60, 125, 190, 240, 249})
    torch.manual_seed(13)
    t0 = time.time()
    ref = smp.sample(1, noise=x_T, classes=cls, strength=3.0, verbose=False)
    undo()
    assert np.array_equal(C.load_golden("smallcfg_ddpm250_cfg3")["samples"], ref.samples.numpy()), "the spied chain differs from the committed golden"
    rec["classes"] = cls.numpy()
    np.savez_compressed(os.path.join(HERE, "smallcfg_ddpm250_cfg3_steps.npz"), **rec)
    man["smallcfg_ddpm250_cfg3_steps"] = dict(
        note="backbone input x_t and guided eps (strength 3.0) at sampler steps 0, 60, 125, 190, 240, 249 of the smallcfg_ddpm250_cfg3 chain",
        chain_reproduced_bit_identically=True, ref_seconds=round(time.time() - t0, 1), t={k: int(v) for k, v in rec.items() if k.startswith("t_")})
    print("smallcfg_ddpm250_cfg3_steps", man["smallcfg_ddpm250_cfg3_steps"], flush=True)


@torch.no_grad()
def inpaint50():
    args = C.MINI128_COND
    m = rb.AdmUnet2d(**args).eval()
    m.load_state_dict(C.synth_weights(args, 2), strict=True)
    fw = rf.InpaintCFG(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
    smp = rs.DdimSampler(fw)
    g = C.load_golden("sample_all_scene_ref")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    color, depth = T(g["cond_color"]), T(g["cond_depth"])
    mask = T(g["cond_mask"]).permute(0, 3, 1, 2)
    mask_rgb = T(g["cond_mask_rgb"]).permute(0, 3, 1, 2)
    convex = T(g["cond_depth_convex"]).permute(0, 3, 1, 2) * 2 - 1
    y = torch.cat([color, depth], 1)
    x_T, cls = C.seeded_randn(411, 2, 4, 128, 128), torch.tensor([2, 8])
    kw = dict(y=y, mask=mask, mask_rgb=mask_rgb, replace_rgb=(0.1, color, mask_rgb), replace_depth=(0.2, depth, mask),
              constrain_depth=(0.5, convex))
    rec, undo = spy_on(fw, m, {0, 20, 45})
    torch.manual_seed(17)
    t0 = time.time()
    ref = smp.sample(2, noise=x_T, classes=cls, steps=50, strength=3.0, verbose=False, **kw)
    undo()
    assert np.array_equal(C.load_golden("mini128cond_inpaint50")["samples"], ref.samples.numpy()), "the spied chain differs from the committed golden"
    rec["classes"] = cls.numpy()
    np.savez_compressed(os.path.join(HERE, "mini128cond_inpaint50_steps.npz"), **rec)
    man["mini128cond_inpaint50_steps"] = dict(
        note="10-channel backbone input (with that step's hole noise) and guided eps (strength 3.0) at sampler steps 0, 20, 45 of the "
             "mini128cond_inpaint50 chain", chain_reproduced_bit_identically=True, ref_seconds=round(time.time() - t0, 1),
        t={k: int(v) for k, v in rec.items() if k.startswith("t_")})
    print("mini128cond_inpaint50_steps", man["mini128cond_inpaint50_steps"], flush=True)


which = sys.argv[1:] or ["inpaint", "ddpm"]
if "inpaint" in which:
    inpaint50()
if "ddpm" in which:
    ddpm_cfg3()
json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
