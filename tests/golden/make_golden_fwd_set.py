"""Reference forwards on the inputs a sampling chain actually feeds the network (round 4; SURVEY.md §8(c): "full forward
at t in {0, 20, 500, 999}"), generated from the LIVE reference (/root/reference), build container only:
    python tests/golden/make_golden_fwd_set.py [large] [small] [largemid] [smallmid]
x_t = sqrt(abar_t) * x0 + sqrt(1 - abar_t) * n, x0 = two synthetic RGBD scenes (tests/warp_common.synthetic_rgbd), t in
{0, 20, 250, 500, 750, 999} (tests/common.fwd_set_inputs: the same recipe rebuilds the inputs on the GPU box).  The
reference's own q-sample is used to make x_t (GaussianDiffusion.diffuse, gaussian_diffusion.py:45-56) and checked against the
recipe.  large: rgbd_imagenet_adm_128_large_cfg backbone (fp32), both CFG branches of every input (24 forwards) ->
large128_fwd_set.npz; small: rgbd_singlecategory_adm_128_small (12 forwards) -> small128_fwd_set.npz.  The oracle
restatement is checked against every one of them on the spot (manifest.json).
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

import common as C  # noqa: E402
import warp_common as WC  # noqa: E402
from oracle import adm_oracle  # noqa: E402

torch.set_num_threads(os.cpu_count())


@torch.no_grad()
def run(name, args, seed, with_classes, ts=None, seed_base=7000):
    """seed: int or (int, variant) -- tests/common.synth_weights"""
    ts = C.FWD_SET_T if ts is None else ts
    m = rb.AdmUnet2d(**args).eval()
    sd = C.synth_weights(args, seed)
    m.load_state_dict(sd, strict=True)
    fw = rf.GaussianDiffusion(m, timesteps=1000, beta_schedule="linear")
    arrays, worst, t0 = {}, 0.0, time.time()
    inputs = C.fwd_set_inputs(args["in_channels"], args["image_size"], ts, seed_base)
    for si, (tag, kw, _) in enumerate(C.FWD_SET_SCENES):   # the recipe's x_t IS the reference's q-sample
        x0 = torch.from_numpy(WC.synthetic_rgbd(args["image_size"], **kw)).float()
        for ti, t in enumerate(ts):
            n = C.seeded_randn(seed_base + 100 * si + ti, 1, 4, args["image_size"], args["image_size"])
            xr = fw.diffuse(x0, torch.tensor([t]), n)
            assert C.rel_l2(inputs[si * len(ts) + ti][1], xr) < 2e-7
    for key, x, t, cls in inputs:
        tt = torch.tensor([t], dtype=torch.long)
        branches = [("c", torch.tensor([cls])), ("u", None)] if with_classes else [("u", None)]
        for b, cl in branches:
            ref = m(x, tt, cl)
            orc = adm_oracle.unet_forward(sd, args, x, tt, cl)
            worst = max(worst, C.rel_l2(orc, ref))
            arrays[f"{key}_{b}"] = ref.numpy()[0]
        arrays[f"{key}_xsum"] = np.float64(x.double().sum())
        print(name, key, f"|eps| rms {float(ref.pow(2).mean().sqrt()):.3f}", flush=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    mf = os.path.join(HERE, "manifest.json")
    man = json.load(open(mf))
    man[name] = dict(note=f"{len(arrays) - len(inputs)} reference forwards (fp32) on x_t = q_sample(synthetic RGBD scene, t), "
                          f"t in {list(ts)}, scenes {[s[0] for s in C.FWD_SET_SCENES]}"
                          + (", classes [7] / [416] and the null class" if with_classes else "")
                          + (f"; synthetic checkpoint {seed}" if isinstance(seed, tuple) or seed not in (3, 4) else ""),
                     oracle_vs_reference=dict(rel_l2_max=worst, ref_seconds=round(time.time() - t0, 1)))
    json.dump(man, open(mf, "w"), indent=1, sort_keys=True)
    print(name, man[name], flush=True)


which = sys.argv[1:] or ["large", "small"]
if "large" in which:
    run("large128_fwd_set", C.LARGE128, 4, True)
if "small" in which:
    run("small128_fwd_set", C.SMALL128, 3, False)
# between the low-noise rows (t = 50, 100, 150, 350): where the adaptive precision mode switches plans
if "largemid" in which:
    run("large128_fwd_set_mid", C.LARGE128, 4, True, C.FWD_SET_T_MID, 7050)
if "smallmid" in which:
    run("small128_fwd_set_mid", C.SMALL128, 3, False, C.FWD_SET_T_MID, 7050)
# round 5: further synthetic checkpoints (more seeds + the "trained-like" variant), rows at the adaptive mode's plan changes
for tag, (sargs, sseed, sname, _make, _crop) in C.FWD_SETS_SEEDS.items():
    if tag in which or "seeds" in which:
        run(sname, sargs, sseed, sargs["num_classes"] is not None, C.FWD_SET_T_SEEDS, 7300)
