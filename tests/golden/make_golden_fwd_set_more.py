"""Representative forward sets of the CONDITIONAL and the SUPER-RESOLUTION model (round 4), from the LIVE reference
(/root/reference), build container only:   python tests/golden/make_golden_fwd_set_more.py [cond] [sr] [condmid] [srmid]
  largecond128_fwd_set.npz  rgbd_imagenet_adm_128_large_cond backbone (10 input channels, fp32): x_t = q_sample(scene, t) conditioned,
                            through the reference's own InpaintCFG.make_cond_inputs (inpaint_cfg.py:24-49), on the scene seen through the
                            visibility masks of the scene fixture; t in {0, 20, 500, 999}, 2 scenes, both guidance branches (16 forwards)
  sr256_fwd_set.npz         rgbd_imagenet_adm_256_128_small_sr backbone (8 input channels, 256^2): x_t conditioned, through the
                            reference's SuperResCFG.make_cond_inputs (sr_cfg.py:23-36), on the average-pooled scene; same t / scenes /
                            branches; the 128 x 128 centre window of every output is stored (tests/common.SR_CROP)
The seeded recipes in tests/common.py (fwd_set_inputs_cond / fwd_set_inputs_sr) rebuild the inputs on the GPU box; they are checked
here against what the reference's framework code builds.  The oracle is checked against every output on the spot."""
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
man_path = os.path.join(HERE, "manifest.json")
man = json.load(open(man_path))


TS, FILL_BASE = C.FWD_SET_T_MORE, 9000     # what check_cond needs to re-draw the recipe's fills (set per run below)


@torch.no_grad()
def run(name, args, seed, inputs, check_input, crop):
    m = rb.AdmUnet2d(**args).eval()
    sd = C.synth_weights(args, seed)
    m.load_state_dict(sd, strict=True)
    arrays, worst, t0 = {}, 0.0, time.time()
    for key, x, t, cls in inputs:
        check_input(m, key, x, t)
        tt = torch.tensor([t], dtype=torch.long)
        for b, cl in (("c", torch.tensor([cls])), ("u", None)):
            ref = m(x, tt, cl)
            worst = max(worst, C.rel_l2(adm_oracle.unet_forward(sd, args, x, tt, cl), ref))
            r = ref.numpy()[0]
            arrays[f"{key}_{b}"] = r if crop is None else np.ascontiguousarray(r[(slice(None),) + crop])
        arrays[f"{key}_xsum"] = np.float64(x.double().sum())
        print(name, key, f"|eps| rms {float(ref.pow(2).mean().sqrt()):.3f}", flush=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    man[name] = dict(note=f"{len(arrays) - len(inputs)} reference forwards (fp32) of {name.split('_')[0]} on conditioned q-samples of two synthetic "
                          f"RGBD scenes, t in {list(TS)}, both guidance branches" + (", 128 x 128 centre window stored" if crop else ""),
                     oracle_vs_reference=dict(rel_l2_max=worst, ref_seconds=round(time.time() - t0, 1)))
    print(name, man[name], flush=True)


def check_cond(m, key, x, t):
    """the recipe's 10-channel input IS what the reference's InpaintCFG.make_cond_inputs builds (same generator draws)"""
    fw = rf.InpaintCFG(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
    si = [s[0] for s in C.FWD_SET_SCENES].index(key.split("_t")[0]); ti = list(TS).index(t)
    sc = C.load_golden("sample_all_scene_ref")
    mask = torch.from_numpy(sc["cond_mask"][:1].astype(np.float32)).permute(0, 3, 1, 2)
    mask_rgb = torch.from_numpy(sc["cond_mask_rgb"][:1].astype(np.float32)).permute(0, 3, 1, 2)
    x0 = torch.from_numpy(WC.synthetic_rgbd(128, **C.FWD_SET_SCENES[si][1])).float()
    torch.manual_seed(FILL_BASE + 10 * si + ti)
    ref_in = fw.make_cond_inputs(x[:, :4], x0, mask, mask_rgb=mask_rgb)
    assert torch.equal(ref_in, x), key


def check_sr(m, key, x, t):
    fw = rf.SuperResCFG(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    si = [s[0] for s in C.FWD_SET_SCENES].index(key.split("_t")[0])
    x0 = torch.from_numpy(WC.synthetic_rgbd(256, **C.FWD_SET_SCENES[si][1])).float()
    low = torch.nn.functional.avg_pool2d(x0, 2).clamp(-1, 1)
    assert torch.equal(fw.make_cond_inputs(x[:, :4], low), x), key


which = sys.argv[1:] or ["cond", "sr"]
if "cond" in which:
    run("largecond128_fwd_set", C.LARGE128_COND, 2, C.fwd_set_inputs_cond(), check_cond, None)
if "sr" in which:
    run("sr256_fwd_set", C.SR256, 6, C.fwd_set_inputs_sr(), check_sr, C.SR_CROP)
# around the adaptive precision mode's threshold (t = 100, 250, 350)
TS, FILL_BASE = C.FWD_SET_T_MID_MORE, 9050
if "condmid" in which:
    run("largecond128_fwd_set_mid", C.LARGE128_COND, 2, C.FWD_SETS["largecond128_mid"][3](), check_cond, None)
if "srmid" in which:
    run("sr256_fwd_set_mid", C.SR256, 6, C.FWD_SETS["sr256_mid"][3](), check_sr, C.SR_CROP)
# round 5: t = 150, 200 (the island threshold of the adaptive modes moved from 250 to 150)
TS, FILL_BASE = C.FWD_SET_T_MID2, 9080
if "condmid2" in which:
    run("largecond128_fwd_set_mid2", C.LARGE128_COND, 2, C.FWD_SETS["largecond128_mid2"][3](), check_cond, None)
if "srmid2" in which:
    run("sr256_fwd_set_mid2", C.SR256, 6, C.FWD_SETS["sr256_mid2"][3](), check_sr, C.SR_CROP)
json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
