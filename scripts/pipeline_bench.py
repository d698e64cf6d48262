#!/usr/bin/env python
"""Driver-level measurement (SURVEY.md §8d iv / rows a14-a23): the depth-warp kernels alone and the whole multiview loop
(uncond sample -> mesh -> warp -> conditional inpainting) at full size on one MI355X, synthetic weights.

  part 1  WarpRenderer at B = 32, 128^2, SSAA 3: add_view (mesh build) and conditions (rasterise all source views +
          aggregate + resolve) for 1 / 8 / 26 stored views; ms per call and effective GB/s of the algorithmic bytes
          (per (sample, source view) 0.6 MB vertices + 0.2 MB texture in, 9.4 MB aggregation R+W at 384^2).
  part 2  sample_all with the large-128 uncond + cond models (bf16), bs 32, reduced step counts (DDIM 10 / 10) and 4
          views: samples/s and the share of the time spent outside the UNet (warp + glue).
Prints one JSON object; tuning / reporting aid, not the bench contract."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import common as C  # noqa: E402
import warp_common as WC  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def warp_part(noise=False):
    """noise=True: white-noise depth maps (what random-init weights generate): every triangle is a sliver in the other views."""
    from ivid_amd.rgbd_3d import WarpRenderer, camera
    B, S = 32, 128
    r = WarpRenderer(B, S, 3, 27)
    views = camera.viewset("3x9")
    rgbd = torch.from_numpy(np.concatenate([WC.synthetic_rgbd(S, k, layers="noise" if noise else False) for k in range(B)])).cuda()
    out = {"B": B, "S": S, "ssaa": 3, "scene": "white-noise depth" if noise else "smooth surfaces + one step"}
    out["add_view_ms"] = round(timed(lambda: (r.reset(), r.add_view(rgbd, views[0])), 10), 3)
    r.reset()
    rows = []
    for v in range(26):
        r.add_view(rgbd, views[v])
        if v + 1 in (1, 8, 26):
            ms = timed(lambda: r.conditions(views[v + 1]), 5)
            byt = B * (v + 1) * (0.6e6 + 0.2e6 + 9.4e6) + B * 2.9e6
            torch.cuda.synchronize()
            rows.append({"source_views": v + 1, "queued_large_triangles": int(r.work[0].item()), "conditions_ms": round(ms, 3), "ms_per_sample_and_source_view": round(ms / B / (v + 1), 5),
                         "algorithmic_GB_per_s": round(byt / ms / 1e6, 1)})
    out["conditions"] = rows
    return out


def pipeline_part():
    from ivid_amd.diffusion import frameworks
    from ivid_amd.diffusion.backbones import AdmUnet2d
    from ivid_amd.inference.sample import sample_all
    from ivid_amd.rgbd_3d import camera
    bs, nviews, su, sc = 32, 4, 10, 10
    mu = AdmUnet2d(**C.LARGE128, precision="bf16"); mu.load_state_dict(C.synth_weights(C.LARGE128, 0)); mu = mu.cuda()
    cargs = dict(C.LARGE128, in_channels=10)          # rgbd_imagenet_adm_128_large_cond.json: class-conditional too
    mc = AdmUnet2d(**cargs, precision="bf16"); mc.load_state_dict(C.synth_weights(cargs, 2)); mc = mc.cuda()
    fu = frameworks.ClassifierFreeGuidance(mu, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    fc = frameworks.InpaintCFG(mc, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
    views = camera.viewset("3x9")[:nviews]
    seeds = list(range(bs))
    classes = [s % 1000 for s in seeds]

    g = torch.Generator(device="cuda").manual_seed(1)
    nf = (lambda shape: torch.randn(tuple(shape), device="cuda", generator=g)) if os.environ.get("NOISE") == "global" else None

    def run():
        return list(sample_all(fu, fc, seeds, su, sc, views, classes=classes, guidance=3.0, batchsize=bs, noise_fn=nf))
    run()                                   # warm-up: plans, graphs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(torch.isfinite(r[0]).all() for r in res)
    fwd = 2 * bs * su + (nviews - 1) * 2 * bs * sc    # CFG: 2 forwards per step for both models
    return {"bs": bs, "views": nviews, "steps_uncond": su, "steps_cond": sc, "seconds": round(dt, 3),
            "samples_per_s": round(bs / dt, 3), "views_per_s": round(bs * nviews / dt, 2),
            "sample_forwards": fwd, "sample_fwd_per_s_end_to_end": round(fwd / dt, 1)}


if __name__ == "__main__":
    out = {"warp": warp_part(), "warp_noise_depth": warp_part(noise=True)}
    if os.environ.get("SKIP_PIPELINE") != "1":
        out["pipeline"] = pipeline_part()
    print(json.dumps(out, indent=1))
