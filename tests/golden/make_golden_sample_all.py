"""Golden vectors of the WHOLE iterative multiview loop from the reference's own `sample_all`.

Run in the build container only (needs /root/reference and `make -C oracle`).  /root/reference/inference/sample.py is
imported unchanged and its `sample_all` (sample.py:30-147) is executed on the CPU: unconditional DDIM with classifier-free
guidance -> depth_to_mesh -> aggregate_conditions on the reference's AggregationRenderer (real OpenGL: Mesa llvmpipe through
oracle/glshim) -> the conditioning wiring of sample.py:99-120 -> conditional DDIM with InpaintCFG -> ... for three views of the
`3x9` viewset.  What had to be bent, and only outside the arithmetic:
  * `.cuda()` / `device='cuda'` are hard-wired in sample.py: Tensor.cuda becomes the identity and torch.randn drops a
    'cuda' device, so everything runs on the host;
  * every random draw (torch.randn / randn_like) comes from ONE seeded CPU generator, in the reference's own call order --
    the GPU test replays the same stream through the product's `noise_fn` hook;
  * missing third-party modules get stand-ins that take no part in the arithmetic (imageio, torchvision, plyfile, tqdm is
    present) or are restated (glm, cv2.erode: see oracle/glshim/glm.py, make_golden_warp.py).
Models: the mini UNet at 128 x 128 (tests/common.py MINI128 / MINI128_COND; sample_all hard-wires 128), synthetic weights.
Stores tests/golden/sample_all_ref.npz: the three views, the conditioning tensors handed to the conditional sampler.

    python tests/golden/make_golden_sample_all.py          -> sample_all_ref.npz        (untrained weights: noise-like depth maps)
    python tests/golden/make_golden_sample_all.py scene    -> sample_all_scene_ref.npz  (well-formed scenes)

`scene`: untrained weights saturate the samples, the depth maps are noise and the conditioning masks of the next view are almost
empty (0 % colour / 1.4 % depth coverage in sample_all_ref.npz) -- `replace_rgb` / `mask_rgb` / `constrain_depth` never act.
For a fixture whose generated views are well-formed scenes, (1) the output convolution `out.2` of both synthetic checkpoints
is scaled by 4e-6, so the predicted noise is a small texture instead of the whole signal, and (2) the three x_T draws
(draw 0, 4 and 11 of the stream) are sqrt(alpha_bar_T) * (a smooth synthetic RGBD scene, tests/warp_common.py) + 4e-6 * (the
stream's draw): DDIM's first x_0 prediction is then the scene plus texture (ddim.py:36-37).  Everything else -- the reference's
sample_all, samplers, frameworks, renderer, shaders -- runs unchanged.  The generator ASSERTS mask coverage >= 50 % and
mask_rgb >= 30 % on both conditional views, and also stores what aggregate_conditions returned (mask, mask_rgb, depth_convex).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.glshim import glm, moderngl  # noqa: E402
import common as C  # noqa: E402


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
    __setattr__ = dict.__setitem__


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _erode(img, kernel, iterations=1):
    out = ndimage.minimum_filter(np.asarray(img, np.float64), size=kernel.shape, mode="constant", cval=np.inf)
    return out.astype(np.asarray(img).dtype)


sys.modules["moderngl"] = moderngl
sys.modules["glm"] = glm
_mod("easydict", EasyDict=EasyDict)
_mod("cv2", erode=_erode)
_mod("plyfile")
_mod("imageio")
tv = _mod("torchvision")
tv.utils = _mod("torchvision.utils")

SCENE = len(sys.argv) > 1 and sys.argv[1] == "scene"
XT_DRAWS = (0, 4, 11)          # positions of the three x_T draws in the stream (uncond: x_T + 3 step draws; cond view: x_T + 2 x 3)
XT_NOISE, OUT_SCALE = 4e-6, 4e-6
XT_BASE = {}                   # draw index -> tensor added to XT_NOISE * draw (filled below, scene mode)

# ---- one seeded CPU noise stream for every draw, no CUDA ----
GEN = torch.Generator(device="cpu")
GEN.manual_seed(20260926)
DRAWS = []
_randn = torch.randn


def randn(*shape, **kw):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    kw.pop("device", None)
    if kw.get("generator") is not None:               # an explicit generator (oracle/synth.py's weights): not part of the stream
        return _randn(*shape, **kw)
    t = _randn(*shape, generator=GEN, dtype=kw.get("dtype", torch.float32))
    if len(DRAWS) in XT_BASE:
        t = XT_BASE[len(DRAWS)] + XT_NOISE * t
    DRAWS.append(tuple(t.shape))
    return t


torch.randn = randn
torch.randn_like = lambda x, **kw: randn(*x.shape)
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/inference")
import diffusion.backbones as rb  # noqa: E402
import diffusion.frameworks as rf  # noqa: E402
spec = importlib.util.spec_from_file_location("ref_inference_sample", "/root/reference/inference/sample.py")
ref_sample = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_sample)

torch.set_num_threads(os.cpu_count())


def model(args, seed):
    m = rb.AdmUnet2d(**args).eval()
    sd = C.synth_weights(args, seed)
    if SCENE:
        sd["out.2.weight"] = sd["out.2.weight"] * OUT_SCALE
        sd["out.2.bias"] = sd["out.2.bias"] * OUT_SCALE
    m.load_state_dict(sd, strict=True)
    return m


fu = rf.ClassifierFreeGuidance(model(C.MINI128, 0), timesteps=1000, beta_schedule="linear", p_uncond=0.1)
fc = rf.InpaintCFG(model(C.MINI128_COND, 2), timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
import warp_common as WC  # noqa: E402
vs = WC.viewset_3x9()
view_ids = [0, 3, 7]                                  # front, yaw +0.15, yaw -0.15 / pitch +0.15
views = [glm.lookAt(glm.vec3(np.sin(y) * np.cos(p), np.sin(p), np.cos(y) * np.cos(p)), glm.vec3(0, 0, 0), glm.vec3(0, 1, 0))
         for y, p in (vs[k] for k in view_ids)]        # inference/sample.py:331-335
SU, SC, GUID, ERODE, CLS = 3, 2, 0.5, 1, [3]
captured = []
if SCENE:
    xt_scale = float(fu.alphas_cumprod[-1]) ** 0.5          # both chains start at t = 999 (ddim.py:141-146)
    for k, d in enumerate(XT_DRAWS):
        XT_BASE[d] = xt_scale * torch.from_numpy(WC.synthetic_rgbd(128, 40 + k, smooth_color=True))
    import rgbd_3d.utils as ref_utils
    _agg = ref_utils.aggregate_conditions

    def agg(*a, **kw):
        c = _agg(*a, **kw)
        captured.append({k: np.asarray(c[k]).copy() for k in ("color", "depth", "mask", "mask_rgb", "depth_convex")})
        return c
    ref_utils.aggregate_conditions = agg
    ref_sample.rgbd_3d.utils.aggregate_conditions = agg
out = list(ref_sample.sample_all(fu, fc, 1, SU, SC, views, classes=CLS, guidance=GUID, batchsize=1, erode_rgb=ERODE))
meshes, colors, samples, conds = out[0]
print("views", tuple(samples.shape), "conds", {k: tuple(v.shape) for k, v in conds.items()}, "draws", len(DRAWS))
print("depth range of view 0 (in [-1,1]):", float(samples[0, 3].min()), float(samples[0, 3].max()))
extra = {}
name = "sample_all_ref.npz"
if SCENE:
    name = "sample_all_scene_ref.npz"
    assert len(captured) == 2 and DRAWS[0] == DRAWS[4] == DRAWS[11] == (1, 4, 128, 128)
    for j, c in enumerate(captured):
        cov, cov_rgb = float(c["mask"].mean()), float(c["mask_rgb"].mean())
        print(f"view {j + 1}: mask coverage {cov:.3f}, mask_rgb coverage {cov_rgb:.3f}")
        assert cov >= 0.5 and cov_rgb >= 0.3, "the fixture must exercise the conditioning"
        for k in ("mask", "mask_rgb"):
            extra[f"cond_{k}"] = np.stack([cc[k] for cc in captured]).astype(np.uint8)
        extra["cond_depth_convex"] = np.stack([cc["depth_convex"] for cc in captured]).astype(np.float32)
    assert float(samples.abs().max()) <= 1.0 + 1e-6
    extra.update(xt_scale=np.float64(xt_scale), xt_noise=np.float64(XT_NOISE), out_scale=np.float64(OUT_SCALE),
                 xt_draws=np.array(XT_DRAWS), scene_seeds=np.array([40, 41, 42]))
np.savez_compressed(os.path.join(HERE, name), **extra, samples=samples.numpy().astype(np.float32),
                    cond_color=conds["color"].numpy().astype(np.float32), cond_depth=conds["depth"].numpy().astype(np.float32),
                    view_ids=np.array(view_ids), cfg=np.array([SU, SC, ERODE]), guidance=np.array(GUID), classes=np.array(CLS),
                    draws=np.array([("x".join(map(str, d))) for d in DRAWS]), noise_seed=np.array(20260926))
print("wrote", name, round(os.path.getsize(os.path.join(HERE, name)) / 1e6, 2), "MB")
