"""DdimSampler (reference: diffusion/samplers/ddim.py:12-165) with the whole per-step update —
CFG combine, x0 prediction, clamp, replace_rgb / replace_depth / convex-hull depth constraint, eps
re-derivation and x_{t-1} — fused into ONE HIP kernel (ivid_ddim_step), and both guidance branches
evaluated by one stacked UNet forward.

Same constructor and `sample(num, image_size, noise, classes, steps, clip_denoised, eta, verbose,
**kwargs)` / `sample_once(x_t, t, t_prev, classes, clip_denoised, eta, replace_rgb, replace_depth,
constrain_depth, **kwargs)` signatures and the same result keys (`samples`, `pred_x_t`, `pred_x_0`).
Extra, optional kwargs: `noise_fn(shape)->tensor` (inject the noise stream, used for seed parity
with the CPU reference; draw order is the reference's: [cond rgb, cond depth,] step noise) and
`keep_intermediates=False` (do not retain every step's tensors, ddim.py:161-162), and
`device_loop=True`: the whole loop as ONE C call (`ivid_sample`, samplers/device_loop.py) — no host
code between the steps, bit-identical samples; returns `pred_x_t = []`, `pred_x_0 = [the last step's]`.
"""
import ctypes as C

import numpy as np
import torch

from ... import _lib
from ...utils import AttrDict, default_noise
from .utils import announce_timestep, as_f32, f32, framework_eps, uniform_timestep

try:  # progress bars are optional plumbing
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None


class DdimSampler:
    def __init__(self, framework):
        self.framework = framework
        betas = self.framework.betas
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)

    def _coef(self, t, t_prev, eta, strength, clip_denoised, w_rgb, w_depth, w_con):
        """Scalars of ddim.py:82-100 for diffusion step t (1-based) -> t_prev, fp32 like the reference."""
        k = _lib.DdimCoef()
        ab = f32(self.alphas_cumprod, t - 1)
        abp = f32(self.alphas_cumprod_prev, t_prev)
        one = np.float32(1.0)
        sigma = np.float32(eta) * np.sqrt((one - abp) / (one - ab)) * np.sqrt(one - ab / abp)
        k.sqrt_recip_ac = f32(self.sqrt_recip_alphas_cumprod, t - 1)
        k.sqrt_recipm1_ac = f32(self.sqrt_recipm1_alphas_cumprod, t - 1)
        k.sqrt_ac_prev = np.sqrt(abp)
        k.dir_coef = np.sqrt(one - abp - sigma * sigma)
        k.sigma = sigma
        k.nonzero = 1.0 if t_prev != 0 else 0.0
        k.cfg_strength = strength
        k.replace_rgb_w, k.replace_depth_w, k.constrain_w = w_rgb, w_depth, w_con
        k.clip_denoised = 1 if clip_denoised else 0
        return k

    @torch.no_grad()
    def sample_once(self, x_t, t, t_prev, classes=None, clip_denoised=False, eta=0.0, replace_rgb=None,
                    replace_depth=None, constrain_depth=None, **kwargs):
        ti, tpi = uniform_timestep(t), uniform_timestep(t_prev)
        noise_fn = kwargs.get("noise_fn") or (lambda shape: default_noise(shape, x_t.device))
        x_t = as_f32(x_t)
        b, c, h, w = x_t.shape
        assert c == 4, "the fused step kernel is specialised for RGBD (4-channel) samples"
        t_model = torch.full((b,), ti - 1, dtype=torch.int64, device=x_t.device)
        announce_timestep(self.framework, ti - 1)
        try:
            eps_c, eps_u, strength = framework_eps(self.framework, x_t, t_model, classes, kwargs)
        finally:   # the announcement is for THIS model call: whether it ran, raised, or the framework never called the backbone
            announce_timestep(self.framework, None)
        rgb = rgb_m = dep = dep_m = convex = None
        w_rgb = w_dep = w_con = -1.0
        # raw pointers with fixed layouts go to the kernel: broadcast like the reference's tensor expressions would
        # (ddim.py:88-95), refuse shapes that cannot be (a multi-channel mask has no counterpart in the fused step)
        ex = lambda tns, ch: as_f32(tns.expand(b, ch, h, w))
        if replace_rgb is not None:          # tested with `is not None` in the reference (ddim.py:86)
            w_rgb, rgb, rgb_m = float(replace_rgb[0]), ex(replace_rgb[1], 3), ex(replace_rgb[2], 1)
        if replace_depth:                    # tested by truthiness in the reference (ddim.py:90)
            w_dep, dep, dep_m = float(replace_depth[0]), ex(replace_depth[1], 1), ex(replace_depth[2], 1)
            if constrain_depth:
                w_con, convex = float(constrain_depth[0]), ex(constrain_depth[1], 1)
        k = self._coef(ti, tpi, eta, strength, clip_denoised, w_rgb, w_dep, w_con)
        # the reference draws randn_like(x_t) every step even when eta == 0 (ddim.py:101): an injected
        # noise stream must advance identically; the on-device generator is only consulted when used
        need = float(k.sigma) != 0.0 and tpi != 0
        noise = noise_fn(tuple(x_t.shape)) if (need or kwargs.get("noise_fn")) else None
        noise = as_f32(noise) if need else None
        x_prev, x0 = torch.empty_like(x_t), torch.empty_like(x_t)
        eps_c = as_f32(eps_c)
        eps_u = as_f32(eps_u) if eps_u is not None else None
        _lib.call("ivid_ddim_step", _lib.ptr(x_t), _lib.ptr(eps_c), _lib.ptr(eps_u), C.byref(k), _lib.ptr(rgb),
                  _lib.ptr(rgb_m), _lib.ptr(dep), _lib.ptr(dep_m), _lib.ptr(convex), _lib.ptr(noise),
                  _lib.ptr(x_prev), _lib.ptr(x0), b, h * w, torch.cuda.current_stream(x_t.device).cuda_stream)
        return AttrDict({"pred_x_prev": x_prev, "pred_x_0": x0})

    @torch.no_grad()
    def sample(self, num, image_size=None, noise=None, classes=None, steps=None, clip_denoised=False, eta=0.0,
               verbose=True, **kwargs):
        backbone = self.framework.backbone.module if hasattr(self.framework.backbone, "module") else self.framework.backbone
        backbone.eval()
        keep = kwargs.pop("keep_intermediates", True)
        if image_size is None:
            image_size = backbone.image_size
        shape = (num, backbone.out_channels, image_size, image_size)
        device = backbone.device
        noise_fn = kwargs.get("noise_fn") or (lambda s: default_noise(s, device))
        img = noise if noise is not None else noise_fn(shape)
        img = img.to(device)
        steps = steps if steps is not None else self.framework.timesteps
        jump = self.framework.timesteps // steps
        pairs = [(jump * (i + 1), jump * i) for i in reversed(range(steps))]
        if kwargs.pop("device_loop", False):   # the whole loop as ONE C call (ivid_sample): device_loop.py
            from . import device_loop

            def step(t, t_prev):
                probe = self._coef(t, t_prev, eta, 0.0, clip_denoised, -1.0, -1.0, -1.0)
                return (t - 1, lambda s, wr, wd, wc: self._coef(t, t_prev, eta, s, clip_denoised, wr, wd, wc),
                        float(probe.sigma) != 0.0 and t_prev != 0)
            ret = device_loop.run(self, _lib.SAMPLE_DDIM, img, [step(t, tp) for t, tp in pairs], classes, kwargs)
            backbone.train()
            return ret
        ret = AttrDict({"samples": None, "pred_x_t": [], "pred_x_0": []})
        it = tqdm(pairs, desc="DDIM Sampling", disable=not verbose) if tqdm is not None else pairs
        for t, t_prev in it:
            out = self.sample_once(img, t, t_prev, classes, clip_denoised, eta, **kwargs)
            img = out.pred_x_prev
            if keep:
                ret.pred_x_t.append(out.pred_x_prev)
                ret.pred_x_0.append(out.pred_x_0)
        ret.samples = img
        backbone.train()  # the reference leaves the backbone in train mode (ddim.py:164)
        return ret
