"""DdpmSampler (reference: diffusion/samplers/ddpm.py:12-187): ancestral sampling over ALL
framework.timesteps steps (its `steps` argument is ignored by the reference too, ddpm.py:137,177),
fixed-small posterior variance with the clipped log-variance table, 0-based t.  The per-step update
(CFG combine, x0 prediction, clamp, posterior mean, noise) is one HIP kernel (ivid_ddpm_step).
Positional order of `sample` differs from DdimSampler exactly as in the reference (ddpm.py:134-144).
"""
import ctypes as C

import numpy as np
import torch

from ... import _lib
from ...utils import AttrDict, default_noise
from .utils import announce_timestep, as_f32, f32, framework_eps, uniform_timestep

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None


class DdpmSampler:
    def __init__(self, framework):
        self.framework = framework
        betas = self.framework.betas
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        # q(x_{t-1} | x_t, x_0), ddpm.py:36-41
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    def _coef(self, t, strength, clip_denoised):
        k = _lib.DdpmCoef()
        k.sqrt_recip_ac = f32(self.sqrt_recip_alphas_cumprod, t)
        k.sqrt_recipm1_ac = f32(self.sqrt_recipm1_alphas_cumprod, t)
        k.coef1 = f32(self.posterior_mean_coef1, t)
        k.coef2 = f32(self.posterior_mean_coef2, t)
        # nonzero_mask * exp(0.5 * log_variance) in fp32 (ddpm.py:129-130)
        k.std = float(np.exp(np.float32(0.5) * f32(self.posterior_log_variance_clipped, t))) if t != 0 else 0.0
        k.cfg_strength = strength
        k.clip_denoised = 1 if clip_denoised else 0
        return k

    @torch.no_grad()
    def sample_once(self, x_t, t, classes=None, clip_denoised=False, **kwargs):
        ti = uniform_timestep(t)
        noise_fn = kwargs.get("noise_fn") or (lambda shape: default_noise(shape, x_t.device))
        x_t = as_f32(x_t)
        b = x_t.shape[0]
        hw = x_t[0].numel() // 4
        assert x_t.shape[1] == 4, "the fused step kernel is specialised for RGBD (4-channel) samples"
        t_model = torch.full((b,), ti, dtype=torch.int64, device=x_t.device)
        announce_timestep(self.framework, ti)
        try:
            eps_c, eps_u, strength = framework_eps(self.framework, x_t, t_model, classes, kwargs)
        finally:   # the announcement is for THIS model call: whether it ran, raised, or the framework never called the backbone
            announce_timestep(self.framework, None)
        k = self._coef(ti, strength, clip_denoised)
        # drawn every step by the reference, also at t == 0 where it is multiplied by 0 (ddpm.py:128-130)
        noise = noise_fn(tuple(x_t.shape)) if (ti != 0 or kwargs.get("noise_fn")) else None
        noise = as_f32(noise) if ti != 0 else None
        x_prev, x0 = torch.empty_like(x_t), torch.empty_like(x_t)
        eps_c = as_f32(eps_c)
        eps_u = as_f32(eps_u) if eps_u is not None else None
        _lib.call("ivid_ddpm_step", _lib.ptr(x_t), _lib.ptr(eps_c), _lib.ptr(eps_u), C.byref(k), _lib.ptr(noise),
                  _lib.ptr(x_prev), _lib.ptr(x0), b, hw, torch.cuda.current_stream(x_t.device).cuda_stream)
        return AttrDict({"pred_x_prev": x_prev, "pred_x_0": x0})

    @torch.no_grad()
    def sample(self, num, steps=None, image_size=None, noise=None, classes=None, clip_denoised=False, verbose=True,
               **kwargs):
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
        indices = list(range(self.framework.timesteps))[::-1]
        if kwargs.pop("device_loop", False):   # the whole loop as ONE C call (ivid_sample): device_loop.py
            from . import device_loop
            ret = device_loop.run(self, _lib.SAMPLE_DDPM, img,
                                  [(i, (lambda s, *_, i=i: self._coef(i, s, clip_denoised)), i != 0) for i in indices], classes, kwargs)
            backbone.train()
            return ret
        ret = AttrDict({"samples": None, "pred_x_t": [], "pred_x_0": []})
        it = tqdm(indices, desc="DDPM Sampling", disable=not verbose) if tqdm is not None else indices
        for i in it:
            out = self.sample_once(img, i, classes, clip_denoised, **kwargs)
            img = out.pred_x_prev
            if keep:
                ret.pred_x_t.append(out.pred_x_prev)
                ret.pred_x_0.append(out.pred_x_0)
        ret.samples = img
        backbone.train()
        return ret
