"""GaussianDiffusion framework (reference: diffusion/frameworks/gaussian_diffusion.py:12-116).

Inference surface only: constructor, `betas` / `timesteps` / `backbone` / `backbone_args`,
`model_inference`, plus `diffuse` / `reverse_diffuse`.  `training_losses` is training-only and out
of scope.

Extension used by the fused samplers: `eps_branches(...) -> (eps_cond, eps_uncond | None, strength)`
returns the classifier-free-guidance branches UNCOMBINED so the DDIM/DDPM step kernel fuses
`(1+s)*eps_c - s*eps_u` (classifier_free_guidance.py:39-42) into the update instead of running it
as separate elementwise passes.
"""
import inspect

import numpy as np
import torch

from ...utils import AttrDict
from .utils import get_betas_by_name


class GaussianDiffusion:
    def __init__(self, backbone, timesteps=1000, beta_schedule="linear"):
        self.backbone = backbone
        self.timesteps = timesteps
        self.beta_schedule = beta_schedule
        fwd = self.backbone.module.forward if hasattr(self.backbone, "module") else self.backbone.forward
        self.backbone_args = AttrDict(inspect.signature(fwd).parameters)
        betas = get_betas_by_name(beta_schedule, timesteps).astype(np.float64)
        assert betas.ndim == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all(), "betas must be in (0, 1]"
        self.betas = betas
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)

    # ---- q(x_t | x_0) helpers (gaussian_diffusion.py:45-74); batch-uniform or per-sample t ----
    def _tab(self, arr, t, like):
        v = torch.from_numpy(arr).to(like.device)[t].float()
        return v.view(-1, *([1] * (like.dim() - 1)))

    def diffuse(self, x_0, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_0)
        assert noise.shape == x_0.shape, "noise must have same shape as x_0"
        return self._tab(self.sqrt_alphas_cumprod, t, x_0) * x_0 + self._tab(self.sqrt_one_minus_alphas_cumprod, t, x_0) * noise

    def reverse_diffuse(self, x_t, t, noise):
        assert noise.shape == x_t.shape, "noise must have same shape as x_t"
        return (x_t - self._tab(self.sqrt_one_minus_alphas_cumprod, t, x_t) * noise) / self._tab(self.sqrt_alphas_cumprod, t, x_t)

    def _filter(self, kwargs):
        return {k: v for k, v in kwargs.items() if k in self.backbone_args}

    @torch.no_grad()
    def model_inference(self, x, t, classes=None, **kwargs):
        return self.backbone(x, t, classes, **self._filter(kwargs))

    @torch.no_grad()
    def eps_branches(self, x, t, classes=None, **kwargs):
        return self.backbone(x, t, classes, **self._filter(kwargs)), None, 0.0


def cfg_branches(backbone, x, t, classes, strength):
    """(eps_cond, eps_uncond, strength) with one stacked forward when the backbone supports it."""
    if classes is None:
        return backbone(x, t, None), None, 0.0
    if not strength > 0:
        # classifier_free_guidance.py:39-42: (1 + s) * eps_c - (s * eps_u if s > 0 else 0): no second forward, but the
        # conditional branch is still scaled by (1 + s) for negative strengths
        ec = backbone(x, t, classes)
        return (ec if strength == 0 else (1 + strength) * ec), None, 0.0
    if hasattr(backbone, "forward_cfg"):
        if hasattr(backbone, "note_guidance"):      # the guidance-aware precision tier (AdmUnet2d.note_guidance)
            backbone.note_guidance(strength)
        ec, eu = backbone.forward_cfg(x, t, classes)
        return ec, eu, float(strength)
    return backbone(x, t, classes), backbone(x, t, None), float(strength)


def cfg_combine(ec, eu, strength):
    if eu is None:
        return ec
    return (1 + strength) * ec - strength * eu  # classifier_free_guidance.py:39-42 (compat path)
