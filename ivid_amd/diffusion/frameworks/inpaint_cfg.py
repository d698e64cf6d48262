"""InpaintCFG (reference: diffusion/frameworks/inpaint_cfg.py:11-83): the conditional model sees
cat[x, mask_rgb, y_rgb*m_rgb + N(0,1)*(1-m_rgb), y_d*m + N(0,1)*(1-m), mask] with FRESH noise in
the holes at every step, shared by both guidance branches."""
import torch

from ... import _lib
from ...utils import default_noise
from .gaussian_diffusion import GaussianDiffusion, cfg_branches, cfg_combine


class InpaintCFG(GaussianDiffusion):
    def __init__(self, backbone, *, p_uncond=0.1, p_uncond_img=0.0, **kwargs):
        super().__init__(backbone, **kwargs)
        self.p_uncond = p_uncond
        self.p_uncond_img = p_uncond_img

    def make_cond_inputs(self, x, y, mask, **kwargs):
        """inpaint_cfg.py:24-49 as one HIP kernel.  Draw order (rgb noise, then depth noise) matches
        the reference; `noise_fn(shape)` in kwargs overrides the on-device generator."""
        noise_fn = kwargs.get("noise_fn") or (lambda shape: default_noise(shape, x.device))
        mask_rgb = kwargs.get("mask_rgb")
        b, _, h, w = x.shape
        n_rgb = noise_fn((b, 3, h, w)).float().contiguous()
        n_d = noise_fn((b, 1, h, w)).float().contiguous()
        out = torch.empty(b, 10 if mask_rgb is not None else 9, h, w, dtype=torch.float32, device=x.device)
        # the kernel reads fixed layouts through raw pointers: broadcast exactly like the reference's tensor expressions
        # would (e.g. a [1,1,H,W] mask), refuse anything else instead of reading out of bounds
        x = x.float().contiguous()
        y = y.float().expand(b, 4, h, w).contiguous()
        mask = mask.float().expand(b, 1, h, w).contiguous()
        mr = mask_rgb.float().expand(b, 1, h, w).contiguous() if mask_rgb is not None else None
        _lib.call("ivid_inpaint_cond", _lib.ptr(x), _lib.ptr(y), _lib.ptr(mask), _lib.ptr(mr), _lib.ptr(n_rgb),
                  _lib.ptr(n_d), _lib.ptr(out), b, h * w, torch.cuda.current_stream(x.device).cuda_stream)
        return out

    def make_uncond_inputs(self, x):
        return torch.cat([x, torch.randn_like(x), torch.zeros_like(x[:, :1])], dim=1)

    @torch.no_grad()
    def eps_branches(self, x, t, y, mask, classes=None, strength=3.0, **kwargs):
        cond = self.make_cond_inputs(x, y, mask, **kwargs)
        if classes is None:
            return self.backbone(cond, t, None), None, 0.0
        return cfg_branches(self.backbone, cond, t, classes, strength)

    @torch.no_grad()
    def model_inference(self, x, t, y, mask, classes=None, strength=3.0, **kwargs):
        return cfg_combine(*self.eps_branches(x, t, y, mask, classes, strength, **kwargs))
