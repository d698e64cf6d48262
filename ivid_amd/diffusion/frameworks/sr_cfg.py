"""SuperResCFG (reference: diffusion/frameworks/sr_cfg.py:11-60): condition = bilinear x(S/s)
upsample of the low-resolution RGBD (align_corners=False) concatenated behind x."""
import torch
import torch.nn.functional as F

from .gaussian_diffusion import GaussianDiffusion, cfg_branches, cfg_combine


class SuperResCFG(GaussianDiffusion):
    def __init__(self, backbone, *, p_uncond=0.1, **kwargs):
        super().__init__(backbone, **kwargs)
        self.p_uncond = p_uncond

    def make_cond_inputs(self, x, y, **kwargs):
        # sr_cfg.py:31-36.  The resize runs once per step on [B,4,s,s]; it is plumbing next to the UNet.
        scale = x.shape[-1] // y.shape[-1]
        y = F.interpolate(y, scale_factor=scale, mode="bilinear", align_corners=False)
        return torch.cat([x, y], dim=1)

    @torch.no_grad()
    def eps_branches(self, x, t, y, classes=None, strength=3.0, **kwargs):
        cond = self.make_cond_inputs(x, y, **kwargs)
        if classes is None:
            return self.backbone(cond, t, None), None, 0.0
        return cfg_branches(self.backbone, cond, t, classes, strength)

    @torch.no_grad()
    def model_inference(self, x, t, y, classes=None, strength=3.0, **kwargs):
        return cfg_combine(*self.eps_branches(x, t, y, classes, strength, **kwargs))
