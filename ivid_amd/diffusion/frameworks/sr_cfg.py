"""SuperResCFG (reference: diffusion/frameworks/sr_cfg.py:11-60): condition = bilinear x(S/s)
upsample of the low-resolution RGBD (align_corners=False) concatenated behind x."""
import torch

from ... import _lib
from .gaussian_diffusion import GaussianDiffusion, cfg_branches, cfg_combine


class SuperResCFG(GaussianDiffusion):
    def __init__(self, backbone, *, p_uncond=0.1, **kwargs):
        super().__init__(backbone, **kwargs)
        self.p_uncond = p_uncond

    def make_cond_inputs(self, x, y, **kwargs):
        """sr_cfg.py:31-36 as one HIP kernel (ivid_sr_cond): bilinear upsample of y (align_corners=False) + channel concat."""
        b, cx, S, _ = x.shape
        y = y.float().expand(b, -1, -1, -1).contiguous()
        x = x.float().contiguous()
        out = torch.empty(b, cx + y.shape[1], S, S, dtype=torch.float32, device=x.device)
        _lib.call("ivid_sr_cond", _lib.ptr(x), _lib.ptr(y), _lib.ptr(out), b, cx, y.shape[1], S, y.shape[-1],
                  torch.cuda.current_stream(x.device).cuda_stream)
        return out

    @torch.no_grad()
    def eps_branches(self, x, t, y, classes=None, strength=3.0, **kwargs):
        cond = self.make_cond_inputs(x, y, **kwargs)
        if classes is None:
            return self.backbone(cond, t, None), None, 0.0
        return cfg_branches(self.backbone, cond, t, classes, strength)

    @torch.no_grad()
    def model_inference(self, x, t, y, classes=None, strength=3.0, **kwargs):
        return cfg_combine(*self.eps_branches(x, t, y, classes, strength, **kwargs))
