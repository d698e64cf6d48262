"""ClassifierFreeGuidance (reference: diffusion/frameworks/classifier_free_guidance.py:12-42)."""
import torch

from .gaussian_diffusion import GaussianDiffusion, cfg_branches, cfg_combine


class ClassifierFreeGuidance(GaussianDiffusion):
    def __init__(self, backbone, *, p_uncond=0.1, **kwargs):
        super().__init__(backbone, **kwargs)
        self.p_uncond = p_uncond

    @torch.no_grad()
    def eps_branches(self, x, t, classes=None, strength=3.0, **kwargs):
        if classes is None:
            # reference: (1+s)*eps(None) - s*eps(None); identical branches -> one forward
            return self.backbone(x, t, None), None, 0.0
        return cfg_branches(self.backbone, x, t, classes, strength)

    @torch.no_grad()
    def model_inference(self, x, t, classes=None, strength=3.0, **kwargs):
        return cfg_combine(*self.eps_branches(x, t, classes, strength))
