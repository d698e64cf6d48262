from .gaussian_diffusion import GaussianDiffusion  # noqa: F401
from .classifier_free_guidance import ClassifierFreeGuidance  # noqa: F401
from .inpaint_cfg import InpaintCFG  # noqa: F401
from .sr_cfg import SuperResCFG  # noqa: F401
