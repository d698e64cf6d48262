"""ivid_amd — MI355X-native (gfx950) sampling hot path of JeffreyXiang/ivid.

Mirrors the reference's Python plug-in surface:
    ivid_amd.diffusion.backbones.AdmUnet2d
    ivid_amd.diffusion.frameworks.{GaussianDiffusion, ClassifierFreeGuidance, InpaintCFG, SuperResCFG}
    ivid_amd.diffusion.samplers.{DdimSampler, DdpmSampler}
    ivid_amd.rgbd_3d
and forwards every device op to the C ABI of libivid_hip.so (include/ivid_hip.h).

`ivid_amd.install()` registers these packages under the reference's own top-level names
(`diffusion`, `rgbd_3d`) so that `inference/sample.py`-style drivers import them unchanged.
"""
import sys

__version__ = "0.1.0"


def install():
    """Alias ivid_amd.diffusion -> `diffusion` (and rgbd_3d) in sys.modules (drop-in for the reference)."""
    import importlib
    for name in ("diffusion", "diffusion.backbones", "diffusion.frameworks", "diffusion.samplers", "rgbd_3d"):
        try:
            sys.modules[name] = importlib.import_module("ivid_amd." + name)
        except ModuleNotFoundError:
            pass
