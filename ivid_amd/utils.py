"""Small host-side helpers shared by the ivid_amd Python surface."""


class AttrDict(dict):
    """dict with attribute access — stand-in for easydict.EasyDict, which the reference uses as its
    result container (ddim.py:7, gaussian_diffusion.py:7) and which is not a dependency here."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def default_noise(shape, device):
    """Gaussian noise on the device (torch Philox).  Parity runs inject CPU-generator noise instead
    through the `noise_fn` hooks (SURVEY.md §7 hard part 6)."""
    import torch
    return torch.randn(shape, device=device)
