from .ddpm import DdpmSampler  # noqa: F401
from .ddim import DdimSampler  # noqa: F401
