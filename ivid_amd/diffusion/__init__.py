from . import backbones, frameworks, samplers  # noqa: F401
