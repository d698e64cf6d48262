"""rgbd_3d — GPU depth-warp conditioning (HIP scatter/z-buffer kernels instead of moderngl/OpenGL).

Same public names as the reference package (rgbd_3d/__init__.py): `SimpleRenderer`, `AggregationRenderer`, `utils`;
plus `WarpRenderer`, the batched device-resident renderer the sampling driver uses."""
from . import camera, utils  # noqa: F401
from .moderngl_renderer import AggregationRenderer, SimpleRenderer  # noqa: F401
from .warp import WarpRenderer  # noqa: F401
