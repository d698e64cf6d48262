"""rgbd_3d — GPU depth-warp conditioning (HIP scatter/z-buffer kernels instead of moderngl/OpenGL)."""
from . import camera, utils  # noqa: F401
from .warp import WarpRenderer  # noqa: F401

# the reference's class name: a per-sample renderer is a batch-of-one WarpRenderer
def AggregationRenderer(render_size=128, image_size=128, near=0.01, far=200.0, device=0, max_views=27):
    dev = f"cuda:{device}" if isinstance(device, int) else device
    return WarpRenderer(1, image_size, render_size // image_size, max_views, near, far, dev)
