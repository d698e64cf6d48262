"""rgbd_3d.utils — the reference's function names (rgbd_3d/utils.py) on top of the HIP warp kernels.

`linearize_depth` / `project_depth` are host numpy helpers with the reference's semantics; `depth_to_mesh`
and `aggregate_conditions` keep the reference's per-sample signatures for drop-in use but run on the GPU
(batch of one).  The batched, device-resident API the sampling driver uses is `rgbd_3d.WarpRenderer`.
"""
import numpy as np
import torch

from ..utils import AttrDict
from .warp import WarpRenderer


def linearize_depth(depth, near=0.5, far=100, mode="z_buffer"):
    """utils.py:38-58."""
    if mode == "z_buffer":
        depth = np.clip(depth, 1e-6, 1.0 - 1e-6)
        return near * far / (far - (far - near) * depth)
    if mode == "linear":
        return near + (far - near) * depth
    return depth


def project_depth(depth, near=0.5, far=100, mode="z_buffer"):
    """utils.py:61-67."""
    if mode == "z_buffer":
        depth = np.clip(depth, near, far)
        return (1 / near - 1 / depth) / (1 / near - 1 / far)
    if mode == "linear":
        return (depth - near) / (far - near)
    return depth


def depth_to_mesh(depth, padding="frustum", fov=45, modelview=None, atol=None, rtol=None, erode_rgb=None, cal_normal=True,
                  near=0.6, far=5.0):
    """utils.py:144-260 for the configuration ivid samples with (padding='frustum', cal_normal=True): depth is the
    LINEARISED [S,S,1] depth; returns the reference's mesh dict (host numpy) built by the HIP kernel."""
    if padding != "frustum" or not cal_normal:
        raise NotImplementedError("only padding='frustum', cal_normal=True (inference/sample.py:128-138) is implemented")
    S = depth.shape[0]
    z = project_depth(np.asarray(depth, dtype=np.float32), near, far)          # back to the network's encoding
    rgbd = np.zeros((1, 4, S, S), dtype=np.float32)
    rgbd[0, 3] = z[..., 0] * 2 - 1
    r = WarpRenderer(1, S, 1, 1)
    r.add_view(torch.from_numpy(rgbd).cuda(), np.eye(4, dtype=np.float32) if modelview is None else modelview, fov, near,
               far, atol, rtol, erode_rgb)
    m = r.mesh_numpy(0, 0)
    m["depth"], m["fov"] = depth, fov
    return m


def aggregate_conditions(renderer, meshes, colors, modelview, fov=45, near=0.5, mode="z_buffer", far=100, atol=0.02,
                         rtol=0.02, erode_rgb=2):
    """utils.py:420-477 signature.  `renderer` must be a WarpRenderer that already holds the source views (meshes /
    colors are implied by its state and ignored): returns host numpy arrays [S,S,C] like the reference."""
    c = renderer.conditions(modelview, fov, near, far, atol, rtol, erode_rgb)
    hw = lambda t: t[0].permute(1, 2, 0).cpu().numpy()
    return AttrDict(color=hw(c.color).astype(np.float64), depth=hw(c.depth), mask=hw(c.mask), mask_rgb=hw(c.mask_rgb),
                    depth_convex=hw(c.depth_convex))
