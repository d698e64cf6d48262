"""rgbd_3d.utils — the reference's functions (rgbd_3d/utils.py), same names / arguments / return containers, on top of
the HIP warp kernels.

`linearize_depth` / `project_depth` / `to8b` are host numpy helpers with the reference's semantics.  `depth_to_mesh`,
`aggregate_conditions` and `forward_backward_warp` keep the reference's per-sample signatures (host numpy in, edict of
host numpy out) so that the reference's own `sample_all` (inference/sample.py:87-139) and `WarpDataset`
(datasets/base.py:219-238) run against them unchanged; the work itself happens on the GPU (batch of one).  The batched,
device-resident API the sampling driver of this package uses is `rgbd_3d.WarpRenderer`.
"""
import numpy as np
import torch

from ..utils import AttrDict
from . import camera
from .warp import WarpRenderer

_mesh_builders = {}


def to8b(x):
    """utils.py:34-35."""
    return (np.clip(x, 0, 1) * 255).astype(np.uint8)


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


def _builder(S, device="cuda"):
    """One cached batch-of-one WarpRenderer per (image size, RESOLVED device): the mesh-building scratch of depth_to_mesh.
    'cuda' means the current device at the time of the call, so a process / thread that switched GPUs gets its own."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (S, str(dev))
    if key not in _mesh_builders:
        _mesh_builders[key] = WarpRenderer(1, S, 1, 1, device=dev)
    return _mesh_builders[key]


def clear_mesh_builders():
    """Release the cached mesh-building renderers (their device buffers)."""
    _mesh_builders.clear()


def depth_to_mesh(depth, padding=None, fov=45, modelview=None, atol=None, rtol=None, erode_rgb=None, cal_normal=False):
    """utils.py:144-260.  depth: LINEARISED (metric) depth [S,S,1]; padding None | 'frustum' | pixels; atol = rtol = None
    switches the discontinuity test off (:223); returns the reference's mesh edict (host numpy; `vertices.normal` only
    with cal_normal).  Built by ivid_mesh_build; positions / normals are transformed by inverse(modelview) as in :232-237."""
    depth = np.asarray(depth, dtype=np.float32)
    S = depth.shape[0]
    rgbd = np.zeros((1, 4, S, S), dtype=np.float32)
    rgbd[0, 3] = depth[..., 0]
    r = _builder(S)
    r.reset()
    r.add_view(torch.from_numpy(rgbd).to(r.device), camera.as_matrix(modelview), fov, atol=atol, rtol=rtol, erode_rgb=erode_rgb,
               padding=padding, metric=True)
    m = r.mesh_numpy(0, 0, unpadded=padding is None)
    if not cal_normal:
        del m.vertices["normal"]
    m["depth"], m["fov"], m["modelview"] = depth, fov, modelview
    return m


def depth_edge(depth, atol=0.02, rtol=0.02):
    """utils.py:311-332 (host numpy; the device version runs inside ivid_warp_resolve)."""
    def dd(x, y):
        x, y = np.maximum(x, 1e-6), np.maximum(y, 1e-6)
        return np.logical_and(np.abs(x - y) > atol, np.abs(1 / x - 1 / y) > rtol)
    cnt = np.zeros((depth.shape[0], depth.shape[1], 1), dtype=np.uint8)
    m = dd(depth[:, 1:], depth[:, :-1]); cnt[:, 1:] += m; cnt[:, :-1] += m
    m = dd(depth[1:, :], depth[:-1, :]); cnt[1:, :] += m; cnt[:-1, :] += m
    m = dd(depth[1:, 1:], depth[:-1, :-1]); cnt[1:, 1:] += m; cnt[:-1, :-1] += m
    m = dd(depth[1:, :-1], depth[:-1, 1:]); cnt[1:, :-1] += m; cnt[:-1, 1:] += m
    return cnt < 3


def _resolve_buffers(warp, res):
    """A render result in the REFERENCE's form (host numpy: color float [R,R,3], depth [R,R,1], mask_color / mask_depth
    bool [R,R,1]) -> the device buffers ivid_warp_resolve reads."""
    dev = warp.device
    return dict(color8=torch.from_numpy(to8b(np.asarray(res["color"]))[None]).to(dev).contiguous(),
                depth=torch.from_numpy(np.asarray(res["depth"], dtype=np.float32)[None, ..., 0]).to(dev).contiguous(),
                mask_color=torch.from_numpy(np.asarray(res["mask_color"]).astype(np.uint8)[None, ..., 0]).to(dev).contiguous(),
                mask_depth=torch.from_numpy(np.asarray(res["mask_depth"]).astype(np.uint8)[None, ..., 0]).to(dev).contiguous())


def aggregate_conditions(renderer, meshes, colors, modelview, fov=45, near=0.5, mode="z_buffer", far=100, atol=0.02,
                         rtol=0.02, erode_rgb=2):
    """utils.py:420-477: render the source `meshes` / `colors` into the camera `modelview` with `renderer`
    (is_autoregressive=True: only the last mesh is new) and resolve the conditioning images.

    renderer = this package's AggregationRenderer: everything stays on the device between the rasteriser and the
    resolve kernels.  Any other object with the reference's `.render_size` / `.render(...)` contract works too (its
    host buffers are uploaded for the resolve) — that is how the resolve is pinned to the reference in the tests."""
    if mode != "z_buffer":
        raise NotImplementedError("aggregate_conditions: only mode='z_buffer' (what inference/sample.py uses)")
    image_size = np.asarray(colors[0]).shape[0]
    from .moderngl_renderer import AggregationRenderer
    if isinstance(renderer, AggregationRenderer):
        renderer.render_device(meshes, colors, modelview, fov, is_autoregressive=True)
        warp, buffers = renderer.warp, None
    else:
        ssaa = renderer.render_size // image_size
        res = renderer.render(meshes, colors, modelview, fov, is_autoregressive=True)
        key = ("resolve", image_size, ssaa)
        if key not in _mesh_builders:
            _mesh_builders[key] = WarpRenderer(1, image_size, ssaa, 1)
        warp = _mesh_builders[key]
        buffers = _resolve_buffers(warp, res)
    c = warp.resolve(near, far, atol, rtol, erode_rgb, buffers=buffers)
    torch.cuda.synchronize(warp.device)
    hw = lambda t: t[0].permute(1, 2, 0).cpu().numpy()
    return AttrDict(color=hw(c.color).astype(np.float64), depth=hw(c.depth), mask=hw(c.mask), mask_rgb=hw(c.mask_rgb),
                    depth_convex=hw(c.depth_convex))


def forward_backward_warp(renderer, rgbd, modelview1, modelview0=None, padding=None, fov=45, near=0.5, far=100,
                          mode="z_buffer", atol=0.02, rtol=0.02):
    """utils.py:335-417 — the training-time augmentation of WarpDataset (datasets/base.py:238): lift `rgbd` [S,S,4]
    (RGB and encoded depth in [0,1]) to a mesh in view 0, render it from view 1, lift THAT to a mesh, render it back
    into view 0: the round trip leaves exactly the holes / edge artefacts the conditional model has to inpaint.
    `renderer` = this package's SimpleRenderer; all four stages run on the device, the result is host numpy."""
    from .moderngl_renderer import SimpleRenderer
    if not isinstance(renderer, SimpleRenderer):
        raise TypeError("forward_backward_warp needs ivid_amd.rgbd_3d.SimpleRenderer")
    rgbd = np.asarray(rgbd, dtype=np.float32)
    S = rgbd.shape[0]
    w = renderer.warp
    ssaa = renderer.render_size // S
    off = (ssaa - 1) // 2
    mv0 = camera.look_at((0.0, 0.0, 1.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)) if modelview0 is None else camera.as_matrix(modelview0)
    mv1 = camera.as_matrix(modelview1)
    # backproject view 0 (no discontinuity test: atol = rtol = None), render from view 1
    lin = linearize_depth(rgbd[:, :, 3:], near, far, mode).astype(np.float32)
    src = np.concatenate([rgbd[:, :, :3], lin], axis=-1).transpose(2, 0, 1)[None]
    w.reset()
    w.add_view(torch.from_numpy(np.ascontiguousarray(src)).to(w.device), mv0, fov, atol=None, rtol=None, erode_rgb=None,
               padding=padding, metric=True)
    r1 = w.simple_render(mv1, fov)
    color1 = w.lanczos8(r1.color).float() / 255.0                       # [1,S,S,3]
    depth1 = r1.depth[:, off::ssaa, off::ssaa]                          # metric, centre sub-pixel
    # backproject view 1 (padding=None, discontinuities flagged), render from view 0
    src1 = torch.cat([color1.permute(0, 3, 1, 2), depth1[:, None]], dim=1).contiguous()
    w.reset()
    w.add_view(src1, mv1, fov, atol=atol, rtol=rtol, erode_rgb=None, padding=None, metric=True)
    r0 = w.simple_render(mv0, fov)
    c8 = w.lanczos8(r0.color)
    torch.cuda.synchronize(w.device)
    # depth / mask as utils.py:402-409 (host numpy on three small arrays)
    depth = project_depth(r0.depth[0, off::ssaa, off::ssaa].cpu().numpy()[..., None], near, far, mode)
    mask = r0.mask[0].cpu().numpy().astype(bool).reshape(S, ssaa, S, ssaa, 1).sum(axis=(1, 3)) > 0.75 * ssaa ** 2
    mask &= depth_edge(depth, atol=atol, rtol=rtol)
    color = c8[0].cpu().numpy() / 255.0
    return AttrDict(color=color * mask, depth=depth * mask, mask=mask.astype(np.float32))
