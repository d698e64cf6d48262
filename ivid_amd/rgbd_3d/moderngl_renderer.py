"""The reference's renderer classes (rgbd_3d/moderngl_renderer.py) on the HIP z-buffer kernels — same constructor
arguments, same `render(...)` signatures, same return containers (host numpy arrays in edicts), no OpenGL/EGL/moderngl.

    AggregationRenderer(render_size, image_size, near, far, device, max_views)      moderngl_renderer.py:151-340
        .render(meshes, colors, modelview, fov=45.0, is_autoregressive=False, verbose=False, tqdm_args={})
    SimpleRenderer(render_size, image_size, near, far, device)                      moderngl_renderer.py:11-148
        .render(mesh, color, modelview, fov=45.0)

`meshes` are the dicts `rgbd_3d.utils.depth_to_mesh` returns (vertices.{position,normal,uv,flag}, faces, modelview):
the height-field meshes of ivid.  They are uploaded into the batched `WarpRenderer`'s buffers (batch of one); the
index buffer is reduced to one byte per quad (`warp.diag_from_faces`), anything that is not a depth_to_mesh grid is
refused.  The sampling driver itself does not go through these per-sample classes: it keeps the whole batch on the
device (`WarpRenderer.add_view / conditions`).
"""
import numpy as np
import torch

from .. import _lib
from ..utils import AttrDict
from . import camera
from .warp import WarpRenderer, diag_from_faces


def _dev(device):
    return f"cuda:{device}" if isinstance(device, int) else device


def _mesh_arrays(mesh, P):
    """Reference mesh dict -> (vertex buffer [P*P, 9] in the reference's VBO order, diag [Q*Q])."""
    vt = mesh["vertices"]
    pos = np.asarray(vt["position"], dtype=np.float32)
    nrm = np.asarray(vt["normal"], dtype=np.float32) if "normal" in vt else np.zeros_like(pos)
    vb = np.concatenate([pos, nrm, np.asarray(vt["uv"], dtype=np.float32),
                         np.asarray(vt["flag"], dtype=np.float32).reshape(-1, 1)], axis=-1)
    S = P - 2
    if pos.shape[0] == P * P:
        return vb, diag_from_faces(mesh["faces"], P)
    if pos.shape[0] == S * S:
        # depth_to_mesh(padding=None): embedded in the padded grid with a ring of COPIES of its border vertices --
        # zero-area triangles, which the rasteriser never draws (the device form of ivid_mesh_build's padding = -2)
        vb = np.pad(vb.reshape(S, S, 9), ((1, 1), (1, 1), (0, 0)), "edge").reshape(P * P, 9)
        diag = np.zeros((P - 1, P - 1), dtype=np.uint8)
        diag[1:-1, 1:-1] = diag_from_faces(mesh["faces"], S).reshape(S - 1, S - 1)
        return vb, diag.ravel()
    raise ValueError(f"mesh has {pos.shape[0]} vertices: expected a depth_to_mesh height field of {S}x{S} (padding=None) "
                     f"or {P}x{P} (padded) vertices")


class AggregationRenderer(object):
    """Drop-in for rgbd_3d.AggregationRenderer: z-buffers every source mesh alone and blends the views per pixel
    (aggregation.vsh/.fsh/.csh restated in csrc/warp.hip).  Stateful like the reference: with is_autoregressive=True
    only the LAST mesh of the list is uploaded, the earlier ones are the buffers of the previous calls
    (moderngl_renderer.py:281-283)."""

    def __init__(self, render_size=128, image_size=128, near=0.01, far=200.0, device=0, max_views=27):
        if render_size % image_size:
            raise ValueError("render_size must be a multiple of image_size (ssaa = render_size // image_size, utils.py:450)")
        self.render_size = render_size
        self.image_size = image_size
        self.near = near
        self.far = far
        self.max_views = max_views
        self.warp = WarpRenderer(1, image_size, render_size // image_size, max_views, near, far, _dev(device))

    def _upload(self, meshes, colors, is_autoregressive):
        P = self.image_size + 2
        if len(meshes) > self.max_views:
            raise _lib.IvidHipError(f"AggregationRenderer holds at most {self.max_views} views")
        for i, mesh in enumerate(meshes):
            if is_autoregressive and i != len(meshes) - 1:
                continue
            vb, diag = _mesh_arrays(mesh, P)
            self.warp.upload_view(i, vb[None], diag[None], np.asarray(colors[i], dtype=np.float32)[None],
                                  camera.as_matrix(mesh["modelview"]))
        self.warp.num_views = len(meshes)

    @torch.no_grad()
    def render_device(self, meshes, colors, modelview, fov=45.0, is_autoregressive=False):
        """render() without the read-back: uploads, rasterises ONE target view and leaves the hi-res buffers on the
        device (the WarpRenderer's color8 / depth / masks): what aggregate_conditions consumes."""
        self._upload(meshes, colors, is_autoregressive)
        return self.warp.render(camera.as_matrix(modelview), fov, want_float_color=True)

    @torch.no_grad()
    def render(self, meshes, colors, modelview, fov=45.0, is_autoregressive=False, verbose=False, tqdm_args={}):
        self._upload(meshes, colors, is_autoregressive)
        mvs = modelview if isinstance(modelview, list) else [modelview]
        ret = []
        for mv in mvs:
            r = self.warp.render(camera.as_matrix(mv), fov, want_float_color=True)
            torch.cuda.synchronize(self.warp.device)
            ret.append(AttrDict(color=r.color[0].cpu().numpy(),
                                depth=r.depth[0].cpu().numpy()[..., None],
                                mask_color=r.mask_color[0].cpu().numpy().astype(bool)[..., None],
                                mask_depth=r.mask_depth[0].cpu().numpy().astype(bool)[..., None]))
        return ret if len(ret) > 1 else ret[0]


class SimpleRenderer(object):
    """Drop-in for rgbd_3d.SimpleRenderer (the training-time warp renderer, datasets/base.py:219): one textured mesh,
    one depth-tested draw; colour = NEAREST texel, alpha = 0 on back faces and on discontinuity edges
    (simple.vsh / simple.fsh), depth = linearised window depth (far where nothing was drawn)."""

    def __init__(self, render_size=128, image_size=128, near=0.01, far=200.0, device=0):
        if render_size % image_size:
            raise ValueError("render_size must be a multiple of image_size")
        self.render_size = render_size
        self.image_size = image_size
        self.near = near
        self.far = far
        self.warp = WarpRenderer(1, image_size, render_size // image_size, 1, near, far, _dev(device))

    @torch.no_grad()
    def render_device(self, modelview, fov=45.0):
        """The mesh currently stored in view slot 0 -> (color fp32 [1,R,R,3], depth fp32 [1,R,R], mask u8 [1,R,R])."""
        return self.warp.simple_render(camera.as_matrix(modelview), fov)

    @torch.no_grad()
    def render(self, mesh, color, modelview, fov=45.0):
        P = self.image_size + 2
        vb, diag = _mesh_arrays(mesh, P)
        mv_src = mesh["modelview"] if mesh.get("modelview") is not None else np.eye(4, dtype=np.float32)
        self.warp.upload_view(0, vb[None], diag[None], np.asarray(color, dtype=np.float32)[None], camera.as_matrix(mv_src))
        mvs = modelview if isinstance(modelview, list) else [modelview]
        ret = []
        for mv in mvs:
            r = self.render_device(mv, fov)
            torch.cuda.synchronize(self.warp.device)
            ret.append(AttrDict(color=r.color[0].cpu().numpy(), depth=r.depth[0].cpu().numpy()[..., None],
                                mask=r.mask[0].cpu().numpy().astype(bool)[..., None]))
        return ret if len(ret) > 1 else ret[0]
