"""Device-resident, batched depth-warp renderer: the HIP replacement of the reference's
`AggregationRenderer` + `depth_to_mesh` + `aggregate_conditions` trio (rgbd_3d/moderngl_renderer.py:151-340,
rgbd_3d/utils.py:144-260, 420-477).

The reference keeps one OpenGL renderer PER SAMPLE and, between two views, walks the batch on the CPU:
numpy meshing, VBO/texture upload, one draw + one compute dispatch per source view, three read-backs,
PIL + cv2 post-processing, eight host->device copies (inference/sample.py:87-138) — with the GPU idle.
Here the whole view loop stays on the GPU: `add_view` turns the B freshly sampled RGBD images into B meshes
with one call, `conditions` rasterises all source views of all samples and resolves the conditioning tensors
the conditional sampler needs, all as a handful of launches on the current stream.
"""
import os

import numpy as np
import torch

from .. import _lib
from ..utils import AttrDict
from . import camera
from .resample import lanczos_tables


class WarpRenderer:
    def __init__(self, batch, image_size=128, ssaa=3, max_views=27, near=0.01, far=200.0, device="cuda"):
        self.B, self.S, self.ssaa, self.max_views = batch, image_size, ssaa, max_views
        self.R = image_size * ssaa
        self.render_size, self.image_size = self.R, image_size   # names used by aggregate_conditions (utils.py:450)
        self.near, self.far = float(near), float(far)
        self.device = torch.device(device)
        B, S, R, NV = batch, image_size, self.R, max_views
        P = S + 2
        dev = self.device
        f32, u8 = torch.float32, torch.uint8
        self.verts = torch.zeros(NV, B, P * P, 9, dtype=f32, device=dev)
        self.diag = torch.zeros(NV, B, (P - 1) * (P - 1), dtype=u8, device=dev)
        self.colors = torch.zeros(NV, B, S, S, 3, dtype=f32, device=dev)
        self.campos = torch.zeros(NV, B, 3, dtype=f32, device=dev)
        self.scratch_depth = torch.empty(B, P * P, dtype=f32, device=dev)
        self.scratch_flags = torch.empty(B, P * P, dtype=torch.int32, device=dev)
        self.zbuf = torch.empty(NV, B, R * R, dtype=torch.int64, device=dev)
        # queue of the large triangles (second rasterisation pass): counter + (mesh, triangle) pairs
        # (skirt: 8 (S+1) triangles per mesh, + discontinuity sheets; 4096 per mesh leaves a 3x margin at S = 128 --
        #  an overflowing queue only costs speed, the kernel then walks the triangle in its own thread)
        self.work_cap = int(os.environ.get("IVID_WARP_QUEUE", str(NV * B * max(4096, 32 * (S + 1)))))
        self.work = torch.zeros(2 + 2 * max(self.work_cap, 1), dtype=torch.int32, device=dev)
        self.color8 = torch.empty(B, R, R, 3, dtype=u8, device=dev)
        self.depth_lin = torch.empty(B, R, R, dtype=f32, device=dev)
        self.mask_c = torch.empty(B, R, R, dtype=u8, device=dev)
        self.mask_d = torch.empty(B, R, R, dtype=u8, device=dev)
        self.tmp_h = torch.empty(B, R, S, 3, dtype=u8, device=dev)
        self.tmp_small = torch.empty(B, S, S, 3, dtype=u8, device=dev)
        self.tmp_dproj = torch.empty(B, S, S, dtype=f32, device=dev)
        self.tmp_masks = torch.empty(3, B, S, S, dtype=u8, device=dev)
        bounds, coeffs, self.ksize = lanczos_tables(R, S)
        self.bounds = torch.from_numpy(bounds).to(dev)
        self.coeffs = torch.from_numpy(coeffs).to(dev)
        # float32(float64(i)/255.0): the value `color / 255.0` takes after sample.py's .float() (utils.py:454, sample.py:102)
        self.lut255 = torch.from_numpy((np.arange(256, dtype=np.float64) / 255.0).astype(np.float32)).to(dev)
        self.modelviews = []     # per view: float32 [B,4,4]
        self.num_views = 0

    def reset(self):
        self.modelviews = []
        self.num_views = 0

    def _per_sample(self, mv):
        mv = np.asarray(mv, dtype=np.float32)
        if mv.ndim == 2:
            mv = np.broadcast_to(mv, (self.B, 4, 4))
        assert mv.shape == (self.B, 4, 4)
        return np.ascontiguousarray(mv)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    @torch.no_grad()
    def add_view(self, rgbd, modelview, fov=45.0, near=0.6, far=5.0, atol=0.03, rtol=0.03, erode_rgb=3, padding="frustum",
                 metric=False):
        """Append the meshes of a view.  Default: rgbd [B,4,S,S] is the network output in [-1,1] (device) and the mesh
        gets the frustum skirt (sample.py:129-133).  metric=True: rgbd holds RGB in [0,1] and METRIC depth (a stored
        scene); padding = a number of pixels gives load_scene's numeric skirt (inference/utils.py:108-111)."""
        v = self.num_views
        if v >= self.max_views:
            raise _lib.IvidHipError(f"WarpRenderer holds at most {self.max_views} views")
        assert rgbd.shape == (self.B, 4, self.S, self.S)
        rgbd = rgbd.to(self.device, torch.float32).contiguous()
        mv = self._per_sample(modelview)
        inv = np.stack([camera.inverse(m) for m in mv])
        inv_d = torch.from_numpy(inv.reshape(self.B, 16)).to(self.device)
        self.campos[v].copy_(torch.from_numpy(np.ascontiguousarray(inv[:, :3, 3])))
        pad = -1.0 if padding == "frustum" else float(padding)
        _lib.call("ivid_mesh_build", _lib.ptr(rgbd), self.B, self.S, _lib.ptr(inv_d), float(fov), float(near), float(far),
                  float(atol if atol is not None else 0.0), float(rtol if rtol is not None else 0.0),
                  int(erode_rgb or 0), pad, 1 if metric else 0, _lib.ptr(self.verts[v]), _lib.ptr(self.diag[v]),
                  _lib.ptr(self.colors[v]), _lib.ptr(self.scratch_depth), _lib.ptr(self.scratch_flags), self._stream())
        self.modelviews.append(mv)
        self.num_views += 1

    @torch.no_grad()
    def render(self, modelview, fov=45.0):
        """All stored views -> target camera: returns the 3x-supersampled buffers (device tensors, row 0 = top):
        color8 u8 [B,R,R,3], depth (metric) [B,R,R], mask_color / mask_depth u8 [B,R,R]."""
        assert self.num_views > 0, "no source views"
        mv = self._per_sample(modelview)
        proj = camera.perspective(np.deg2rad(fov), 1.0, self.near, self.far)
        mvp = np.stack([(proj @ m).astype(np.float32) for m in mv]).reshape(self.B, 16)
        mvp_d = torch.from_numpy(np.ascontiguousarray(mvp)).to(self.device)
        _lib.call("ivid_warp_render", _lib.ptr(self.verts), _lib.ptr(self.diag), _lib.ptr(self.colors),
                  _lib.ptr(self.campos), self.num_views, self.B, self.S, _lib.ptr(mvp_d), self.R, self.near, self.far,
                  _lib.ptr(self.zbuf), _lib.ptr(self.color8), _lib.ptr(self.depth_lin), _lib.ptr(self.mask_c),
                  _lib.ptr(self.mask_d), _lib.ptr(self.work), self.work_cap, self._stream())
        return AttrDict(color8=self.color8, depth=self.depth_lin, mask_color=self.mask_c, mask_depth=self.mask_d)

    @torch.no_grad()
    def conditions(self, modelview, fov=45.0, near=0.6, far=5.0, atol=0.03, rtol=0.03, erode_rgb=3):
        """aggregate_conditions for the whole batch: fresh device tensors in [0,1] —
        color [B,3,S,S], depth [B,1,S,S], mask [B,1,S,S], mask_rgb [B,1,S,S], depth_convex [B,1,S,S]."""
        self.render(modelview, fov)
        B, S = self.B, self.S
        mk = lambda c: torch.empty(B, c, S, S, dtype=torch.float32, device=self.device)
        color, depth, mask, mask_rgb, convex = mk(3), mk(1), mk(1), mk(1), mk(1)
        _lib.call("ivid_warp_resolve", _lib.ptr(self.color8), _lib.ptr(self.depth_lin), _lib.ptr(self.mask_c),
                  _lib.ptr(self.mask_d), B, S, self.ssaa, _lib.ptr(self.bounds), _lib.ptr(self.coeffs), self.ksize,
                  _lib.ptr(self.lut255), float(near), float(far), float(atol), float(rtol), int(erode_rgb),
                  _lib.ptr(self.tmp_h), _lib.ptr(self.tmp_small), _lib.ptr(self.tmp_dproj), _lib.ptr(self.tmp_masks),
                  _lib.ptr(color), _lib.ptr(depth), _lib.ptr(mask), _lib.ptr(mask_rgb), _lib.ptr(convex), self._stream())
        return AttrDict(color=color, depth=depth, mask=mask, mask_rgb=mask_rgb, depth_convex=convex)

    def mesh_numpy(self, view, sample):
        """Host copy of one mesh in the reference's layout (depth_to_mesh's return, utils.py:251-258)."""
        P = self.S + 2
        v = self.verts[view, sample].cpu().numpy()
        ft = self.diag[view, sample].cpu().numpy().astype(bool).reshape(P - 1, P - 1)
        idx = np.arange(P * P).reshape(P, P)
        faces = np.stack([idx[:-1, 1:].ravel(), idx[:-1, :-1].ravel(), np.where(ft, idx[1:, 1:], idx[1:, :-1]).ravel(),
                          idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), np.where(ft, idx[:-1, :-1], idx[:-1, 1:]).ravel()],
                         axis=-1).reshape(-1, 3)
        return AttrDict(faces=faces, modelview=self.modelviews[view][sample],
                        vertices=AttrDict(position=v[:, 0:3], normal=v[:, 3:6], uv=v[:, 6:8], flag=v[:, 8:9]))
