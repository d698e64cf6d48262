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


def diag_from_faces(faces, P):
    """The reference's index buffer (triangulate, utils.py:113-134) -> one byte per quad: 1 = split along the 00-11
    diagonal.  faces int [(2 (P-1)^2), 3] in triangulate's own order (two triangles per quad, quad-major); anything
    else is not a depth_to_mesh height field and is refused."""
    faces = np.asarray(faces).reshape(-1, 6)
    Q = P - 1
    if faces.shape[0] != Q * Q:
        raise ValueError(f"expected {2 * Q * Q} faces of a {P}x{P} height-field mesh, got {faces.shape[0] * 2}")
    idx = np.arange(P * P).reshape(P, P)
    i00, i01, i10, i11 = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    ft = faces[:, 2] == i11
    ok = ((faces[:, 0] == i01) & (faces[:, 1] == i00) & (faces[:, 2] == np.where(ft, i11, i10)) &
          (faces[:, 3] == i10) & (faces[:, 4] == i11) & (faces[:, 5] == np.where(ft, i00, i01)))
    if not ok.all():
        raise ValueError("faces are not in rgbd_3d.utils.triangulate order")
    return ft.astype(np.uint8)


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
        # queue of the large triangles (second rasterisation pass): counter + one id (mesh * ntri + triangle) per entry.
        # Sized for EVERY triangle (4 B each: 115 MB at NV = 27, B = 32, S = 128): smooth scenes queue ~1150 per mesh (the
        # skirt, 8 (S+1) triangles, + discontinuity sheets), but a noisy depth map makes every triangle a long sliver in
        # the other views, and an overflowing queue falls back to one thread walking a box of up to R^2 pixels.
        self.work_cap = int(os.environ.get("IVID_WARP_QUEUE", str(NV * B * 2 * (P - 1) * (P - 1))))
        self.work = torch.zeros(2 + max(self.work_cap, 1), dtype=torch.int32, device=dev)
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
        self.color_f32 = None    # allocated on first use (reference-contract render() only)

    def reset(self):
        self.modelviews = []
        self.num_views = 0

    def _per_sample(self, mv):
        mv = np.asarray(mv, dtype=np.float32)
        if mv.ndim == 2:
            mv = np.broadcast_to(mv, (self.B, 4, 4))
        assert mv.shape == (self.B, 4, 4)
        return np.ascontiguousarray(mv)

    @staticmethod
    def _tols(atol, rtol):
        """depth_to_mesh (utils.py:223-227): both None = NO discontinuity flagging at all; exactly one None = that one is
        0.  The kernel flags a triangle iff diff > atol AND invdiff > rtol, so +inf switches the test off."""
        if atol is None and rtol is None:
            return float("inf"), float("inf")
        return float(atol if atol is not None else 0.0), float(rtol if rtol is not None else 0.0)

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
        pad = -1.0 if padding == "frustum" else (-2.0 if padding is None else float(padding))   # ivid_hip.h: ivid_mesh_build
        _lib.call("ivid_mesh_build", _lib.ptr(rgbd), self.B, self.S, _lib.ptr(inv_d), float(fov), float(near), float(far),
                  *self._tols(atol, rtol),
                  int(erode_rgb or 0), pad, 1 if metric else 0, _lib.ptr(self.verts[v]), _lib.ptr(self.diag[v]),
                  _lib.ptr(self.colors[v]), _lib.ptr(self.scratch_depth), _lib.ptr(self.scratch_flags), self._stream())
        self.modelviews.append(mv)
        self.num_views += 1

    @torch.no_grad()
    def upload_view(self, view, verts, diag, colors, modelview, sample=None):
        """Store a mesh that was built ELSEWHERE in the reference's layout (AggregationRenderer.render's buffer-object
        writes, moderngl_renderer.py:280-293): verts fp32 [(S+2)^2, 9] = position, normal, uv, flag; diag u8 [(S+1)^2]
        (see `diag_from_faces`); colors fp32 [S,S,3]; modelview 4x4 of the SOURCE camera.  sample = None writes every
        sample of the batch (arrays may carry a leading batch axis), else that one."""
        if view >= self.max_views:
            raise _lib.IvidHipError(f"WarpRenderer holds at most {self.max_views} views")
        sl = slice(None) if sample is None else sample
        dev = self.device
        self.verts[view, sl].copy_(torch.as_tensor(np.ascontiguousarray(verts, dtype=np.float32)).to(dev))
        self.diag[view, sl].copy_(torch.as_tensor(np.ascontiguousarray(diag, dtype=np.uint8)).to(dev))
        self.colors[view, sl].copy_(torch.as_tensor(np.ascontiguousarray(colors, dtype=np.float32)).to(dev))
        mv = self._per_sample(modelview) if sample is None else np.asarray(modelview, dtype=np.float32).reshape(1, 4, 4)
        inv = np.stack([camera.inverse(m) for m in mv])
        self.campos[view, sl].copy_(torch.from_numpy(np.ascontiguousarray(inv[:, :3, 3] if sample is None else inv[0, :3, 3])))
        while len(self.modelviews) <= view:
            self.modelviews.append(np.tile(np.eye(4, dtype=np.float32), (self.B, 1, 1)))
        if sample is None:
            self.modelviews[view] = mv
        else:
            self.modelviews[view][sample] = mv[0]
        self.num_views = max(self.num_views, view + 1)

    @torch.no_grad()
    def render(self, modelview, fov=45.0, want_float_color=False):
        """All stored views -> target camera: returns the 3x-supersampled buffers (device tensors, row 0 = top):
        color8 u8 [B,R,R,3], depth (metric) [B,R,R], mask_color / mask_depth u8 [B,R,R]; want_float_color adds
        color fp32 [B,R,R,3] (the colour before to8b: what the reference's render() returns)."""
        assert self.num_views > 0, "no source views"
        mv = self._per_sample(modelview)
        proj = camera.perspective(np.deg2rad(fov), 1.0, self.near, self.far)
        mvp = np.stack([(proj @ m).astype(np.float32) for m in mv]).reshape(self.B, 16)
        mvp_d = torch.from_numpy(np.ascontiguousarray(mvp)).to(self.device)
        _lib.call("ivid_warp_render", _lib.ptr(self.verts), _lib.ptr(self.diag), _lib.ptr(self.colors),
                  _lib.ptr(self.campos), self.num_views, self.B, self.S, _lib.ptr(mvp_d), self.R, self.near, self.far,
                  _lib.ptr(self.zbuf), _lib.ptr(self.color8), _lib.ptr(self.depth_lin), _lib.ptr(self.mask_c),
                  _lib.ptr(self.mask_d), _lib.ptr(self.work), self.work_cap,
                  _lib.ptr(self._color_f32()) if want_float_color else None, self._stream())
        out = AttrDict(color8=self.color8, depth=self.depth_lin, mask_color=self.mask_c, mask_depth=self.mask_d)
        if want_float_color:
            out["color"] = self.color_f32
        return out

    @torch.no_grad()
    def simple_render(self, modelview, fov=45.0):
        """SimpleRenderer.render (moderngl_renderer.py:96-148) of the meshes in view slot 0: color fp32 [B,R,R,3], depth
        fp32 [B,R,R] (linearised window depth, `far` where empty), mask u8 [B,R,R]."""
        assert self.num_views > 0, "no mesh stored"
        mv = self._per_sample(modelview)
        proj = camera.perspective(np.deg2rad(fov), 1.0, self.near, self.far)
        mvp = np.stack([(proj @ m).astype(np.float32) for m in mv]).reshape(self.B, 16)
        mvp_d = torch.from_numpy(np.ascontiguousarray(mvp)).to(self.device)
        _lib.call("ivid_simple_render", _lib.ptr(self.verts[0]), _lib.ptr(self.diag[0]), _lib.ptr(self.colors[0]), self.B,
                  self.S, _lib.ptr(mvp_d), self.R, self.near, self.far, _lib.ptr(self.zbuf[0]), _lib.ptr(self.work),
                  self.work_cap, _lib.ptr(self._color_f32()), _lib.ptr(self.depth_lin), _lib.ptr(self.mask_d), self._stream())
        return AttrDict(color=self.color_f32, depth=self.depth_lin, mask=self.mask_d)

    @torch.no_grad()
    def lanczos8(self, color_f32):
        """`Image.fromarray(to8b(color)).resize((S, S), LANCZOS)` (utils.py:387,401,454) of fp32 [B,R,R,3] on the device:
        u8 [B,S,S,3], bit-exact with Pillow."""
        out = torch.empty(self.B, self.S, self.S, 3, dtype=torch.uint8, device=self.device)
        _lib.call("ivid_resample8_lanczos", _lib.ptr(color_f32.contiguous()), self.B, self.R, self.S, _lib.ptr(self.bounds),
                  _lib.ptr(self.coeffs), self.ksize, _lib.ptr(self.color8), _lib.ptr(self.tmp_h), _lib.ptr(out), self._stream())
        return out

    def _color_f32(self):
        if getattr(self, "color_f32", None) is None:
            self.color_f32 = torch.empty(self.B, self.R, self.R, 3, dtype=torch.float32, device=self.device)
        return self.color_f32

    @torch.no_grad()
    def resolve(self, near=0.6, far=5.0, atol=0.03, rtol=0.03, erode_rgb=3, buffers=None):
        """aggregate_conditions AFTER its render call (utils.py:454-477) on the hi-res buffers of the last render() --
        or on caller-supplied ones (buffers = dict(color8 u8 [B,R,R,3], depth fp32 [B,R,R], mask_color / mask_depth u8
        [B,R,R]) on the device): 8-bit LANCZOS, centre-sample depth, 7-of-9 masks, depth_edge, erosion."""
        b = buffers if buffers is not None else dict(color8=self.color8, depth=self.depth_lin, mask_color=self.mask_c,
                                                     mask_depth=self.mask_d)
        B, S = self.B, self.S
        mk = lambda c: torch.empty(B, c, S, S, dtype=torch.float32, device=self.device)
        color, depth, mask, mask_rgb, convex = mk(3), mk(1), mk(1), mk(1), mk(1)
        _lib.call("ivid_warp_resolve", _lib.ptr(b["color8"]), _lib.ptr(b["depth"]), _lib.ptr(b["mask_color"]),
                  _lib.ptr(b["mask_depth"]), B, S, self.ssaa, _lib.ptr(self.bounds), _lib.ptr(self.coeffs), self.ksize,
                  _lib.ptr(self.lut255), float(near), float(far), float(atol), float(rtol), int(erode_rgb),
                  _lib.ptr(self.tmp_h), _lib.ptr(self.tmp_small), _lib.ptr(self.tmp_dproj), _lib.ptr(self.tmp_masks),
                  _lib.ptr(color), _lib.ptr(depth), _lib.ptr(mask), _lib.ptr(mask_rgb), _lib.ptr(convex), self._stream())
        return AttrDict(color=color, depth=depth, mask=mask, mask_rgb=mask_rgb, depth_convex=convex)

    @torch.no_grad()
    def conditions(self, modelview, fov=45.0, near=0.6, far=5.0, atol=0.03, rtol=0.03, erode_rgb=3):
        """aggregate_conditions for the whole batch: fresh device tensors in [0,1] —
        color [B,3,S,S], depth [B,1,S,S], mask [B,1,S,S], mask_rgb [B,1,S,S], depth_convex [B,1,S,S]."""
        self.render(modelview, fov)
        return self.resolve(near, far, atol, rtol, erode_rgb)

    def mesh_numpy(self, view, sample, unpadded=False):
        """Host copy of one mesh in the reference's layout (depth_to_mesh's return, utils.py:251-258).  unpadded: the
        mesh was built with padding=None -- return the S x S grid the reference would (the ring of copies is dropped)."""
        if unpadded:
            return self._mesh_numpy_unpadded(view, sample)
        P = self.S + 2
        v = self.verts[view, sample].cpu().numpy()
        ft = self.diag[view, sample].cpu().numpy().astype(bool).reshape(P - 1, P - 1)
        idx = np.arange(P * P).reshape(P, P)
        faces = np.stack([idx[:-1, 1:].ravel(), idx[:-1, :-1].ravel(), np.where(ft, idx[1:, 1:], idx[1:, :-1]).ravel(),
                          idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), np.where(ft, idx[:-1, :-1], idx[:-1, 1:]).ravel()],
                         axis=-1).reshape(-1, 3)
        return AttrDict(faces=faces, modelview=self.modelviews[view][sample],
                        vertices=AttrDict(position=v[:, 0:3], normal=v[:, 3:6], uv=v[:, 6:8], flag=v[:, 8:9]))

    def _mesh_numpy_unpadded(self, view, sample):
        P, S = self.S + 2, self.S
        v = self.verts[view, sample].cpu().numpy().reshape(P, P, 9)[1:-1, 1:-1].reshape(S * S, 9)
        ft = self.diag[view, sample].cpu().numpy().astype(bool).reshape(P - 1, P - 1)[1:-1, 1:-1]
        idx = np.arange(S * S).reshape(S, S)
        faces = np.stack([idx[:-1, 1:].ravel(), idx[:-1, :-1].ravel(), np.where(ft, idx[1:, 1:], idx[1:, :-1]).ravel(),
                          idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), np.where(ft, idx[:-1, :-1], idx[:-1, 1:]).ravel()],
                         axis=-1).reshape(-1, 3)
        return AttrDict(faces=faces, modelview=self.modelviews[view][sample],
                        vertices=AttrDict(position=v[:, 0:3], normal=v[:, 3:6], uv=v[:, 6:8], flag=v[:, 8:9]))
