"""Pin the warp oracle's mesh / depth / edge restatements to the LIVE reference code.

Run in the build container only.  Imports /root/reference/rgbd_3d/utils.py itself, with stand-in modules for
its missing third-party imports (none of which takes part in the arithmetic that is pinned, except two):
  glm      -> `inverse` / `mat3` on plain numpy 4x4 math matrices (the reference only multiplies with them),
  cv2      -> `erode` = min filter whose border never erodes (cv2's default border value is +inf),
  plyfile, moderngl -> empty (unused on this path).
Stores the reference's outputs for synthetic depth maps in tests/golden/warp_mesh.npz (meshes: frustum / numeric /
no padding, with and without the discontinuity test), tests/golden/warp_resolve.npz (the reference's OWN
aggregate_conditions run on a stub renderer that returns stored hi-res buffers: pins the SSAA resolve) and
tests/golden/warp_fbw.npz (the reference's OWN forward_backward_warp run on a stub SimpleRenderer whose rasteriser is
oracle/warp_raster.c: pins everything of that function but the rasteriser).
"""
import importlib.util
import os
import sys
import types

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
    __setattr__ = dict.__setitem__


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


_mod("easydict", EasyDict=EasyDict)
_mod("glm", inverse=lambda m: np.linalg.inv(np.asarray(m, np.float32)).astype(np.float32), mat3=lambda m: np.asarray(m)[:3, :3])


def _erode(img, kernel, iterations=1):
    """cv2.erode with a ones kernel: min filter whose border never erodes (cv2's default border value is +inf)."""
    out = ndimage.minimum_filter(np.asarray(img, np.float64), size=kernel.shape, mode="constant", cval=np.inf)
    return out.astype(np.asarray(img).dtype)


_mod("cv2", erode=_erode)
_mod("plyfile")
spec = importlib.util.spec_from_file_location("ref_rgbd_utils", "/root/reference/rgbd_3d/utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

from oracle import warp_oracle as W  # noqa: E402


def synthetic_rgbd(S, seed):
    """Smooth surface + a step discontinuity + random colours, encoded like the network output ([-1,1])."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:S, 0:S] / (S - 1.0)
    z = 1.0 + 0.25 * np.exp(-((xx - 0.45) ** 2 + (yy - 0.55) ** 2) / 0.05) + 0.05 * np.sin(7 * xx) * np.cos(5 * yy)
    z[int(0.6 * S):, int(0.55 * S):] += 0.8                      # a foreground/background step
    d01 = W.project_depth(z, 0.6, 5.0)
    rgbd = np.concatenate([rng.uniform(0, 1, (S, S, 3)), d01[..., None]], -1).astype(np.float32)
    return (rgbd * 2 - 1).transpose(2, 0, 1)[None].copy()        # [1,4,S,S]


out = {}
for S, seed, yaw, pitch in [(16, 0, 0.0, 0.0), (32, 1, 0.3, -0.15)]:
    rgbd = synthetic_rgbd(S, seed)
    hw = rgbd[0].transpose(1, 2, 0) * 0.5 + 0.5                  # sample.py:83
    depth_lin = ref.linearize_depth(hw[:, :, 3:], 0.6, 5.0)      # sample.py:129-130
    mv = W.look_at((np.sin(yaw) * np.cos(pitch), np.sin(pitch), np.cos(yaw) * np.cos(pitch)), (0, 0, 0), (0, 1, 0))
    mesh = ref.depth_to_mesh(depth_lin, padding="frustum", fov=45, modelview=mv, atol=0.03, rtol=0.03, erode_rgb=3, cal_normal=True)
    mine = W.depth_to_mesh(depth_lin, 45, mv, 0.03, 0.03, 3)
    vb = np.concatenate([mesh.vertices.position, mesh.vertices.normal, mesh.vertices.uv, mesh.vertices.flag], -1).astype(np.float32)
    assert np.array_equal(mesh.faces, mine["faces"])
    assert np.array_equal(mesh.vertices.flag[:, 0], mine["verts"][:, 8])
    err = np.abs(vb - mine["verts"]).max()
    print(f"S={S}: oracle vs reference vertex buffer max abs diff {err:.2e}; flags {np.unique(mesh.vertices.flag)}")
    assert err < 1e-6
    dproj = ref.project_depth(depth_lin.astype(np.float32), 0.6, 5.0)
    edge = ref.depth_edge(dproj, atol=0.03, rtol=0.03)
    assert np.array_equal(edge, W.depth_edge(dproj, 0.03, 0.03))
    # the mesh load_scene rebuilds from a stored scene (inference/utils.py:108-111: numeric padding 32, metric depth)
    mesh32 = ref.depth_to_mesh(depth_lin.astype(np.float32), 32, 45, mv, atol=0.03, rtol=0.03, erode_rgb=3, cal_normal=True)
    out[f"vbo_pad32_{S}"] = np.concatenate([mesh32.vertices.position, mesh32.vertices.normal, mesh32.vertices.uv,
                                            mesh32.vertices.flag], -1).astype(np.float32)
    out[f"faces_pad32_{S}"] = mesh32.faces.astype(np.int32)
    out[f"rgbd_{S}"] = rgbd
    out[f"modelview_{S}"] = mv
    out[f"vbo_{S}"] = vb
    out[f"faces_{S}"] = mesh.faces.astype(np.int32)
    out[f"depth_lin_{S}"] = depth_lin
    out[f"edge_{S}"] = edge
    # forward_backward_warp's two meshes (utils.py:374-398): numeric padding = image size without the discontinuity test
    # and without normals; no padding at all with the test
    meshS = ref.depth_to_mesh(depth_lin.astype(np.float32), S, 45, mv, atol=None, rtol=None)
    out[f"vbo_padS_{S}"] = np.concatenate([meshS.vertices.position, meshS.vertices.uv, meshS.vertices.flag], -1).astype(np.float32)
    out[f"faces_padS_{S}"] = meshS.faces.astype(np.int32)
    mesh0 = ref.depth_to_mesh(depth_lin.astype(np.float32), None, 45, mv, atol=0.03, rtol=0.03)
    out[f"vbo_nopad_{S}"] = np.concatenate([mesh0.vertices.position, mesh0.vertices.uv, mesh0.vertices.flag], -1).astype(np.float32)
    out[f"faces_nopad_{S}"] = mesh0.faces.astype(np.int32)
np.savez_compressed(os.path.join(HERE, "warp_mesh.npz"), **out)
print("wrote warp_mesh.npz")

# ---------------------------------------------------------------------------------------------------------------------
# aggregate_conditions of the reference itself on a stub renderer (the GL part replaced by stored buffers)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import warp_common as WC  # noqa: E402


class StubAggregationRenderer:
    """What aggregate_conditions needs of a renderer: .render_size and .render(...) -> edict (utils.py:450-453)."""

    def __init__(self, render_size, res):
        self.render_size, self.res = render_size, res

    def render(self, meshes, colors, modelview, fov, is_autoregressive=False):
        return EasyDict(self.res)


res_out = {}
for tag, S, ssaa, views, target, erode in [("S32x3", 32, 3, [(0.0, 0.0), (0.3, 0.0)], (0.15, 0.15), 3),
                                           ("S16x5", 16, 5, [(0.0, 0.0)], (0.45, -0.15), 2),
                                           ("S32x3_wide", 32, 3, [(0.0, 0.0), (-0.3, 0.15), (0.6, 0.0)], (-0.6, -0.15), 1)]:
    R = S * ssaa
    meshes, cols = zip(*[WC.oracle_mesh(WC.synthetic_rgbd(S, 20 + i, layers=(i % 2 == 1))[0], WC.orbit(*v)) for i, v in enumerate(views)])
    hi = W.render(list(meshes), list(cols), WC.orbit(*target), 45, S, R)           # any plausible hi-res buffer will do
    hi = {k: hi[k] for k in ("color", "depth", "mask_color", "mask_depth")}
    o = ref.aggregate_conditions(StubAggregationRenderer(R, hi), None, [np.zeros((S, S, 3))], None, fov=45, near=0.6, far=5,
                                 atol=0.03, rtol=0.03, erode_rgb=erode)
    mine = W.resolve(hi, S, ssaa, 0.6, 5.0, 0.03, 0.03, erode)
    for k in ("color", "depth", "mask", "mask_rgb", "depth_convex"):
        assert np.array_equal(np.asarray(o[k], np.float64), np.asarray(mine[k], np.float64)), (tag, k)
        res_out[f"{tag}_out_{k}"] = np.asarray(o[k])
    for k, v in hi.items():
        res_out[f"{tag}_in_{k}"] = v
    res_out[f"{tag}_cfg"] = np.array([S, ssaa, erode])
    print(f"resolve {tag}: oracle == reference; mask {o.mask.mean():.3f} mask_rgb {o.mask_rgb.mean():.3f}")
np.savez_compressed(os.path.join(HERE, "warp_resolve.npz"), **res_out)
print("wrote warp_resolve.npz")

# ---------------------------------------------------------------------------------------------------------------------
# forward_backward_warp of the reference itself on a stub SimpleRenderer (rasteriser = oracle/warp_raster.c)


class StubSimpleRenderer:
    def __init__(self, render_size, image_size, near, far):
        self.render_size, self.image_size, self.near, self.far = render_size, image_size, near, far

    def render(self, mesh, color, modelview, fov=45.0):
        S = self.image_size
        m = W.from_reference_mesh(mesh, S)
        return EasyDict(W.simple_render(m, color, modelview, fov, S, self.render_size, self.near, self.far))


fbw = {}
for tag, S, seed, yaw, pitch in [("S32", 32, 31, 0.2, -0.1), ("S64", 64, 32, -0.25, 0.12)]:
    hw = WC.synthetic_rgbd(S, seed, smooth_color=True)[0].transpose(1, 2, 0) * 0.5 + 0.5       # [S,S,4] in [0,1]
    mv0, mv1 = WC.orbit(0.0, 0.0), WC.orbit(yaw, pitch)
    o = ref.forward_backward_warp(StubSimpleRenderer(3 * S, S, 0.1, 200.0), hw.astype(np.float64), mv1, mv0, padding=S,
                                  fov=45, near=0.6, far=5.0, atol=0.02, rtol=0.02)
    fbw[f"{tag}_rgbd"] = hw.astype(np.float32)
    fbw[f"{tag}_mv1"] = mv1
    for k in ("color", "depth", "mask"):
        fbw[f"{tag}_{k}"] = np.asarray(o[k], np.float32)
    print(f"forward_backward_warp {tag}: mask {o.mask.mean():.3f}")
np.savez_compressed(os.path.join(HERE, "warp_fbw.npz"), **fbw)
print("wrote warp_fbw.npz")
