"""Pin the warp oracle's mesh / depth / edge restatements to the LIVE reference code.

Run in the build container only.  Imports /root/reference/rgbd_3d/utils.py itself, with stand-in modules for
its missing third-party imports (none of which takes part in the arithmetic that is pinned, except two):
  glm      -> `inverse` / `mat3` on plain numpy 4x4 math matrices (the reference only multiplies with them),
  cv2      -> `erode` = min filter whose border never erodes (cv2's default border value is +inf),
  plyfile, moderngl -> empty (unused on this path).
Stores the reference's outputs for synthetic depth maps in tests/golden/warp_mesh.npz.
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
_mod("cv2", erode=lambda img, kernel, iterations=1: ndimage.minimum_filter(img, size=kernel.shape, mode="constant", cval=np.inf))
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
np.savez_compressed(os.path.join(HERE, "warp_mesh.npz"), **out)
print("wrote warp_mesh.npz")
