"""Golden vectors of the depth-warp from REAL OpenGL: the reference's own renderer code and GLSL shaders, run here.

Run in the build container only (`python tests/golden/make_golden_gl.py`; needs `make -C oracle`).  The reference's
rgbd_3d/moderngl_renderer.py needs `moderngl` on an EGL device; this image has neither, but it has Mesa's software
rasteriser.  oracle/glshim/ provides an off-screen llvmpipe OpenGL 4.5 context (through the swrast DRI driver, no X / EGL)
and a stand-in `moderngl` module implementing the API subset the reference uses, so this script IMPORTS AND RUNS, unchanged:
  /root/reference/rgbd_3d/moderngl_renderer.py   AggregationRenderer / SimpleRenderer (GL call sequence)
  /root/reference/rgbd_3d/shaders/*.{vsh,fsh,csh} the shaders, loaded by that file itself
  /root/reference/rgbd_3d/utils.py               depth_to_mesh, aggregate_conditions, forward_backward_warp
with stand-ins only for glm (oracle/glshim/glm.py), cv2.erode, easydict, plyfile.  What it stores:
  warp_gl.npz      per scene of tests/warp_common.gl_scenes(): colour / depth / mask_color / mask_depth of
                   AggregationRenderer.render (the hi-res buffers) and the outputs of aggregate_conditions on that renderer
  warp_gl_fbw.npz  SimpleRenderer.render and forward_backward_warp on a real SimpleRenderer
Inputs are not stored: the tests rebuild them from the same seeds (tests/warp_common.py).
OpenGL implementations agree on WHICH pixels a triangle covers (the specification fixes the rule) up to their sub-pixel
snapping (llvmpipe: 8 bits); the tests therefore compare masks by IoU and depth / colour by tolerance.
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
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.glshim import glm, moderngl  # noqa: E402


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


def _erode(img, kernel, iterations=1):
    """cv2.erode with a ones kernel: min filter whose border never erodes (cv2's default border value is +inf)."""
    out = ndimage.minimum_filter(np.asarray(img, np.float64), size=kernel.shape, mode="constant", cval=np.inf)
    return out.astype(np.asarray(img).dtype)


sys.modules["moderngl"] = moderngl
sys.modules["glm"] = glm
_mod("easydict", EasyDict=EasyDict)
_mod("cv2", erode=_erode)
_mod("plyfile")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


ref_u = _load("ref_rgbd_utils", "/root/reference/rgbd_3d/utils.py")
ref_r = _load("ref_rgbd_renderer", "/root/reference/rgbd_3d/moderngl_renderer.py")
ref_u.SimpleRenderer = ref_r.SimpleRenderer        # utils.py takes the renderer as an argument; nothing else to wire

import warp_common as WC  # noqa: E402


def ref_mesh(rgbd1, mv):
    """[4,S,S] in [-1,1] -> (reference mesh with .modelview, colour texture) exactly as inference/sample.py:83,128-139 does."""
    hw = rgbd1.transpose(1, 2, 0) * 0.5 + 0.5
    depth_lin = ref_u.linearize_depth(hw[:, :, 3:], 0.6, 5.0)
    mesh = ref_u.depth_to_mesh(depth_lin, padding="frustum", fov=45, modelview=glm.mat4(mv), atol=0.03, rtol=0.03, erode_rgb=3,
                               cal_normal=True)
    mesh.modelview = glm.mat4(mv)
    return mesh, np.ascontiguousarray(hw[:, :, :3])


out = {}
info = moderngl.create_context().info
print("OpenGL:", info)
for tag, S, ssaa, near, far, views, target in WC.gl_scenes():
    R = S * ssaa
    meshes, cols = zip(*[ref_mesh(WC.synthetic_rgbd(S, seed, layers=layers)[0], mv) for mv, seed, layers in views])
    rend = ref_r.AggregationRenderer(R, S, near=near, far=far, device=0, max_views=max(27, len(views)))
    hi = rend.render(list(meshes), list(cols), glm.mat4(target), 45)
    for k in ("color", "depth", "mask_color", "mask_depth"):
        out[f"{tag}/{k}"] = np.asarray(hi[k]).astype(np.float32 if k in ("color", "depth") else np.bool_)
    # the reference's whole condition step on the real renderer (sample.py:87-98: near 0.6 / far 5 for the read-back)
    cond = ref_u.aggregate_conditions(rend, list(meshes), list(cols), glm.mat4(target), fov=45, near=0.6, far=5, atol=0.03, rtol=0.03,
                                      erode_rgb=3)
    if ssaa == 3:       # aggregate_conditions resizes render_size -> render_size // 3 (utils.py:450-454)
        for k in ("color", "depth", "mask", "mask_rgb", "depth_convex"):
            out[f"{tag}/cond_{k}"] = np.asarray(cond[k]).astype(np.float32 if k in ("color", "depth", "depth_convex") else np.bool_)
    print(f"{tag}: R={R} coverage depth {hi.mask_depth.mean():.3f} colour {hi.mask_color.mean():.3f}")
    del rend
np.savez_compressed(os.path.join(HERE, "warp_gl.npz"), gl_version=np.array(info["GL_VERSION"]), gl_renderer=np.array(info["GL_RENDERER"]), **out)
print("wrote warp_gl.npz", round(os.path.getsize(os.path.join(HERE, "warp_gl.npz")) / 1e6, 2), "MB")

# ---------------------------------------------------------------------------------------------------------------------
# SimpleRenderer + forward_backward_warp (training-time augmentation, utils.py:335-417) on the real renderer: the same two
# cases as tests/golden/warp_fbw.npz (which runs the same reference function on the oracle rasteriser)
fbw = {}
for tag, S, seed, yaw, pitch in [("S32", 32, 31, 0.2, -0.1), ("S64", 64, 32, -0.25, 0.12)]:
    hw = WC.synthetic_rgbd(S, seed, smooth_color=True)[0].transpose(1, 2, 0) * 0.5 + 0.5       # [S,S,4] in [0,1]
    mv0, mv1 = WC.orbit(0.0, 0.0), WC.orbit(yaw, pitch)
    rend = ref_r.SimpleRenderer(3 * S, S, near=0.1, far=200.0, device=0)
    o = ref_u.forward_backward_warp(rend, hw.astype(np.float64), glm.mat4(mv1), glm.mat4(mv0), padding=S, fov=45, near=0.6, far=5.0,
                                    atol=0.02, rtol=0.02)
    for k in ("color", "depth", "mask"):
        fbw[f"{tag}_{k}"] = np.asarray(o[k], np.float32)
    # one plain SimpleRenderer.render as well (the forward half: view 0's mesh seen from view 1)
    mesh0 = ref_u.depth_to_mesh(ref_u.linearize_depth(hw[:, :, 3:], 0.6, 5.0), padding=S, fov=45, modelview=glm.mat4(mv0), atol=None, rtol=None)
    res = rend.render(mesh0, hw[:, :, :3], glm.mat4(mv1), 45)
    for k in ("color", "depth", "mask"):
        fbw[f"{tag}_fwd_{k}"] = np.asarray(res[k]).astype(np.bool_ if k == "mask" else np.float32)
    print(f"forward_backward_warp {tag}: mask {o.mask.mean():.3f}; forward render coverage {res.mask.mean():.3f}")
    del rend
np.savez_compressed(os.path.join(HERE, "warp_gl_fbw.npz"), **fbw)
print("wrote warp_gl_fbw.npz", round(os.path.getsize(os.path.join(HERE, "warp_gl_fbw.npz")) / 1e6, 2), "MB")

# ---------------------------------------------------------------------------------------------------------------------
# Two more call patterns of the reference (stored separately: tests/golden/warp_gl_more.npz)
more = {}
# (1) free-view fusion rendering (inference/render.py:40-88): load_scene's meshes -- NUMERIC padding 32, metric depth
#     (inference/utils.py:108-111) -- on an SSAA-5 renderer with near 0.1 / far 200
S = 32
views = [(WC.orbit(0.0, 0.0), 90, False), (WC.orbit(0.3, 0.1), 91, True)]
target = WC.orbit(0.6 * np.cos(2.0), 0.15 * np.sin(2.0))
meshes, cols = [], []
for v, (mv, seed, layers) in enumerate(views):
    hw = WC.synthetic_rgbd(S, seed, layers=layers)[0].transpose(1, 2, 0) * 0.5 + 0.5
    depth_lin = ref_u.linearize_depth(hw[:, :, 3:], 0.6, 5.0).astype(np.float32)
    mesh = ref_u.depth_to_mesh(depth_lin, 32, 45, glm.mat4(mv), atol=0.03, rtol=0.03, erode_rgb=3, cal_normal=True)
    mesh.modelview = glm.mat4(mv)
    meshes.append(mesh)
    cols.append(np.ascontiguousarray(hw[:, :, :3]))
    more[f"pad32/vbo_{v}"] = np.concatenate([mesh.vertices.position, mesh.vertices.normal, mesh.vertices.uv, mesh.vertices.flag], -1).astype(np.float32)
    more[f"pad32/faces_{v}"] = mesh.faces.astype(np.int32)
rend = ref_r.AggregationRenderer(5 * S, S, near=0.1, far=200.0, device=0)
hi = rend.render(meshes, cols, glm.mat4(target), 45)
for k in ("color", "depth", "mask_color", "mask_depth"):
    more[f"pad32/{k}"] = np.asarray(hi[k]).astype(np.float32 if k in ("color", "depth") else np.bool_)
more["pad32/target"] = target
print(f"free view on load_scene meshes: coverage {hi.mask_depth.mean():.3f}")
del rend
# (2) the autoregressive chain of inference/sample.py:87-139: ONE renderer, aggregate_conditions(is_autoregressive=True) before
#     every new view -- only the newest mesh is uploaded per call, the earlier ones persist in the renderer
vs = WC.viewset_3x9()
rend = ref_r.AggregationRenderer(3 * S, S, near=0.01, far=200.0, device=0, max_views=27)
meshes, cols = [], []
for k in range(5):
    mv = WC.orbit(*vs[k])
    if k > 0:
        cond = ref_u.aggregate_conditions(rend, meshes, cols, glm.mat4(mv), fov=45, near=0.6, far=5, atol=0.03, rtol=0.03, erode_rgb=3)
        for key in ("color", "depth", "mask", "mask_rgb", "depth_convex"):
            more[f"chain/{k}/{key}"] = np.asarray(cond[key]).astype(np.float32 if key in ("color", "depth", "depth_convex") else np.bool_)
        print(f"chain view {k}: mask {cond.mask.mean():.3f} mask_rgb {cond.mask_rgb.mean():.3f}")
    mesh, col = ref_mesh(WC.synthetic_rgbd(S, 200 + k, layers=(k == 2))[0], mv)
    meshes.append(mesh)
    cols.append(col)
np.savez_compressed(os.path.join(HERE, "warp_gl_more.npz"), **more)
print("wrote warp_gl_more.npz", round(os.path.getsize(os.path.join(HERE, "warp_gl_more.npz")) / 1e6, 2), "MB")
