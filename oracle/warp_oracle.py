"""CPU restatement of ivid's RGBD depth-warp conditioning — TEST INFRASTRUCTURE (the oracle).

Follows /root/reference/rgbd_3d/utils.py (linearize_depth :38-58, project_depth :61-67, image_uv
:70-86, unproject :89-110, triangulate :113-134, mask_discontinuity :137-141, depth_to_mesh
:144-260, cal_depth_normal :263-274, depth_edge :311-332, aggregate_conditions :420-477) and
/root/reference/rgbd_3d/moderngl_renderer.py:260-340 + shaders/aggregation.* .

Pinning status
  * mesh construction, depth (de)linearisation, depth_edge: PINNED — tests/golden/make_golden.py
    imports the reference's own rgbd_3d/utils.py (with stand-ins for the missing glm / cv2 / plyfile
    modules) and stores its outputs; tests compare this restatement against them.
  * 8-bit LANCZOS resolve: PINNED to the real Pillow (the same library the reference calls).
  * aggregate_conditions' post-processing of the rendered buffers (8-bit LANCZOS, centre-sample depth, 7-of-9
    masks, depth_edge, erosion): PINNED — tests/golden/make_golden_warp.py runs the reference's own
    aggregate_conditions on a stub renderer that returns stored hi-res buffers (tests/golden/warp_resolve.npz).
  * rasterisation + shader arithmetic: PINNED TO REAL OPENGL — moderngl / EGL do not exist offline, but Mesa's software
    rasteriser does: oracle/glshim/ (an off-screen llvmpipe context through the swrast DRI driver + a stand-in `moderngl`
    module) lets tests/golden/make_golden_gl.py run the reference's own rgbd_3d/moderngl_renderer.py and GLSL shaders;
    tests/test_warp_cpu.py checks this restatement (oracle/warp_raster.c: near-plane clipping, top-left fill rule,
    perspective-correct varyings, in a formulation different from the HIP kernel) against those outputs
    (tests/golden/warp_gl.npz, warp_gl_fbw.npz): 7 scenes, 0 mask pixels off, depth to the 24-bit z resolution.
Missing third-party pieces restated here: PyGLM lookAt/perspective/inverse (GLM's documented
formulas, float32 like glm.mat4), cv2.erode (min filter whose border never erodes).
"""
import ctypes
import math
import os
import subprocess

import numpy as np
from PIL import Image
from scipy import ndimage

_HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- cameras (GLM restated, math row/column order)
def look_at(eye, center, up):
    eye, center, up = (np.asarray(v, dtype=np.float32) for v in (eye, center, up))
    f = center - eye
    f = f / np.linalg.norm(f)
    s = np.cross(f, up)
    s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4, dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye)
    return m


def perspective(fovy_rad, aspect, near, far):
    t = np.float32(math.tan(fovy_rad / 2.0))
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = -(far + near) / (far - near)
    m[2, 3] = -(2.0 * far * near) / (far - near)
    m[3, 2] = -1.0
    return m


def inverse(m):
    return np.linalg.inv(np.asarray(m, dtype=np.float32)).astype(np.float32)


def view_3x9():
    """Camera list of the '3x9' viewset, inference/sample.py:325-336 (yaw-major, pitch-minor)."""
    yaws, pitches = [0.0], [0.0]
    for i in range(4):
        yaws += [(i + 1) * 0.15, -(i + 1) * 0.15]
    pitches += [0.15, -0.15]
    return [look_at((np.sin(y) * np.cos(p), np.sin(p), np.cos(y) * np.cos(p)), (0, 0, 0), (0, 1, 0))
            for y in yaws for p in pitches]


# ---------------------------------------------------------------- depth <-> z-buffer
def linearize_depth(depth, near=0.5, far=100):
    depth = np.clip(depth, 1e-6, 1.0 - 1e-6)
    return near * far / (far - (far - near) * depth)


def project_depth(depth, near=0.5, far=100):
    depth = np.clip(depth, near, far)
    return (1 / near - 1 / depth) / (1 / near - 1 / far)


# ---------------------------------------------------------------- mesh
def _sobel_normals(points):
    p = np.pad(points, ((1, 1), (1, 1), (0, 0)), "edge")
    ex = p[:, 2:] - p[:, :-2]
    ey = p[:-2, :] - p[2:, :]
    ex = (ex[:-2] + 2 * ex[1:-1] + ex[2:]) / 4
    ey = (ey[:, :-2] + 2 * ey[:, 1:-1] + ey[:, 2:]) / 4
    n = np.cross(ex, ey)
    return n / np.linalg.norm(n, axis=-1, keepdims=True)


def erode_square(mask01, ksize):
    """cv2.erode(mask, ones((k,k))): min filter; cv2's default border value never erodes."""
    return ndimage.minimum_filter(mask01, size=(ksize, ksize), mode="constant", cval=np.inf)


def depth_to_mesh(depth, fov, modelview, atol, rtol, erode_rgb):
    """depth_to_mesh(depth, padding='frustum', cal_normal=True, ...) — depth: [S,S,1] linear depth (float32)."""
    S = depth.shape[0]
    plane = 2 * np.tan(0.5 * np.deg2rad(fov))
    focal = 0.5 / np.tan(0.5 * np.deg2rad(fov))
    lin = np.linspace(0.5 / S, 1 - 0.5 / S, S)
    uv = np.stack(np.meshgrid(lin, lin, indexing="xy"), axis=-1)
    pts = np.concatenate([(uv - 0.5) / focal, -np.ones((S, S, 1))], axis=-1)[::-1] * depth
    normal = _sobel_normals(pts)
    pad = lambda a: np.pad(a, ((1, 1), (1, 1), (0, 0)), "edge")
    pts, uv, dpad, normal = pad(pts), pad(uv), pad(depth), pad(normal)
    ppp = plane / S
    pts[0, :, 1] += ppp * dpad[0, :, 0]
    pts[-1, :, 1] -= ppp * dpad[-1, :, 0]
    pts[:, 0, 0] -= ppp * dpad[:, 0, 0]
    pts[:, -1, 0] += ppp * dpad[:, -1, 0]
    pts[0, :] *= -0.1 / pts[0, :, 2:]
    pts[-1, :] *= -0.1 / pts[-1, :, 2:]
    pts[:, 0] *= -0.1 / pts[:, 0, 2:]
    pts[:, -1] *= -0.1 / pts[:, -1, 2:]
    P = S + 2
    padflag = np.zeros((P, P), dtype=bool)
    padflag[0, :] = padflag[-1, :] = padflag[:, 0] = padflag[:, -1] = True
    # two triangles per quad along the shorter 3-D diagonal
    idx = np.arange(P * P).reshape(P, P)
    ft = np.linalg.norm(pts[:-1, :-1] - pts[1:, 1:], axis=-1) < np.linalg.norm(pts[:-1, 1:] - pts[1:, :-1], axis=-1)
    faces = np.stack([idx[:-1, 1:].ravel(), idx[:-1, :-1].ravel(), np.where(ft, idx[1:, 1:], idx[1:, :-1]).ravel(),
                      idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), np.where(ft, idx[:-1, :-1], idx[:-1, 1:]).ravel()],
                     axis=-1).reshape(-1, 3)
    d = dpad.reshape(-1)
    df = d[faces]
    disc_tri = np.logical_and(df.max(-1) - df.min(-1) > atol, (1 / df).max(-1) - (1 / df).min(-1) > rtol)
    disc = np.zeros(P * P, dtype=bool)
    disc[faces[disc_tri].ravel()] = True
    inv = inverse(modelview)
    pw = (inv @ np.concatenate([pts.reshape(-1, 3), np.ones((P * P, 1))], axis=-1).T).T[:, :3]
    nw = (inv[:3, :3] @ normal.reshape(-1, 3).T).T
    ero = np.zeros(P * P, dtype=bool)
    if erode_rgb is not None and erode_rgb > 0:
        m = erode_square((~disc).astype(np.float32).reshape(P, P), 2 * erode_rgb + 1)
        ero = (m == 0).ravel()
    flag = 1 * disc + 2 * padflag.ravel() + 4 * ero
    verts = np.concatenate([pw, nw, uv.reshape(-1, 2), flag[:, None]], axis=-1).astype(np.float32)  # the VBO
    return dict(verts=verts, faces=faces, diag=ft.astype(np.uint8).ravel(), modelview=np.asarray(modelview, np.float32))


# ---------------------------------------------------------------- rasteriser (C) + aggregation
_clib = None


def _lib():
    global _clib
    if _clib is None:
        so = os.path.join(_HERE, "_build", "libwarp_oracle.so")
        src = os.path.join(_HERE, "warp_raster.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _clib = ctypes.CDLL(so)
        _clib.oracle_raster.restype = ctypes.c_int
    return _clib


def render(meshes, colors, modelview, fov, S, R, near=0.01, far=200.0):
    """AggregationRenderer.render for one target view (moderngl_renderer.py:260-340): returns colour [R,R,3],
    metric depth [R,R,1], mask_color, mask_depth [R,R,1] (row 0 = image top)."""
    L = _lib()
    mvp = (perspective(np.deg2rad(fov), 1.0, near, far) @ np.asarray(modelview, np.float32)).astype(np.float32)
    acc = np.zeros((R * R, 8), dtype=np.float32)
    clipped = 0
    fp = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    for mesh, col in zip(meshes, colors):
        v = np.ascontiguousarray(mesh["verts"], np.float32)
        dg = np.ascontiguousarray(mesh["diag"], np.uint8)
        c = np.ascontiguousarray(col, np.float32)
        cam = np.ascontiguousarray(inverse(mesh["modelview"])[:3, 3], np.float32)
        d24 = np.empty(R * R, np.uint32)
        tri = np.empty(R * R, np.int32)
        bary = np.zeros((R * R, 3), np.float64)
        front = np.zeros(R * R, np.uint8)
        m = np.ascontiguousarray(mvp)
        clipped += L.oracle_raster(fp(v, ctypes.c_float), fp(dg, ctypes.c_ubyte), S, fp(m, ctypes.c_float), R,
                                   fp(d24, ctypes.c_uint32), fp(tri, ctypes.c_int32), fp(bary, ctypes.c_double),
                                   fp(front, ctypes.c_ubyte))
        L.oracle_shade_aggregate(fp(v, ctypes.c_float), fp(dg, ctypes.c_ubyte), fp(c, ctypes.c_float),
                                 fp(cam, ctypes.c_float), S, R, fp(d24, ctypes.c_uint32), fp(tri, ctypes.c_int32),
                                 fp(bary, ctypes.c_double), fp(front, ctypes.c_ubyte), fp(acc, ctypes.c_float))
    acc = acc.reshape(R, R, 8)
    color = np.where(acc[..., 3:4] > 0.0, acc[..., :3] / np.maximum(acc[..., 3:4], np.float32(1e-24)), np.float32(0.0))
    depth = np.where(acc[..., 5:6] > 0.0, acc[..., 4:5] / np.maximum(acc[..., 5:6], np.float32(1e-24)), np.float32(0.0))
    depth = (np.float32(near) * np.float32(far) / (np.float32(far) - depth * (np.float32(far) - np.float32(near)))).astype(np.float32)
    # clipped = triangles that crossed the near plane (frustum skirt / sheets with a vertex behind the eye): informative
    return dict(color=color, depth=depth, mask_color=acc[..., 7:8] > 0.5, mask_depth=acc[..., 6:7] > 0.5, clipped=clipped,
                lowconf=np.logical_and(acc[..., 5] > 0, acc[..., 5] < 1e-7))


def _raster(mesh, mvp, S, R, no_discard=False):
    """One mesh drawn alone: (depth24 uint32 [R*R], tri int32, bary float64 [R*R,3], front uint8, clipped count)."""
    L = _lib()
    fp = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    v = np.ascontiguousarray(mesh["verts"], np.float32)
    if no_discard:   # simple.fsh never discards: hide the padding bit from the rasteriser's back-face test
        v = v.copy()
        v[:, 8] = (v[:, 8].astype(np.int32) & ~2).astype(np.float32)
    dg = np.ascontiguousarray(mesh["diag"], np.uint8)
    d24, tri = np.empty(R * R, np.uint32), np.empty(R * R, np.int32)
    bary, front = np.zeros((R * R, 3), np.float64), np.zeros(R * R, np.uint8)
    m = np.ascontiguousarray(mvp, np.float32)
    clipped = L.oracle_raster(fp(v, ctypes.c_float), fp(dg, ctypes.c_ubyte), S, fp(m, ctypes.c_float), R,
                              fp(d24, ctypes.c_uint32), fp(tri, ctypes.c_int32), fp(bary, ctypes.c_double),
                              fp(front, ctypes.c_ubyte))
    return d24, tri, bary, front, clipped


def _tri_vertex_ids(tri, diag, P):
    """Vectorised triangulate order (utils.py:113-134): vertex ids [n,3] of triangle ids `tri`."""
    quad, Q = tri >> 1, P - 1
    qr, qc = quad // Q, quad % Q
    i00 = qr * P + qc
    i01, i10, i11 = i00 + 1, i00 + P, i00 + P + 1
    ft = diag[quad].astype(bool)
    even = (tri & 1) == 0
    return np.stack([np.where(even, i01, i10), np.where(even, i00, i11),
                     np.where(even, np.where(ft, i11, i10), np.where(ft, i00, i01))], axis=-1)


def from_reference_mesh(mesh, S):
    """A mesh dict of the REFERENCE's depth_to_mesh (vertices.{position,uv,flag[,normal]}, faces, modelview) -> this
    oracle's arrays (verts [(S+2)^2, 9], diag [(S+1)^2]).  An unpadded S x S mesh (padding=None) is embedded in the
    padded grid with a ring of COPIES of its border vertices: zero-area triangles, which no rasteriser draws."""
    vt = mesh["vertices"]
    pos = np.asarray(vt["position"], np.float32)
    nrm = np.asarray(vt["normal"], np.float32) if "normal" in vt else np.zeros_like(pos)
    vb = np.concatenate([pos, nrm, np.asarray(vt["uv"], np.float32), np.asarray(vt["flag"], np.float32).reshape(-1, 1)], -1)
    faces = np.asarray(mesh["faces"]).reshape(-1, 6)
    P = S + 2
    mv = mesh.get("modelview")
    mv = np.eye(4, dtype=np.float32) if mv is None else np.asarray(mv, np.float32)
    if vb.shape[0] == P * P:
        idx = np.arange(P * P).reshape(P, P)
        ft = faces[:, 2] == idx[1:, 1:].ravel()
        return dict(verts=vb, diag=ft.astype(np.uint8), modelview=mv)
    assert vb.shape[0] == S * S, vb.shape
    idx = np.arange(S * S).reshape(S, S)
    ft = (faces[:, 2] == idx[1:, 1:].ravel()).reshape(S - 1, S - 1)
    vb = np.pad(vb.reshape(S, S, 9), ((1, 1), (1, 1), (0, 0)), "edge").reshape(P * P, 9)
    diag = np.zeros((P - 1, P - 1), np.uint8)
    diag[1:-1, 1:-1] = ft
    return dict(verts=vb, diag=diag.ravel(), modelview=mv)


def simple_render(mesh, color, modelview, fov, S, R, near=0.01, far=200.0):
    """SimpleRenderer.render for one target view (moderngl_renderer.py:96-148; simple.vsh / simple.fsh): colour
    [R,R,3] fp32, depth [R,R,1] fp32 (linearised window depth, `far` where nothing was drawn), mask [R,R,1] bool."""
    mvp = (perspective(np.deg2rad(fov), 1.0, near, far) @ np.asarray(modelview, np.float32)).astype(np.float32)
    d24, tri, bary, front, _ = _raster(mesh, mvp, S, R, no_discard=True)
    P = S + 2
    hit = tri >= 0
    vi = _tri_vertex_ids(np.where(hit, tri, 0), np.asarray(mesh["diag"], np.uint8), P)
    V = np.asarray(mesh["verts"], np.float32)
    bw = bary.astype(np.float32)
    u = (bw * V[vi, 6]).sum(-1, dtype=np.float32)
    v = (bw * V[vi, 7]).sum(-1, dtype=np.float32)
    fe = (bw * (V[vi, 8].astype(np.int32) & 1).astype(np.float32)).sum(-1, dtype=np.float32)
    tx = np.clip(np.floor(u * np.float32(S)).astype(np.int64), 0, S - 1)
    ty = np.clip(np.floor(v * np.float32(S)).astype(np.int64), 0, S - 1)
    ff = hit & (front > 0)
    col = np.where(ff[:, None], np.asarray(color, np.float32)[ty, tx], np.float32(0.0))
    alpha = np.where(ff & ~(fe > 0.999), 1.0, 0.0)
    d = np.where(hit, d24.astype(np.float32) / np.float32(16777215.0), np.float32(1.0)).astype(np.float32)
    n, f = np.float32(near), np.float32(far)
    depth = (n * f / (f - d * (f - n))).astype(np.float32)
    return dict(color=col.reshape(R, R, 3).astype(np.float32), depth=depth.reshape(R, R, 1), mask=(alpha > 0.5).reshape(R, R, 1))


def depth_edge(depth, atol, rtol):
    def dd(x, y):
        x, y = np.maximum(x, 1e-6), np.maximum(y, 1e-6)
        return np.logical_and(np.abs(x - y) > atol, np.abs(1 / x - 1 / y) > rtol)
    cnt = np.zeros(depth.shape[:2] + (1,), dtype=np.uint8)
    m = dd(depth[:, 1:], depth[:, :-1]); cnt[:, 1:] += m; cnt[:, :-1] += m
    m = dd(depth[1:, :], depth[:-1, :]); cnt[1:, :] += m; cnt[:-1, :] += m
    m = dd(depth[1:, 1:], depth[:-1, :-1]); cnt[1:, 1:] += m; cnt[:-1, :-1] += m
    m = dd(depth[1:, :-1], depth[:-1, 1:]); cnt[1:, :-1] += m; cnt[:-1, 1:] += m
    return cnt < 3


def resolve(res, S, ssaa, near, far, atol, rtol, erode_rgb):
    """aggregate_conditions after the render call (utils.py:454-477)."""
    off = (ssaa - 1) // 2
    c8 = (np.clip(res["color"], 0, 1) * 255).astype(np.uint8)
    color = np.array(Image.fromarray(c8).resize((S, S), Image.Resampling.LANCZOS)) / 255.0
    depth = project_depth(res["depth"][off::ssaa, off::ssaa, :], near, far)
    mask = res["mask_depth"].reshape(S, ssaa, S, ssaa, 1).sum(axis=(1, 3)) > 0.75 * ssaa ** 2
    mask_rgb = res["mask_color"].reshape(S, ssaa, S, ssaa, 1).sum(axis=(1, 3)) > 0.75 * ssaa ** 2
    convex = depth.copy()
    mask = mask & depth_edge(depth, atol, rtol)
    k = 2 * erode_rgb - 1
    mask_rgb = mask_rgb & (erode_square(mask[..., 0].astype(np.float32), k)[..., None] > 0)
    return dict(color=color * mask_rgb, depth=depth * mask, mask=mask.astype(np.float32),
                mask_rgb=mask_rgb.astype(np.float32), depth_convex=convex)


def aggregate_conditions(meshes, colors, modelview, S, ssaa=3, fov=45, near=0.5, far=100, atol=0.02, rtol=0.02, erode_rgb=2):
    return resolve(render(meshes, colors, modelview, fov, S, S * ssaa), S, ssaa, near, far, atol, rtol, erode_rgb)


# ---------------------------------------------------------------- Pillow's 8-bit LANCZOS coefficient tables
def lanczos_tables(in_size, out_size):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the LANCZOS filter (support 3):
    bounds int32 [out][2] = (first source index, tap count); coeffs int32 [out][ksize], 22 fractional bits."""
    support0 = 3.0
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = support0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    sinc = lambda x: 1.0 if x == 0.0 else math.sin(x * math.pi) / (x * math.pi)
    lanczos = lambda x: sinc(x) * sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / fscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize
