"""CPU: the warp oracle against the reference-generated mesh fixtures, Pillow, and its own geometry."""
import os

import numpy as np
import pytest
from PIL import Image

import common as C
import warp_common as WC
from oracle import warp_oracle as W


def test_mesh_oracle_matches_reference_fixture():
    g = C.load_golden("warp_mesh")
    for S, seed in ((16, 0), (32, 1)):
        rgbd = WC.synthetic_rgbd(S, seed)
        assert np.array_equal(rgbd, g[f"rgbd_{S}"])
        mesh, _ = WC.oracle_mesh(rgbd[0], g[f"modelview_{S}"])
        assert np.array_equal(mesh["faces"], g[f"faces_{S}"])
        assert np.array_equal(mesh["verts"][:, 8], g[f"vbo_{S}"][:, 8])           # flags
        assert np.abs(mesh["verts"] - g[f"vbo_{S}"]).max() < 1e-6
        hw = rgbd[0].transpose(1, 2, 0) * 0.5 + 0.5
        dproj = W.project_depth(W.linearize_depth(hw[:, :, 3:], 0.6, 5.0).astype(np.float32), 0.6, 5.0)
        assert np.array_equal(W.depth_edge(dproj, 0.03, 0.03), g[f"edge_{S}"])


def test_product_lanczos_tables_are_bit_exact_with_pillow():
    from ivid_amd.rgbd_3d.resample import lanczos_tables, resample8_reference
    rng = np.random.default_rng(3)
    for R, S in ((384, 128), (96, 32), (640, 128)):
        img = rng.integers(0, 256, (R, R, 3), dtype=np.uint8)
        img[: R // 4] = 0
        ref = np.array(Image.fromarray(img).resize((S, S), Image.Resampling.LANCZOS))
        assert np.array_equal(resample8_reference(img, S), ref)
        b0, k0, ks0 = lanczos_tables(R, S)
        b1, k1, ks1 = W.lanczos_tables(R, S)
        assert ks0 == ks1 and np.array_equal(b0, b1) and np.array_equal(k0, k1)


def test_product_cameras_match_glm_formulas():
    from ivid_amd.rgbd_3d import camera
    mv = camera.orbit(0.3, -0.15)
    assert np.allclose(mv, WC.orbit(0.3, -0.15))
    eye = np.array([np.sin(0.3) * np.cos(-0.15), np.sin(-0.15), np.cos(0.3) * np.cos(-0.15), 1.0])
    assert np.allclose(mv @ eye, [0, 0, 0, 1], atol=1e-6)                 # camera sits at the eye
    assert np.allclose(mv @ np.array([0, 0, 0, 1.0]), [0, 0, -1, 1], atol=1e-6)  # looks down -Z at distance 1
    p = camera.perspective(np.deg2rad(45), 1.0, 0.01, 200.0)
    near = p @ np.array([0, 0, -0.01, 1.0]); far = p @ np.array([0, 0, -200.0, 1.0])
    assert abs(near[2] / near[3] + 1) < 1e-5 and abs(far[2] / far[3] - 1) < 1e-4
    vs = camera.viewset("3x9")
    assert len(vs) == 27 and np.allclose(vs[0], camera.orbit(0, 0)) and np.allclose(vs[3], camera.orbit(0.15, 0.0))
    assert np.allclose(vs[1], camera.orbit(0.0, 0.15))                    # yaw-major, pitch-minor (sample.py:330-336)
    rnd = camera.viewset("random", 4, np.random.default_rng(0))
    assert len(rnd) == 4 and len(rnd[0]) == 2


def test_oracle_reprojection_identity():
    """Rendering a view's own mesh from its own camera must give the view back: full coverage, the source depth at
    the pixel centres, the source colours (size-independent geometric property of the whole warp chain)."""
    S, R = 32, 96
    rgbd = WC.synthetic_rgbd(S, 5, smooth_color=True)
    mv = WC.orbit(0.0, 0.0)
    mesh, col = WC.oracle_mesh(rgbd[0], mv)
    res = W.render([mesh], [col], mv, 45, S, R)
    assert res["clipped"] == 0
    assert res["mask_depth"].mean() > 0.85          # everything but discontinuity sheets
    hw = rgbd[0].transpose(1, 2, 0) * 0.5 + 0.5
    depth_src = W.linearize_depth(hw[:, :, 3:], 0.6, 5.0)
    centre = res["depth"][1::3, 1::3]
    ok = res["mask_depth"][1::3, 1::3]
    # vertices sit exactly at pixel centres -> centre sub-pixel depth equals the source depth (24-bit z-buffer precision)
    assert np.abs(centre[ok] - depth_src[ok]).max() < 2e-3
    out = W.resolve(res, S, 3, 0.6, 5.0, 0.03, 0.03, 3)
    m = out["mask_rgb"][..., 0] > 0
    assert m.mean() > 0.3   # 7x7 erosion around the step discontinuity and the border eats a lot at S=32
    assert np.abs(out["color"][m] - col[m]).max() < 0.03   # NEAREST x3 up + LANCZOS down of band-limited colours ~ identity


def test_oracle_novel_view_has_holes_and_hull():
    S, R = 32, 96
    rgbd = WC.synthetic_rgbd(S, 6)
    mesh, col = WC.oracle_mesh(rgbd[0], WC.orbit(0.0, 0.0))
    out = W.aggregate_conditions([mesh], [col], WC.orbit(0.45, 0.0), S, 3, 45, 0.6, 5.0, 0.03, 0.03, 3)
    assert 0.2 < out["mask"].mean() < 0.98                               # disocclusions appear
    assert (out["mask_rgb"] <= out["mask"]).all()                        # colour mask is a subset of the depth mask
    assert out["depth_convex"][out["mask"] > 0].min() > 0


def test_oracle_resolve_equals_the_references_own_aggregate_conditions():
    """tests/golden/warp_resolve.npz: outputs of /root/reference's aggregate_conditions run on a stub renderer that
    returns the stored hi-res buffers (make_golden_warp.py) -- the restatement must reproduce them exactly."""
    g = C.load_golden("warp_resolve")
    for tag in ("S32x3", "S16x5", "S32x3_wide"):
        S, ssaa, erode = (int(v) for v in g[f"{tag}_cfg"])
        hi = {k: g[f"{tag}_in_{k}"] for k in ("color", "depth", "mask_color", "mask_depth")}
        out = W.resolve(hi, S, ssaa, 0.6, 5.0, 0.03, 0.03, erode)
        for k in ("color", "depth", "mask", "mask_rgb", "depth_convex"):
            assert np.array_equal(np.asarray(out[k], np.float64), np.asarray(g[f"{tag}_out_{k}"], np.float64)), (tag, k)
        assert 0.02 < g[f"{tag}_out_mask"].mean() < 0.98


def test_oracle_rasteriser_clips_triangles_behind_the_eye():
    """A target camera INSIDE the source frustum's skirt region: skirt / sheet triangles have vertices with w <= 0.  The
    clipped rasterisation must stay finite, and must agree with an unclipped evaluation wherever no clipping happened
    (same scene from a far camera: clipped == 0)."""
    S, R = 32, 96
    rgbd = WC.synthetic_rgbd(S, 9, layers=True)
    mesh, col = WC.oracle_mesh(rgbd[0], WC.orbit(0.0, 0.0))
    near_cam = W.look_at((0.05, 0.02, 0.35), (0.0, 0.0, -1.0), (0, 1, 0))      # well inside the unit sphere
    res = W.render([mesh], [col], near_cam, 45, S, R)
    assert res["clipped"] > 0
    assert np.isfinite(res["depth"]).all() and np.isfinite(res["color"]).all()
    assert 0.05 < res["mask_depth"].mean() <= 1.0
    far = W.render([mesh], [col], WC.orbit(0.6, 0.15), 45, S, R)
    assert far["clipped"] == 0 and far["lowconf"].mean() > 0.01                 # skirt / sheets visible as low-confidence hull


def test_oracle_fill_rule_gives_every_pixel_of_a_shared_edge_to_exactly_one_triangle():
    """Identity view at SSAA 3: mesh vertices project EXACTLY onto target pixel centres (u = (i + 0.5)/S = (3i + 1.5)/(3S)),
    so quad edges and diagonals run through pixel centres -- with the top-left rule the covered area is still exactly the
    image (no double hits are observable, but no gaps either)."""
    S, R = 16, 48
    z = np.full((S, S, 1), 1.0)
    rgbd = np.concatenate([np.full((S, S, 3), 0.5), W.project_depth(z, 0.6, 5.0)], -1).astype(np.float32)
    rgbd = (rgbd * 2 - 1).transpose(2, 0, 1)
    mv = W.look_at((0, 0, 1), (0, 0, 0), (0, 1, 0))
    mesh, col = WC.oracle_mesh(rgbd, mv)
    res = W.render([mesh], [col], mv, 45, S, R)
    md = res["mask_depth"][..., 0]
    # the height field spans the pixel centres 1 .. R-2 (source pixel i sits on target pixel 3i+1); the half-pixel rim is
    # frustum skirt (low confidence).  Inside, every pixel is confident: none is lost on a shared edge ...
    assert md[1:R - 2, 1:R - 2].all()
    assert np.abs(res["depth"][1:R - 2, 1:R - 2] - 1.0).max() < 1e-4
    # ... and nowhere in the image is a pixel left without a fragment (surface or skirt)
    assert (md | res["lowconf"]).all()


def test_oracle_simple_render_and_unpadded_mesh_embedding():
    """SimpleRenderer semantics on the oracle side (used as the rasteriser of the reference's forward_backward_warp in
    make_golden_warp.py): an UNPADDED reference mesh is embedded with a ring of copies (zero-area triangles); rendered
    from its own camera it returns the source depth at the pixel centres it covers."""
    g = C.load_golden("warp_mesh")
    S, R = 16, 48
    vb, faces = g["vbo_nopad_16"], g["faces_nopad_16"]
    ref_mesh = dict(vertices=dict(position=vb[:, 0:3], uv=vb[:, 3:5], flag=vb[:, 5:6]), faces=faces, modelview=g["modelview_16"])
    m = W.from_reference_mesh(ref_mesh, S)
    assert m["verts"].shape == ((S + 2) ** 2, 9) and m["diag"].shape == ((S + 1) ** 2,)
    col = g["rgbd_16"][0, :3].transpose(1, 2, 0) * 0.5 + 0.5
    out = W.simple_render(m, col, g["modelview_16"], 45, S, R, 0.1, 200.0)
    centre, ok = out["depth"][1::3, 1::3, 0], out["mask"][1::3, 1::3, 0]
    assert ok.mean() > 0.3                                    # all but discontinuity triangles (alpha 0; plenty at S = 16)
    assert np.abs(centre[ok] - g["depth_lin_16"][..., 0][ok]).max() < 2e-3
    assert out["depth"].max() > 150.0                         # the unpadded mesh leaves a rim of background (clear depth -> far)


GL_TAGS = [s[0] for s in WC.gl_scenes()]


@pytest.mark.parametrize("tag", GL_TAGS)
def test_oracle_render_equals_the_references_renderer_on_real_opengl(tag):
    """tests/golden/warp_gl.npz holds what the REFERENCE's AggregationRenderer.render / aggregate_conditions return when their
    own code and GLSL shaders run on real OpenGL (Mesa llvmpipe; tests/golden/make_golden_gl.py).  The C rasteriser +
    aggregation + resolve of the oracle must reproduce it: this pins the warp oracle -- against which the HIP kernels are
    checked at every size -- to an actual OpenGL implementation.  Scenes: two views, three views with two depth layers, the
    full `3x9` viewset (26 source views), a camera inside the scene (512 triangles clipped at the near plane), white-noise
    depth (everything low-confidence), an SSAA-5 free-view frame, a full-size 128^2 pair at 384^2, and (round 3) the FULL 26-view
    `3x9` aggregation and a three-view layered scene at full size (128^2 views, 384^2 target)."""
    g = C.load_golden("warp_gl")
    tag_, S, ssaa, near, far, views, target = next(s for s in WC.gl_scenes() if s[0] == tag)
    meshes, cols = zip(*[WC.oracle_mesh(WC.synthetic_rgbd(S, seed, layers=layers)[0], mv) for mv, seed, layers in views])
    o = W.render(list(meshes), list(cols), target, 45, S, S * ssaa, near=near, far=far)
    rr = W.resolve({k: o[k] for k in ("color", "depth", "mask_color", "mask_depth")}, S, ssaa, 0.6, 5.0, 0.03, 0.03, 3) if ssaa == 3 else None
    WC.gl_assert(WC.gl_compare(g, tag, near, o, rr))


def test_oracle_simple_renderer_and_forward_backward_warp_equal_real_opengl():
    """tests/golden/warp_gl_fbw.npz = the reference's forward_backward_warp + SimpleRenderer.render executed with their own
    code and shaders on real OpenGL (make_golden_gl.py).  warp_fbw.npz -- the same reference function run on the ORACLE
    rasteriser (simple_render), the fixture the product is checked against -- must equal the OpenGL run: masks identical,
    colours identical, depth to 1e-5.  (The plain forward render of the real SimpleRenderer is compared with the product on
    the GPU: test_warp_gpu.py.)"""
    g, b = C.load_golden("warp_gl_fbw"), C.load_golden("warp_fbw")
    for tag, S, seed in (("S32", 32, 31), ("S64", 64, 32)):
        m, rm = g[f"{tag}_mask"][..., 0] > 0, b[f"{tag}_mask"][..., 0] > 0
        assert (m ^ rm).sum() <= 2 and 0.3 < m.mean() < 0.95
        both = m & rm
        assert np.abs(g[f"{tag}_depth"][..., 0][both] - b[f"{tag}_depth"][..., 0][both]).max() < 1e-5
        assert (np.abs(g[f"{tag}_color"][both] - b[f"{tag}_color"][both]).max(-1) > 1.5 / 255).mean() < 2e-3


def _edict_mesh(vbo, faces, mv):
    """A reference-layout mesh (what depth_to_mesh returns) from a stored vertex buffer [n, 9] + faces."""
    return dict(vertices=dict(position=vbo[:, :3], normal=vbo[:, 3:6], uv=vbo[:, 6:8], flag=vbo[:, 8:9]), faces=faces, modelview=mv)


def test_oracle_equals_real_opengl_on_load_scene_meshes_and_along_the_autoregressive_chain():
    """tests/golden/warp_gl_more.npz (make_golden_gl.py, the reference's code on real OpenGL): (1) free-view fusion rendering
    (inference/render.py) of load_scene's meshes -- numeric padding 32 -- at SSAA 5; (2) the conditions of views 1..4 of a
    `3x9` chain produced by ONE renderer with aggregate_conditions(is_autoregressive=True) (inference/sample.py:87-139)."""
    g = C.load_golden("warp_gl_more")
    S = 32
    views = [(WC.orbit(0.0, 0.0), 90, False), (WC.orbit(0.3, 0.1), 91, True)]
    meshes = [W.from_reference_mesh(_edict_mesh(g[f"pad32/vbo_{v}"], g[f"pad32/faces_{v}"], views[v][0]), S) for v in range(2)]
    cols = [np.ascontiguousarray(WC.synthetic_rgbd(S, seed, layers=layers)[0].transpose(1, 2, 0)[:, :, :3] * 0.5 + 0.5) for _, seed, layers in views]
    o = W.render(meshes, cols, g["pad32/target"], 45, S, 5 * S, near=0.1, far=200.0)
    WC.gl_assert(WC.gl_compare({k.replace("pad32/", "x/"): v for k, v in g.items()}, "x", 0.1, o))
    vs = WC.viewset_3x9()
    ms, cs = [], []
    for k in range(5):
        mv = WC.orbit(*vs[k])
        if k > 0:
            hi = W.render(ms, cs, mv, 45, S, 3 * S)
            rr = W.resolve({q: hi[q] for q in ("color", "depth", "mask_color", "mask_depth")}, S, 3, 0.6, 5.0, 0.03, 0.03, 3)
            assert (g[f"chain/{k}/mask"] != rr["mask"]).sum() <= 2 and (g[f"chain/{k}/mask_rgb"] != rr["mask_rgb"]).sum() <= 2, k
            assert (np.abs(g[f"chain/{k}/depth"] - rr["depth"]) > 1e-3).sum() <= 2, k
            assert (np.abs(g[f"chain/{k}/depth_convex"] - rr["depth_convex"]) > 1e-3).sum() <= 2, k
            assert (np.abs(g[f"chain/{k}/color"] - rr["color"]) > 1.5 / 255).mean() < 5e-3, k
        m, c = WC.oracle_mesh(WC.synthetic_rgbd(S, 200 + k, layers=(k == 2))[0], mv)
        ms.append(m)
        cs.append(c)
