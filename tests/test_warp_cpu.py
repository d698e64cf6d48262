"""CPU: the warp oracle against the reference-generated mesh fixtures, Pillow, and its own geometry."""
import numpy as np
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
    assert res["skipped"] == 0
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
