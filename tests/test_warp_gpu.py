"""GPU parity of the HIP depth-warp path: mesh build vs the reference-generated fixture (pinned), rasterise +
aggregate vs the C software rasteriser (GL rules restated — unpinned, compared by coverage IoU / tolerances as
SURVEY.md §7 hard part 7 prescribes), SSAA resolve vs Pillow / the numpy restatement (bit-exact)."""
import numpy as np
import pytest
import torch

import common as C
import gpu_util as G
import warp_common as WC
from oracle import warp_oracle as W

pytestmark = pytest.mark.gpu


def renderer(B, S, ssaa=3, max_views=4):
    from ivid_amd.rgbd_3d import WarpRenderer
    return WarpRenderer(B, S, ssaa, max_views)


def test_mesh_build_matches_reference_fixture():
    g = C.load_golden("warp_mesh")
    for S in (16, 32):
        r = renderer(1, S)
        r.add_view(torch.from_numpy(g[f"rgbd_{S}"]).cuda(), g[f"modelview_{S}"], 45, 0.6, 5.0, 0.03, 0.03, 3)
        m = r.mesh_numpy(0, 0)
        vb = g[f"vbo_{S}"]
        assert np.array_equal(m.faces, g[f"faces_{S}"]), "triangulation (diagonal choice) differs"
        assert np.array_equal(m.vertices.flag[:, 0], vb[:, 8]), "discontinuity / padding / erosion flags differ"
        e_pos = np.abs(m.vertices.position - vb[:, 0:3]).max()
        e_nrm = np.abs(m.vertices.normal - vb[:, 3:6]).max()
        e_uv = np.abs(m.vertices.uv - vb[:, 6:8]).max()
        G.report(f"warp/mesh_S{S}", pos=e_pos, normal=e_nrm, uv=e_uv)
        assert e_pos < 2e-6 and e_nrm < 5e-5 and e_uv < 1e-7, (e_pos, e_nrm, e_uv)
        col = r.colors[0, 0].cpu().numpy()
        assert np.array_equal(col, g[f"rgbd_{S}"][0, :3].transpose(1, 2, 0) * 0.5 + 0.5)


def test_mesh_build_scene_mode_matches_reference_fixture():
    """load_scene's mesh: metric depth in, numeric padding 32 (inference/utils.py:108-111) vs the live reference's
    depth_to_mesh(depth, 32, ...) output (tests/golden/make_golden_warp.py)."""
    g = C.load_golden("warp_mesh")
    for S in (16, 32):
        r = renderer(1, S)
        rgbd = np.concatenate([g[f"rgbd_{S}"][0, :3] * 0.5 + 0.5, g[f"depth_lin_{S}"].astype(np.float32).transpose(2, 0, 1)], 0)
        r.add_view(torch.from_numpy(rgbd[None].astype(np.float32)).cuda(), g[f"modelview_{S}"], 45, atol=0.03, rtol=0.03,
                   erode_rgb=3, padding=32, metric=True)
        m = r.mesh_numpy(0, 0)
        vb = g[f"vbo_pad32_{S}"]
        assert np.array_equal(m.faces, g[f"faces_pad32_{S}"])
        assert np.array_equal(m.vertices.flag[:, 0], vb[:, 8])
        e_pos = np.abs(m.vertices.position - vb[:, 0:3]).max()
        e_nrm = np.abs(m.vertices.normal - vb[:, 3:6]).max()
        G.report(f"warp/mesh_pad32_S{S}", pos=e_pos, normal=e_nrm)
        assert e_pos < 4e-5 and e_nrm < 5e-5, (e_pos, e_nrm)   # the numeric skirt reaches |x| ~ 30: 1 fp32 ulp = 2e-6


def test_scene_file_round_trip_and_free_view_render(tmp_path):
    """save_scene -> load_scene -> meshes rebuilt on the GPU -> SSAA-5 free-view frames (inference/render.py:62-84).  The
    frame rendered from a source camera must reproduce that source view."""
    from ivid_amd.inference import render as R, utils as U
    from ivid_amd.rgbd_3d import WarpRenderer
    S = 64
    mvs = [WC.orbit(0.0, 0.0), WC.orbit(0.2, 0.05)]
    views = torch.from_numpy(np.concatenate([WC.synthetic_rgbd(S, 3, smooth_color=True), WC.synthetic_rgbd(S, 3, smooth_color=True)]))
    path = str(tmp_path / "scenes" / "scene_test.npz")
    U.save_scene(path, views, mvs, 45, 0.6, 5.0)
    scene = U.load_scene(path)
    assert len(scene) == 2 and scene[0]["color"].shape == (S, S, 3) and scene[0]["depth"].dtype == np.float32
    rr = WarpRenderer(1, S, 5, 4, near=0.1, far=200.0)
    colors, depths = R.render_scene(rr, scene[:1], [mvs[0], WC.orbit(0.1, 0.0)], ssaa=5)
    assert colors.shape == (2, S, S, 3) and depths.shape == (2, S, S, 3) and colors.dtype == np.uint8
    src = (scene[0]["color"] * 255).astype(np.float64)
    err = np.abs(colors[0].astype(np.float64) - src)
    G.report("warp/free_view_identity", mean_abs_8bit=err.mean(), q95=np.quantile(err, 0.95))
    assert err.mean() < 3.0 and np.quantile(err, 0.95) < 12.0
    tr = R.trajectory("swing", 60, 1)
    assert len(tr) == 60 and np.allclose(tr[0], WC.orbit(0.6, 0.0), atol=1e-6)


def test_mesh_build_full_size_batch_matches_oracle():
    S, B = 128, 3
    rgbd = np.concatenate([WC.synthetic_rgbd(S, s) for s in range(B)])
    mvs = np.stack([WC.orbit(0.15 * b, -0.1 * b) for b in range(B)])
    r = renderer(B, S)
    r.add_view(torch.from_numpy(rgbd).cuda(), mvs, 45, 0.6, 5.0, 0.03, 0.03, 3)
    for b in range(B):
        om, _ = WC.oracle_mesh(rgbd[b], mvs[b])
        m = r.mesh_numpy(0, b)
        d = np.abs(np.concatenate([m.vertices.position, m.vertices.normal, m.vertices.uv], -1) - om["verts"][:, :8])
        G.report(f"warp/mesh_S128_b{b}", faces_equal=float(np.array_equal(m.faces, om["faces"])),
                 flags_equal=float(np.array_equal(m.vertices.flag[:, 0], om["verts"][:, 8])), pos=d[:, :3].max(),
                 normal=d[:, 3:6].max(), uv=d[:, 6:].max(), flag_mismatch=float((m.vertices.flag[:, 0] != om["verts"][:, 8]).sum()))
        assert np.array_equal(m.faces, om["faces"])
        assert np.array_equal(m.vertices.flag[:, 0], om["verts"][:, 8])
        assert d[:, :3].max() < 2e-6 and d.max() < 5e-5


def _compare_render(S, ssaa, views, target, tag):
    B = 2
    R = S * ssaa
    rgbds = [np.concatenate([WC.synthetic_rgbd(S, 10 * v + b) for b in range(B)]) for v in range(len(views))]
    r = renderer(B, S, ssaa)
    for v, mv in enumerate(views):
        r.add_view(torch.from_numpy(rgbds[v]).cuda(), mv, 45, 0.6, 5.0, 0.03, 0.03, 3)
    hi = r.render(target, 45)
    torch.cuda.synchronize()
    cond = r.conditions(target, 45, 0.6, 5.0, 0.03, 0.03, 3)
    for b in range(B):
        meshes, cols = zip(*[WC.oracle_mesh(rgbds[v][b], views[v]) for v in range(len(views))])
        ref = W.render(list(meshes), list(cols), target, 45, S, R)
        assert ref["skipped"] == 0
        md, mc = hi.mask_depth[b].cpu().numpy().astype(bool), hi.mask_color[b].cpu().numpy().astype(bool)
        rd, rc = ref["mask_depth"][..., 0], ref["mask_color"][..., 0]
        iou_d = (md & rd).sum() / max((md | rd).sum(), 1)
        iou_c = (mc & rc).sum() / max((mc | rc).sum(), 1)
        both = md & rd
        dg, dr_ = hi.depth[b].cpu().numpy(), ref["depth"][..., 0]
        drel = np.abs(dg[both] - dr_[both]) / dr_[both]
        c8 = hi.color8[b].cpu().numpy().astype(int)
        r8 = (np.clip(ref["color"], 0, 1) * 255).astype(np.uint8).astype(int)
        cboth = mc & rc
        cdiff = np.abs(c8[cboth] - r8[cboth]).max(axis=-1)
        G.report(f"warp/render_{tag}_b{b}", iou_depth=iou_d, iou_color=iou_c, depth_rel_p999=float(np.quantile(drel, 0.999)),
                 depth_rel_median=float(np.median(drel)), color_exact_frac=float((cdiff == 0).mean()),
                 color_within2_frac=float((cdiff <= 2).mean()), coverage=float(md.mean()))
        assert iou_d > 0.995 and iou_c > 0.99, (iou_d, iou_c)
        assert np.median(drel) < 1e-5 and np.quantile(drel, 0.999) < 1e-2
        assert (cdiff <= 2).mean() > 0.995
        # resolve: device kernels vs the numpy/Pillow restatement applied to the DEVICE's own hi-res buffers (pinned part)
        dev_hi = dict(color=hi.color8[b].cpu().numpy().astype(np.float32) / 255.0 + 1e-4, depth=dg[..., None],
                      mask_color=mc[..., None], mask_depth=md[..., None])
        rr = W.resolve(dev_hi, S, ssaa, 0.6, 5.0, 0.03, 0.03, 3)
        hw = lambda t: t[b].permute(1, 2, 0).cpu().numpy()
        assert np.array_equal(hw(cond.mask), rr["mask"]) and np.array_equal(hw(cond.mask_rgb), rr["mask_rgb"])
        assert np.abs(hw(cond.depth) - rr["depth"]).max() < 1e-6 and np.abs(hw(cond.depth_convex) - rr["depth_convex"]).max() < 1e-6
        assert np.abs(hw(cond.color) - rr["color"]).max() < 1e-7         # 8-bit LANCZOS is bit-exact


def test_render_two_views_small():
    _compare_render(32, 3, [WC.orbit(0.0, 0.0), WC.orbit(0.15, 0.0)], WC.orbit(0.3, 0.15), "S32")


def test_render_three_views_full_size():
    _compare_render(128, 3, [WC.orbit(0.0, 0.0), WC.orbit(0.0, 0.15), WC.orbit(-0.15, 0.0)], WC.orbit(0.15, -0.15), "S128")


def test_reprojection_identity_full_size():
    S, B = 128, 2
    rgbd = np.concatenate([WC.synthetic_rgbd(S, 40 + b, smooth_color=True) for b in range(B)])
    mv = WC.orbit(0.0, 0.0)
    r = renderer(B, S)
    r.add_view(torch.from_numpy(rgbd).cuda(), mv)
    c = r.conditions(mv)
    for b in range(B):
        hw = rgbd[b].transpose(1, 2, 0) * 0.5 + 0.5
        m = c.mask[b, 0].cpu().numpy() > 0
        assert m.mean() > 0.85
        assert np.abs(c.depth[b, 0].cpu().numpy()[m] - hw[:, :, 3][m]).max() < 2e-3   # z-buffer depth comes back
        mr = c.mask_rgb[b, 0].cpu().numpy() > 0
        assert mr.mean() > 0.6
        err = np.abs(c.color[b].permute(1, 2, 0).cpu().numpy()[mr] - hw[:, :, :3][mr])
        # LANCZOS ringing from zero-valued holes leaks one pixel past the 5x5 erosion at a few edge pixels (the oracle
        # shows the same: max 0.07, 99.9th percentile 0.004 at S=128)
        G.report(f"warp/identity_b{b}", mask=m.mean(), mask_rgb=mr.mean(), err_max=err.max(), err_q999=np.quantile(err, 0.999),
                 err_q99=np.quantile(err, 0.99))
        assert np.quantile(err, 0.99) < 0.01 and err.max() < 0.2


def test_compat_functions_keep_reference_signatures():
    from ivid_amd import rgbd_3d
    S = 16
    g = C.load_golden("warp_mesh")
    mesh = rgbd_3d.utils.depth_to_mesh(g["depth_lin_16"], padding="frustum", fov=45, modelview=g["modelview_16"], atol=0.03,
                                       rtol=0.03, erode_rgb=3, cal_normal=True)
    assert np.array_equal(mesh.faces, g["faces_16"]) and np.abs(mesh.vertices.position - g["vbo_16"][:, :3]).max() < 1e-5
    rr = rgbd_3d.AggregationRenderer(S * 3, S)
    assert rr.render_size == 48
    rr.add_view(torch.from_numpy(g["rgbd_16"]).cuda(), g["modelview_16"])
    out = rgbd_3d.utils.aggregate_conditions(rr, None, None, WC.orbit(0.15, 0.0), fov=45, near=0.6, far=5, atol=0.03, rtol=0.03, erode_rgb=3)
    assert out.color.shape == (S, S, 3) and out.mask.shape == (S, S, 1) and set(out.keys()) == {"color", "depth", "mask", "mask_rgb", "depth_convex"}
