"""Synthetic RGBD scenes for the warp tests (same recipe as tests/golden/make_golden_warp.py)."""
import numpy as np

from oracle import warp_oracle as W


def synthetic_rgbd(S, seed, smooth_color=False, layers=False):
    """layers=True: a second scene family -- a near foreground slab (z ~ 0.75) floating in front of a far background
    (z ~ 1.9) with a hole in it: from another camera the discontinuity sheets and frustum skirts of one view cross the
    surfaces of the others (the low-confidence "farther z wins" rule of aggregation.csh:27-34 decides those pixels).
    layers="noise": white-noise depth in [0.7, 4] -- what a randomly initialised network generates: every quad is a
    discontinuity and every triangle a long sliver in any other view (the rasteriser's worst case)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:S, 0:S] / (S - 1.0)
    z = 1.0 + 0.25 * np.exp(-((xx - 0.45) ** 2 + (yy - 0.55) ** 2) / 0.05) + 0.05 * np.sin(7 * xx) * np.cos(5 * yy)
    z[int(0.6 * S):, int(0.55 * S):] += 0.8
    if isinstance(layers, str) and layers == "noise":
        z = rng.uniform(0.7, 4.0, (S, S))
    elif layers:
        z = 1.9 + 0.1 * np.cos(4 * xx + seed) * np.sin(3 * yy)
        fg = (np.abs(xx - 0.5) < 0.28) & (np.abs(yy - 0.45) < 0.3) & ~((np.abs(xx - 0.55) < 0.07) & (np.abs(yy - 0.4) < 0.1))
        z[fg] = 0.75 + 0.05 * xx[fg]
    d01 = W.project_depth(z, 0.6, 5.0)
    col = rng.uniform(0, 1, (S, S, 3))
    if smooth_color:  # band-limited colours: NEAREST x3 up + LANCZOS down is then close to the identity
        col = np.stack([0.5 + 0.4 * np.sin(3 * xx + seed), 0.5 + 0.4 * np.cos(2 * yy), 0.3 + 0.4 * xx * yy], -1)
    rgbd = np.concatenate([col, d01[..., None]], -1).astype(np.float32)
    return (rgbd * 2 - 1).transpose(2, 0, 1)[None].copy()


def orbit(yaw, pitch):
    return W.look_at((np.sin(yaw) * np.cos(pitch), np.sin(pitch), np.cos(yaw) * np.cos(pitch)), (0, 0, 0), (0, 1, 0))


def oracle_mesh(rgbd1, mv):
    """rgbd1: [4,S,S] in [-1,1] -> (oracle mesh, colour texture [S,S,3]) following sample.py:83,128-139."""
    hw = rgbd1.transpose(1, 2, 0) * 0.5 + 0.5
    depth_lin = W.linearize_depth(hw[:, :, 3:], 0.6, 5.0)
    return W.depth_to_mesh(depth_lin, 45, mv, 0.03, 0.03, 3), np.ascontiguousarray(hw[:, :, :3])


def viewset_3x9():
    """(yaw, pitch) of the 27 cameras of the `3x9` viewset, yaw-major / pitch-minor (inference/sample.py:325-336)."""
    yaws, pitches = [0.0], [0.0]
    for i in range(4):
        yaws += [(i + 1) * 0.15, -(i + 1) * 0.15]
    pitches += [0.15, -0.15]
    return [(y, p) for y in yaws for p in pitches]


def gl_scenes():
    """The scenes whose renders by the REFERENCE's own AggregationRenderer + GLSL shaders on real OpenGL (Mesa llvmpipe) are
    committed as tests/golden/warp_gl.npz (tests/golden/make_golden_gl.py).  Each: tag, S, ssaa, near, far,
    source views [(modelview 4x4, seed, layers)], target modelview."""
    vs = viewset_3x9()
    inside = W.look_at((0.05, 0.02, 0.35), (0.0, 0.0, -1.0), (0, 1, 0))       # a camera well inside the unit sphere
    sc = [
        ("two_views_S32", 32, 3, 0.01, 200.0, [(orbit(0.0, 0.0), 0, False), (orbit(0.15, 0.0), 10, False)], orbit(0.3, 0.15)),
        ("wide_layers_S32", 32, 3, 0.01, 200.0,
         [(orbit(0.0, 0.0), 20, False), (orbit(-0.3, 0.15), 21, True), (orbit(0.6, 0.0), 22, False)], orbit(-0.6, -0.15)),
        ("viewset_3x9_S32", 32, 3, 0.01, 200.0, [(orbit(*vs[v]), 100 + v, v % 3 == 1) for v in range(26)], orbit(*vs[26])),
        ("inside_S32", 32, 3, 0.01, 200.0, [(orbit(0.0, 0.0), 30, True), (orbit(0.3, 0.0), 40, False)], inside),
        ("noise_S32", 32, 3, 0.01, 200.0, [(orbit(0.0, 0.0), 50, "noise"), (orbit(0.45, 0.1), 60, "noise")], orbit(-0.5, -0.15)),
        ("free_view_ssaa5_S32", 32, 5, 0.1, 200.0, [(orbit(0.0, 0.0), 70, False), (orbit(-0.3, 0.0), 71, True)],
         orbit(0.6 * np.cos(1.0), 0.15 * np.sin(1.0))),
        ("full_size_S128", 128, 3, 0.01, 200.0, [(orbit(0.0, 0.0), 80, False), (orbit(0.0, 0.15), 81, False)], orbit(0.15, -0.15)),
        # round 3: the FULL `3x9` aggregation at the size the sampler runs it (26 stored views of 128^2, 384^2 render target:
        # the conditioning of the 27th view, sample.py:87-98) and a full-size layered scene seen from the far side
        ("viewset_3x9_S128", 128, 3, 0.01, 200.0, [(orbit(*vs[v]), 200 + v, v % 3 == 1) for v in range(26)], orbit(*vs[26])),
        ("wide_layers_S128", 128, 3, 0.01, 200.0,
         [(orbit(0.0, 0.0), 90, False), (orbit(-0.3, 0.15), 91, True), (orbit(0.6, 0.0), 92, False)], orbit(-0.6, -0.15)),
    ]
    return sc


def gl_compare(g, tag, near, got, resolved=None):
    """got: dict(color, depth, mask_color, mask_depth[, lowconf]) in the reference's read-back form vs the committed result of
    the reference's own renderer on real OpenGL.  Returns the error figures; asserts nothing."""
    md, mc = g[f"{tag}/mask_depth"][..., 0], g[f"{tag}/mask_color"][..., 0]
    od, oc = np.asarray(got["mask_depth"])[..., 0].astype(bool), np.asarray(got["mask_color"])[..., 0].astype(bool)
    dg, do = g[f"{tag}/depth"][..., 0], np.asarray(got["depth"])[..., 0]
    both = md & od
    rel = np.abs(dg[both] - do[both]) / do[both] if both.any() else np.zeros(1)
    # the visual hull: low-confidence pixels (skirts, discontinuity sheets) carry a depth but no mask (aggregation.csh:27-34)
    hg, ho = (~md) & (dg > near * 1.01), (~od) & (do > near * 1.01)
    hb = hg & ho
    hrel = np.abs(dg[hb] - do[hb]) / do[hb] if hb.any() else np.zeros(1)
    cb = mc & oc
    cd = np.abs(g[f"{tag}/color"][cb] - np.asarray(got["color"])[cb]).max(-1) if cb.any() else np.zeros(1)
    out = dict(mask_depth_mismatch=int((md ^ od).sum()), mask_color_mismatch=int((mc ^ oc).sum()), pixels=int(md.size),
               hull_mismatch=int((hg ^ ho).sum()), hull_pixels=int(hg.sum()), depth_rel_p999=float(np.quantile(rel, 0.999)),
               hull_depth_rel_p99=float(np.quantile(hrel, 0.99)), color_off_frac=float((cd > 1e-3).mean()))
    if resolved is not None:
        for k in ("mask", "mask_rgb"):
            out[f"cond_{k}_mismatch"] = int((g[f"{tag}/cond_{k}"] != np.asarray(resolved[k]).astype(bool)).sum())
        for k in ("depth", "depth_convex"):
            out[f"cond_{k}_off"] = int((np.abs(g[f"{tag}/cond_{k}"] - np.asarray(resolved[k])) > 1e-3).sum())
        out["cond_color_off_frac"] = float((np.abs(g[f"{tag}/cond_color"] - np.asarray(resolved["color"])) > 1.5 / 255).mean())
    return out


def gl_assert(e):
    """Bars for 'equals real OpenGL': coverage identical up to a few pixels (OpenGL implementations may differ in sub-pixel
    snapping), depth to the 24-bit z-buffer's resolution, colour up to the few pixels where a z tie picks another fragment."""
    assert e["mask_depth_mismatch"] <= max(2, e["pixels"] // 20000) and e["mask_color_mismatch"] <= max(2, e["pixels"] // 20000), e
    assert e["hull_mismatch"] <= max(2, e["hull_pixels"] // 500), e
    assert e["depth_rel_p999"] < 2e-4 and e["hull_depth_rel_p99"] < 2e-4, e
    assert e["color_off_frac"] < 2e-3, e
    if "cond_mask_mismatch" in e:
        assert e["cond_mask_mismatch"] <= 2 and e["cond_mask_rgb_mismatch"] <= 2, e
        assert e["cond_depth_off"] <= 2 and e["cond_depth_convex_off"] <= 2, e
        assert e["cond_color_off_frac"] < 5e-3, e
