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
