"""Scene wire format (inference/utils.py:74-113), colorize_depth (:25-41) and the render trajectory (render.py:41-61):
host-side pieces of the free-view path, CPU only."""
import io

import numpy as np
import torch

from ivid_amd.inference import utils as U


def _views(V=3, S=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(V, 4, S, S, generator=g) * 2.4 - 1.2        # exceeds [-1,1] on purpose: exercises the clips


def test_save_scene_follows_the_reference_wire_format(tmp_path):
    from PIL import Image
    views = _views()
    mvs = [np.eye(4, dtype=np.float32) * (i + 1) for i in range(3)]
    path = str(tmp_path / "scene_a.npz")
    U.save_scene(path, views, mvs, fov=45, near=0.6, far=5)
    data = np.load(path, allow_pickle=True)["data"]
    assert len(data) == 3 and set(data[0].keys()) == {"color", "depth", "fov", "modelview"}
    hw = views.numpy().transpose(0, 2, 3, 1) * 0.5 + 0.5                                          # sample.py:126
    for i, d in enumerate(data):
        col = np.asarray(Image.open(io.BytesIO(d["color"])))
        assert col.dtype == np.uint8 and col.shape == (16, 16, 3)
        assert np.array_equal(col, np.clip(hw[i, :, :, :3] * 255, 0, 255).astype(np.uint8))      # utils.py:77
        dep = np.asarray(Image.open(io.BytesIO(d["depth"])))
        assert dep.dtype == np.uint8 and dep.shape == (16, 16, 4)                                  # float32 as RGBA8
        z = np.frombuffer(np.ascontiguousarray(dep), dtype=np.float32).reshape(16, 16, 1)
        dd = np.clip(hw[i, :, :, 3:], 1e-6, 1.0 - 1e-6)
        assert np.array_equal(z, (0.6 * 5 / (5 - (5 - 0.6) * dd)).astype(np.float32))              # linearize_depth :38-58
        assert d["fov"] == 45 and np.array_equal(d["modelview"], mvs[i])


def test_load_scene_round_trip(tmp_path):
    views = _views(V=2, S=8, seed=3)
    path = str(tmp_path / "s.npz")
    U.save_scene(path, views, [np.eye(4), np.eye(4)], 45, 0.6, 5)
    sc = U.read_scene(path)
    assert len(sc) == 2
    assert sc[0]["color"].shape == (8, 8, 3) and sc[0]["color"].max() <= 1.0 and sc[0]["color"].min() >= 0.0
    assert sc[0]["depth"].shape == (8, 8, 1) and sc[0]["depth"].dtype == np.float32
    assert (sc[0]["depth"] >= 0.6 - 1e-5).all() and (sc[0]["depth"] <= 5 + 1e-4).all()
    assert sc[1]["modelview"].shape == (4, 4)


class FakeMat4:                          # PyGLM's mat4.to_list() is column-major (a list of columns)
    def __init__(self, m):
        self.m = np.asarray(m, dtype=np.float32)

    def to_list(self):
        return self.m.T.tolist()


def test_load_scene_accepts_glm_like_modelviews(tmp_path):
    m = np.arange(16, dtype=np.float32).reshape(4, 4)
    v = _views(V=1, S=8)
    path = str(tmp_path / "g.npz")
    U.save_scene(path, v, [m], 45, 0.6, 5)
    data = np.load(path, allow_pickle=True)["data"]
    data[0]["modelview"] = FakeMat4(m)
    np.savez_compressed(path, data=data)
    assert np.array_equal(U.read_scene(path)[0]["modelview"], m)


def test_save_scene_writes_glm_mat4_when_pyglm_is_importable(tmp_path, monkeypatch):
    """inference/utils.py:90-101 pickles meshes[i].modelview, a glm.mat4, and the reference's own render.py hands it to
    glm.inverse (moderngl_renderer.py:309).  With a `glm` module present the file must carry mat4 objects (column-major),
    and they must come back as the same math-order matrix."""
    import sys
    import types

    class mat4:                                   # the slice of PyGLM's mat4 that matters: 16 column-major scalars in, to_list out
        def __init__(self, *a):
            assert len(a) == 16
            self.cols = np.asarray(a, dtype=np.float32).reshape(4, 4)     # cols[c][r]

        def to_list(self):
            return self.cols.tolist()

    fake = types.ModuleType("glm")
    fake.mat4 = mat4
    fake.inverse = lambda m: mat4(*np.linalg.inv(m.cols.T).T.reshape(-1))
    monkeypatch.setitem(sys.modules, "glm", fake)
    test_save_scene_writes_glm_mat4_when_pyglm_is_importable.mat4 = mat4   # picklable by reference
    globals()["mat4"] = mat4
    mat4.__module__, mat4.__qualname__ = __name__, "mat4"
    m = np.array([[1, 2, 3, 4], [0, 1, 0, 5], [0, 0, 1, 6], [0, 0, 0, 1]], dtype=np.float32)
    path = str(tmp_path / "glm.npz")
    U.save_scene(path, _views(V=1, S=8), [m], 45, 0.6, 5)
    stored = np.load(path, allow_pickle=True)["data"][0]["modelview"]
    assert isinstance(stored, mat4)
    assert np.array_equal(stored.cols[3, :3], m[:3, 3])                    # column 3 of a column-major mat4 = the translation
    inv = fake.inverse(stored)                                             # what the reference does with it
    assert np.allclose(np.asarray(inv.to_list()).T, np.linalg.inv(m), atol=1e-6)
    assert np.array_equal(U.read_scene(path)[0]["modelview"], m)


def test_save_scene_reference_call_form(tmp_path):
    """save_scene(path, meshes, colors) as inference/sample.py:156,166 calls it: meshes carry .depth (metric), .fov, .modelview."""
    S = 8
    depth = np.linspace(0.7, 4.0, S * S, dtype=np.float64).reshape(S, S, 1)
    col = np.random.default_rng(0).uniform(0, 1, (S, S, 3))
    mesh = dict(depth=depth, fov=45, modelview=np.eye(4, dtype=np.float32))
    path = str(tmp_path / "ref.npz")
    U.save_scene(path, [mesh], [col])
    sc = U.read_scene(path)
    assert np.array_equal(sc[0]["depth"], depth.astype(np.float32)) and sc[0]["fov"] == 45
    assert np.array_equal(sc[0]["color"], np.clip(col * 255, 0, 255).astype(np.uint8) / 255)


def test_reorder_matches_the_references_literal_index_list():
    order = [23, 17, 11, 5, 2, 8, 14, 20, 26, 21, 15, 9, 3, 0, 6, 12, 18, 24, 22, 16, 10, 4, 1, 7, 13, 19, 25]   # utils.py:49-51
    x = [torch.full((1, 2, 2), float(i)) for i in range(27)]
    out = U.reorder(x, "3x9")
    assert out.shape == (27, 1, 2, 2) and [int(v[0, 0, 0]) for v in out] == order
    y = U.reorder(x[1:], "3x9")                      # 26 conditioning images: a blank (-1) view 0 is prepended (:46-47)
    assert [int(v[0, 0, 0]) for v in y] == [o if o else -1 for o in order]
    assert U.parse_int_list("1,2,5-10") == [1, 2, 5, 6, 7, 8, 9, 10] and U.parse_int_list("0-8") == list(range(9))


def test_write_video_pipes_raw_frames_to_ffmpeg(tmp_path, monkeypatch):
    """render.py:87-88 writes <scene>.mp4 at 30 fps.  Without imageio the frames go to an `ffmpeg` executable as raw rgb24;
    a stand-in executable records its arguments and stdin."""
    import os
    import stat
    fake = tmp_path / "bin" / "ffmpeg"
    fake.parent.mkdir()
    fake.write_text("#!/bin/sh\nfor a; do last=$a; done\necho \"$@\" > \"$last.args\"\ncat > \"$last\"\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake.parent) + os.pathsep + os.environ["PATH"])
    frames = np.random.default_rng(1).integers(0, 256, (5, 16, 24, 3), dtype=np.uint8)
    out = U.write_video(str(tmp_path / "videos" / "a.mp4"), frames, fps=30)
    assert out.endswith("a.mp4") and open(out, "rb").read() == frames.tobytes()
    args = open(out + ".args").read()
    assert "-s 24x16" in args and "-r 30" in args and "rgb24" in args and "libx264" in args
    # no ffmpeg, no imageio: an animated GIF with the same frames
    monkeypatch.setenv("PATH", str(tmp_path / "nowhere"))
    out2 = U.write_video(str(tmp_path / "videos" / "b.mp4"), frames, fps=30)
    from PIL import Image
    assert out2.endswith("b.gif") and Image.open(out2).n_frames == 5


def test_colorize_depth_matches_the_reference_mapping():
    d = torch.linspace(-1, 1, 256).reshape(1, 1, 16, 16)
    c = U.colorize_depth(d)                                    # tensor in -> [3,H,W] in [-1,1]
    assert isinstance(c, torch.Tensor) and c.shape == (3, 16, 16)
    lut = U._inferno_lut()
    assert lut.shape == (256, 3) and lut.dtype == np.uint8
    assert tuple(lut[0]) == (0, 0, 4) and tuple(lut[255]) == (252, 255, 164)    # inferno end points #000004 / #fcffa4
    assert tuple(lut[128]) == (188, 55, 84)                                     # inferno(0.5) = #bc3754
    import matplotlib                                                           # the committed table = OpenCV's conversion of
    ref = np.round(np.asarray(matplotlib.colormaps["inferno"](np.arange(256))[:, :3]) * 255).astype(np.uint8)   # matplotlib's data
    assert np.array_equal(lut, ref)
    # depth -1 (near) -> index 255 (bright), depth +1 (far) -> index 0 (dark): utils.py:33-34
    assert np.allclose((c[:, 0, 0].numpy() + 1) / 2 * 255, lut[255], atol=1e-4)
    assert np.allclose((c[:, -1, -1].numpy() + 1) / 2 * 255, lut[0], atol=1e-4)
    a = U.colorize_depth(np.full((4, 4), 0.25), min=0, max=1)  # array in -> [H,W,3] in [0,1]
    assert a.shape == (4, 4, 3) and np.allclose(a[0, 0] * 255, lut[int((1 - 0.25) * 255)])


def test_render_trajectories_follow_render_py():
    from ivid_amd.inference import render as R
    from ivid_amd.rgbd_3d import camera
    tr = R.trajectory("swing", 60, 1)
    ts = np.linspace(0, 2 * np.pi, 60)
    assert len(tr) == 60
    for k in (0, 17, 59):
        assert np.allclose(tr[k], camera.orbit(0.6 * np.cos(ts[k]), 0.15 * np.sin(ts[k])))
    rnd = R.trajectory("random", 60, 5, rng=np.random.default_rng(0))
    assert len(rnd) == 5 and all(len(r) == 1 and np.asarray(r[0]).shape == (4, 4) for r in rnd)


def _tv_make_grid(tensor, nrow, padding=2, value_range=(-1, 1), pad_value=0.0):
    """torchvision.utils.make_grid(normalize=True), restated from its documented algorithm with torch ops in ITS order
    (clamp_, sub_, div_, narrow().copy_) -- independent of the product's slicing code."""
    import math
    t = tensor.clone().float()
    if t.size(1) == 1:
        t = torch.cat((t, t, t), 1)
    lo, hi = value_range
    t.clamp_(min=lo, max=hi)
    t.sub_(lo).div_(max(hi - lo, 1e-5))
    if t.size(0) == 1:
        return t.squeeze(0)
    nmaps = t.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(t.size(2) + padding), int(t.size(3) + padding)
    grid = t.new_full((t.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, y * height + padding, height - padding).narrow(2, x * width + padding, width - padding).copy_(t[k])
            k = k + 1
    return grid


def test_save_grid_is_torchvision_save_image(tmp_path):
    """inference/sample.py:158-165: utils.save_image(x, path, nrow, normalize=True, value_range=(-1, 1)) -- 2-px borders of 0,
    clamp to [-1, 1] -> [0, 1], `mul(255).add(0.5).clamp(0, 255)` to uint8.  The reference's 3x9 grid of 128^2 views is 1172 x 392."""
    from PIL import Image
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 3, 4, 6, generator=g) * 0.8          # 5 images on a 2 x 3 grid: one empty cell, values beyond [-1, 1]
    p = str(tmp_path / "g" / "grid.png")
    U.save_grid(p, x, 3)
    img = np.asarray(Image.open(p))
    assert img.shape == (2 * 6 + 2, 3 * 8 + 2, 3)
    want = _tv_make_grid(x, 3).mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    assert np.array_equal(img, want)
    # hand-computed cells: border and the empty sixth cell are 0; image 4 sits at row 1, column 1
    assert img[:2].max() == 0 and img[:, :2].max() == 0 and img[8:, 18:].max() == 0
    v = float(x[4, 1, 2, 3])
    assert img[8 + 2, 10 + 3, 1] == int(min(max((min(max(v, -1.0), 1.0) + 1) / 2 * 255 + 0.5, 0), 255))
    # rounding, not truncation: 0.0 -> 127.5 + 0.5 = 128
    U.save_grid(p, torch.zeros(2, 3, 2, 2), 2)
    assert int(np.asarray(Image.open(p))[2, 2, 0]) == 128
    # one image: no border (make_grid returns it as is); single-channel input is repeated to RGB
    U.save_grid(p, torch.full((1, 1, 3, 3), 1.0), 9)
    one = np.asarray(Image.open(p))
    assert one.shape == (3, 3, 3) and one.min() == 255
    # the reference's sizes
    assert tuple(U.make_grid(torch.zeros(27, 3, 128, 128), nrow=9, normalize=True, value_range=(-1, 1)).shape) == (3, 392, 1172)
    assert tuple(U.make_grid(torch.zeros(2, 3, 128, 128), nrow=2, normalize=True, value_range=(-1, 1)).shape) == (3, 132, 262)
