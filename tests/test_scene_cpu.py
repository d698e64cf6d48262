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
    sc = U.load_scene(path)
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
    assert np.array_equal(U.load_scene(path)[0]["modelview"], m)


def test_colorize_depth_matches_the_reference_mapping():
    d = torch.linspace(-1, 1, 256).reshape(1, 1, 16, 16)
    c = U.colorize_depth(d)                                    # tensor in -> [3,H,W] in [-1,1]
    assert isinstance(c, torch.Tensor) and c.shape == (3, 16, 16)
    lut = U._inferno_lut()
    assert lut.shape == (256, 3) and lut.dtype == np.uint8
    assert tuple(lut[0]) == (0, 0, 4) and tuple(lut[255]) == (252, 255, 164)    # inferno end points
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
