"""Camera matrices for the warp path.  The reference builds them with PyGLM (`glm.lookAt`,
`glm.perspective`, `glm.inverse`; inference/sample.py:305-336, moderngl_renderer.py:296,310), which is
not a dependency here: these are GLM's documented right-handed / [-1,1]-depth formulas evaluated in
float32 (glm.mat4 is float32).  Matrices are plain 4x4 numpy arrays in math (row, column) order."""
import math

import numpy as np


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


def as_matrix(mv):
    """A modelview in any of the forms callers hold -> 4x4 float32 numpy in math (row, column) order.  numpy arrays /
    nested lists are taken as they are (math order: what this package's own `look_at` returns).  A PyGLM `mat4` (the
    reference's type, inference/sample.py:305-336) is COLUMN-major (`m[col][row]`): its `to_list()` is transposed."""
    if mv is None:
        return np.eye(4, dtype=np.float32)
    if hasattr(mv, "to_list") and not isinstance(mv, np.ndarray):
        return np.asarray(mv.to_list(), dtype=np.float32).T.copy()
    m = np.asarray(mv, dtype=np.float32)
    if m.shape != (4, 4):
        raise ValueError(f"modelview must be 4x4, got {m.shape}")
    return m


def inverse(m):
    return np.linalg.inv(np.asarray(m, dtype=np.float32)).astype(np.float32)


def orbit(yaw, pitch):
    """Camera on the unit sphere looking at the origin, +Y up (every ivid viewset uses this form)."""
    return look_at((np.sin(yaw) * np.cos(pitch), np.sin(pitch), np.cos(yaw) * np.cos(pitch)), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))


def viewset(name, num_samples=1, rng=None):
    """Camera lists of inference/sample.py:304-338.  'uncond': 1 view; 'random': per-sample [front, N(0,.3) yaw /
    N(0,.15) pitch]; '3x9': 27 views, yaw-major over [0,±.15,±.3,±.45,±.6], pitch-minor over [0,±.15]."""
    if name == "uncond":
        return [orbit(0.0, 0.0)]
    if name == "random":
        rng = rng if rng is not None else np.random
        out = []
        for _ in range(num_samples):
            yaw, pitch = 0.3 * rng.normal(), 0.15 * rng.normal()
            out.append([orbit(0.0, 0.0), orbit(yaw, pitch)])
        return out
    if name == "3x9":
        yaws, pitches = [0.0], [0.0]
        for i in range(4):
            yaws += [(i + 1) * 0.15, -(i + 1) * 0.15]
        pitches += [0.15, -0.15]
        return [orbit(y, p) for y in yaws for p in pitches]
    raise NotImplementedError(name)
