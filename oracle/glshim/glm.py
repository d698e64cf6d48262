"""Stand-in for PyGLM (`glm`) with the handful of functions the reference's warp code uses (test infrastructure).

Conventions: a `mat4` holds the MATHEMATICAL matrix (row r, column c); indexing `m[c]` returns COLUMN c like glm;
`to_gl_bytes()` is glm's column-major memory (what moderngl's Uniform.write receives); numpy conversion yields the
mathematical matrix, so that the reference's `np.matmul(glm.inverse(modelview), points.T)` (rgbd_3d/utils.py:234) computes
world = V^-1 . cam as intended.  Arithmetic in float32 like glm's default types."""
import numpy as np


def radians(deg):
    return np.float32(deg) * np.float32(np.pi / 180.0)


class vec3:
    def __init__(self, *a):
        if len(a) == 1:
            a = np.asarray(a[0], np.float32).reshape(-1)[:3]
        self.v = np.asarray(a, np.float32).reshape(3)

    def __array__(self, dtype=None, copy=None):
        return self.v if dtype is None else self.v.astype(dtype)

    def __getitem__(self, i):
        return self.v[i]

    def to_gl_bytes(self):
        return self.v.tobytes()


class mat4:
    def __init__(self, *a):
        """mat4() identity; mat4(m) from a MATHEMATICAL 4x4 matrix (this module's own functions); mat4(16 scalars) in glm's
        constructor order, i.e. column by column (PyGLM's `glm.mat4(x0, y0, z0, w0, x1, ...)`)."""
        if len(a) == 16:
            self.m = np.asarray(a, np.float32).reshape(4, 4).T.copy()
        elif len(a) == 0 or a[0] is None:
            self.m = np.eye(4, dtype=np.float32)
        else:
            self.m = np.asarray(a[0], np.float32).reshape(4, 4).copy()

    def to_list(self):                  # PyGLM: list of COLUMNS
        return [[float(v) for v in self.m[:, c]] for c in range(4)]

    def __array__(self, dtype=None, copy=None):
        return self.m if dtype is None else self.m.astype(dtype)

    def __getitem__(self, c):           # glm: m[c] is column c
        return self.m[:, c].copy()

    def __mul__(self, o):
        if isinstance(o, mat4):
            return mat4(self.m @ o.m)
        return self.m @ np.asarray(o, np.float32)

    def to_gl_bytes(self):
        return np.ascontiguousarray(self.m.T).tobytes()


def _m(x):
    return x.m if isinstance(x, mat4) else np.asarray(x, np.float32).reshape(4, 4)


def inverse(m):
    return mat4(np.linalg.inv(_m(m).astype(np.float64)).astype(np.float32))


def mat3(m):
    return _m(m)[:3, :3]


def perspective(fovy, aspect, near, far):
    t = np.tan(np.float32(fovy) / np.float32(2))
    p = np.zeros((4, 4), np.float32)
    p[0, 0] = 1.0 / (np.float32(aspect) * t)
    p[1, 1] = 1.0 / t
    p[2, 2] = -(np.float32(far) + np.float32(near)) / (np.float32(far) - np.float32(near))
    p[2, 3] = -(np.float32(2) * np.float32(far) * np.float32(near)) / (np.float32(far) - np.float32(near))
    p[3, 2] = -1.0
    return mat4(p)


def lookAt(eye, center, up):
    eye, center, up = (np.asarray(x, np.float32) for x in (eye, center, up))
    f = center - eye
    f = f / np.linalg.norm(f)
    s = np.cross(f, up)
    s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4, dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s @ eye, -u @ eye, f @ eye
    return mat4(m)
