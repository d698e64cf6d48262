"""inference/utils.py helpers of the reference that the sampling path needs (parse_int_list :13-22, colorize_depth
:25-41, reorder :44-55, save_scene / load_scene :74-113) plus PIL-based image writers (the reference uses imageio /
torchvision / cv2, which are not dependencies here)."""
import io
import os
import re

import numpy as np
import torch


def parse_int_list(s):
    """'1,2,5-10' -> [1, 2, 5, 6, 7, 8, 9, 10] (inference/utils.py:13-22)."""
    if isinstance(s, list):
        return s
    out = []
    rng = re.compile(r"^(\d+)-(\d+)$")
    for part in s.split(","):
        m = rng.match(part)
        out.extend(range(int(m.group(1)), int(m.group(2)) + 1) if m else [int(part)])
    return out


def reorder(x, viewset="3x9"):
    """Grid order of the 3x9 viewset (inference/utils.py:44-55).  Views are generated yaw-major / pitch-minor from the
    centre outwards (index = 3*yaw_slot + pitch_slot with yaw slots [0,+.15,-.15,+.3,-.3,...] and pitch slots [0,+.15,-.15]);
    the grid shows rows pitch -.15 / 0 / +.15 and columns yaw +.6 ... -.6.  A 26-entry list (the conditioning images,
    which have no entry for view 0) gets a blank (-1) image prepended first."""
    if viewset != "3x9":
        raise NotImplementedError(viewset)
    x = list(x)
    if len(x) == 26:
        x.insert(0, -torch.ones_like(x[0]))
    yaw_slots = [7, 5, 3, 1, 0, 2, 4, 6, 8]     # +.6, +.45, +.3, +.15, 0, -.15, -.3, -.45, -.6
    pitch_slots = [2, 0, 1]                     # -.15, 0, +.15
    return torch.stack([x[3 * y + p] for p in pitch_slots for y in yaw_slots], dim=0)


def to_uint8_image(chw):
    """[-1,1] CHW tensor -> HWC uint8 (np.clip(x*0.5+0.5,0,1)*255, sample.py:155)."""
    a = chw.detach().float().cpu().numpy().transpose(1, 2, 0) * 0.5 + 0.5
    return (np.clip(a, 0, 1) * 255).astype(np.uint8)


def save_png(path, chw):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(to_uint8_image(chw)).save(path)


def save_grid(path, nchw, nrow):
    from PIL import Image
    n, c, h, w = nchw.shape
    rows = (n + nrow - 1) // nrow
    canvas = np.zeros((rows * h, nrow * w, 3), dtype=np.uint8)
    for i in range(n):
        canvas[(i // nrow) * h:(i // nrow + 1) * h, (i % nrow) * w:(i % nrow + 1) * w] = to_uint8_image(nchw[i, :3])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(canvas).save(path)


_INFERNO = None


def _inferno_lut():
    """cv2.COLORMAP_INFERNO as RGB uint8 [256,3]: OpenCV's table is matplotlib's 256-entry inferno data scaled by 255 and
    rounded (cv2 is not installed here, so this equality is by construction of both tables, not pinned by a run)."""
    global _INFERNO
    if _INFERNO is None:
        import matplotlib
        _INFERNO = np.round(np.asarray(matplotlib.colormaps["inferno"](np.arange(256))[:, :3]) * 255).astype(np.uint8)
    return _INFERNO


def colorize_depth(depth, min=-1, max=1):
    """Depth -> inferno colours, near = bright (inference/utils.py:25-41).  Tensor in -> [N,3,H,W] float tensor out (squeezed),
    array in -> [N,H,W,3]; values mapped back to [min, max] like the reference."""
    is_tensor = isinstance(depth, torch.Tensor)
    d = depth.detach().cpu().numpy() if is_tensor else np.asarray(depth)
    d = d.squeeze()
    if d.ndim == 2:
        d = d[None]
    d = np.clip(1 - (d - min) / (max - min), 0, 1)
    col = _inferno_lut()[(d * 255).astype(np.uint8)] / 255
    if is_tensor:
        col = torch.from_numpy(col).permute(0, 3, 1, 2).float()
    col = col * (max - min) + min
    return col.squeeze()


def _png_bytes(arr):
    from PIL import Image
    with io.BytesIO() as f:
        Image.fromarray(arr).save(f, format="png")
        return f.getvalue()


def _png_array(b):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(b)))


def save_scene(path, views, modelviews, fov=45, near=0.6, far=5):
    """The reference's scene file (inference/utils.py:74-100): np.savez_compressed(path, data=[{color, depth, fov,
    modelview}, ...]) with `color` = PNG bytes of the 8-bit RGB image and `depth` = PNG bytes of the float32 METRIC depth
    reinterpreted as RGBA8.  views: [V,4,S,S] network output in [-1,1]; the metric depth is linearize_depth of the
    z-buffer channel in float32, exactly what the reference's meshes carry (sample.py:126-131).
    Difference: `modelview` is stored as a float32 [4,4] array in math order (the reference pickles a glm.mat4; PyGLM is
    not a dependency here).  load_scene accepts both."""
    v = views.detach().float().cpu().numpy().transpose(0, 2, 3, 1) * 0.5 + 0.5                    # sample.py:126
    data = []
    for i in range(v.shape[0]):
        color = np.clip(v[i, :, :, :3] * 255, 0, 255).astype(np.uint8)                             # utils.py:77
        d = np.clip(v[i, :, :, 3:], 1e-6, 1.0 - 1e-6)
        depth = np.ascontiguousarray((near * far / (far - (far - near) * d)).astype(np.float32))   # linearize_depth
        s = depth.shape[0]
        data.append({"color": _png_bytes(color), "depth": _png_bytes(np.frombuffer(depth, dtype=np.uint8).reshape(s, s, 4)),
                     "fov": fov, "modelview": np.asarray(modelviews[i], dtype=np.float32).reshape(4, 4)})
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, data=np.array(data, dtype=object))


def load_scene(path):
    """-> list of views {color [S,S,3] float in [0,1] (8-bit / 255), depth [S,S,1] float32 metric, fov, modelview [4,4]}
    (the decoded content of inference/utils.py:103-113; the meshes are rebuilt on the GPU by scene_to_renderer)."""
    data = np.load(path, allow_pickle=True)["data"]
    out = []
    for d in data:
        color = _png_array(d["color"])
        s = color.shape[0]
        depth = np.frombuffer(np.ascontiguousarray(_png_array(d["depth"])), dtype=np.float32).reshape(s, s, 1)
        mv = d["modelview"]
        mv = np.asarray(mv.to_list(), dtype=np.float32).T if hasattr(mv, "to_list") else np.asarray(mv, dtype=np.float32)
        out.append({"color": color / 255, "depth": depth, "fov": d["fov"], "modelview": mv.reshape(4, 4)})
    return out


def scene_to_renderer(renderer, scene, atol=0.03, rtol=0.03, erode_rgb=3):
    """Rebuild the meshes of a loaded scene inside a WarpRenderer (batch 1): load_scene's depth_to_mesh(depth, 32, fov,
    modelview, atol, rtol, erode_rgb, cal_normal=True) (inference/utils.py:108-111), on the GPU."""
    renderer.reset()
    for v in scene:
        rgbd = np.concatenate([v["color"].astype(np.float32), v["depth"].astype(np.float32)], axis=-1)
        t = torch.from_numpy(np.ascontiguousarray(rgbd.transpose(2, 0, 1)[None]))
        renderer.add_view(t, v["modelview"], v["fov"], atol=atol, rtol=rtol, erode_rgb=erode_rgb, padding=32, metric=True)
