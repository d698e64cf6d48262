"""inference/utils.py helpers of the reference that the sampling path needs (parse_int_list :13-22, colorize_depth
:25-41, reorder :44-55, save_scene / load_scene :74-113) plus PIL-based image writers and a video writer (the reference
uses imageio / torchvision / cv2, which are not dependencies here)."""
import io
import os
import re
import shutil
import subprocess

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


def make_grid(tensor, nrow=8, padding=2, normalize=False, value_range=None, pad_value=0.0):
    """torchvision.utils.make_grid as the reference calls it through utils.save_image (inference/sample.py:158-165: `nrow`,
    `normalize=True, value_range=(-1, 1)`, default padding 2 / pad_value 0): [N,C,H,W] -> [3, ymaps*(H+2)+2, xmaps*(W+2)+2].
    Single-channel images are repeated to 3 channels; normalize clamps to value_range and maps it to [0, 1]; image k sits at
    row k // xmaps, column k % xmaps behind a `padding`-pixel border of pad_value; ONE image is returned without a border."""
    t = torch.as_tensor(tensor).detach().float().cpu()
    if t.dim() == 3:
        t = t[None]
    if t.shape[1] == 1:
        t = t.repeat(1, 3, 1, 1)
    if normalize:
        lo, hi = (float(value_range[0]), float(value_range[1])) if value_range is not None else (float(t.min()), float(t.max()))
        t = (t.clamp(lo, hi) - lo) / max(hi - lo, 1e-5)
    if t.shape[0] == 1:
        return t[0]
    n = t.shape[0]
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    h, w = t.shape[2] + padding, t.shape[3] + padding
    grid = t.new_full((t.shape[1], h * ymaps + padding, w * xmaps + padding), pad_value)
    for k in range(n):
        y, x = divmod(k, xmaps)
        grid[:, y * h + padding:(y + 1) * h, x * w + padding:(x + 1) * w] = t[k]
    return grid


def save_grid(path, nchw, nrow, padding=2, normalize=True, value_range=(-1, 1)):
    """torchvision.utils.save_image(nchw, path, nrow=nrow, normalize=True, value_range=(-1, 1)) (inference/sample.py:158-165):
    make_grid, then `mul(255).add(0.5).clamp(0, 255)` truncated to uint8 (round to nearest), PNG."""
    from PIL import Image
    grid = make_grid(nchw, nrow=nrow, padding=padding, normalize=normalize, value_range=value_range)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    Image.fromarray(arr).save(path)


_INFERNO = None


def _inferno_lut():
    """cv2.COLORMAP_INFERNO as RGB uint8 [256,3] (committed table, see inferno_lut.py)."""
    global _INFERNO
    if _INFERNO is None:
        from .inferno_lut import INFERNO_RGB
        _INFERNO = np.asarray(INFERNO_RGB, dtype=np.uint8)
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


def _as_glm(mv):
    """Modelview for the scene file: a `glm.mat4` when PyGLM is importable -- the type the reference pickles and whose
    consumers call `glm.inverse` on (inference/utils.py:90-101, moderngl_renderer.py:309) -- else a float32 [4,4] array in
    math (row, column) order.  glm.mat4 is column-major: the 16 scalars are passed column by column."""
    m = np.asarray(mv, dtype=np.float32).reshape(4, 4)
    try:
        import glm
    except ImportError:
        return m
    return glm.mat4(*[float(v) for v in m.T.reshape(-1)])


def _from_glm(mv):
    """glm.mat4 (column-major, `to_list()` = list of columns) or array -> float32 [4,4] in math order."""
    if hasattr(mv, "to_list") and not isinstance(mv, np.ndarray):
        return np.asarray(mv.to_list(), dtype=np.float32).T.copy()
    return np.asarray(mv, dtype=np.float32).reshape(4, 4)


def save_scene(path, meshes, colors, fov=45, near=0.6, far=5):
    """The reference's scene file (inference/utils.py:74-100): np.savez_compressed(path, data=[{color, depth, fov,
    modelview}, ...]) with `color` = PNG bytes of the 8-bit RGB image and `depth` = PNG bytes of the float32 METRIC depth
    reinterpreted as RGBA8.  Two call forms:
      save_scene(path, meshes, colors)                    the reference's: meshes = depth_to_mesh dicts (.depth, .fov,
                                                          .modelview), colors = [S,S,3] float arrays in [0,1]
      save_scene(path, views, modelviews, fov, near, far) the sampling driver's: views [V,4,S,S] network output in [-1,1]
                                                          (the metric depth is linearize_depth of the z-buffer channel in
                                                          float32, what the reference's meshes carry, sample.py:126-131)
    `modelview` is stored as glm.mat4 when PyGLM is installed (readable by the reference's own load_scene / render.py),
    else as a float32 [4,4] array in math order; load_scene here accepts both."""
    data = []
    if isinstance(meshes, torch.Tensor):
        v = meshes.detach().float().cpu().numpy().transpose(0, 2, 3, 1) * 0.5 + 0.5                # sample.py:126
        modelviews = colors
        for i in range(v.shape[0]):
            color = np.clip(v[i, :, :, :3] * 255, 0, 255).astype(np.uint8)                          # utils.py:77
            d = np.clip(v[i, :, :, 3:], 1e-6, 1.0 - 1e-6)
            depth = np.ascontiguousarray((near * far / (far - (far - near) * d)).astype(np.float32))  # linearize_depth
            data.append((color, depth, fov, modelviews[i]))
    else:
        for mesh, col in zip(meshes, colors):
            color = np.clip(np.asarray(col) * 255, 0, 255).astype(np.uint8)
            data.append((color, np.ascontiguousarray(np.asarray(mesh["depth"]).astype(np.float32)), mesh["fov"], mesh["modelview"]))
    out = []
    for color, depth, f, mv in data:
        s = depth.shape[0]
        out.append({"color": _png_bytes(color), "depth": _png_bytes(np.frombuffer(depth, dtype=np.uint8).reshape(s, s, 4)),
                    "fov": f, "modelview": _as_glm(_from_glm(mv))})
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, data=np.array(out, dtype=object))


def read_scene(path):
    """Decoded content of a scene file: list of views {color [S,S,3] float in [0,1] (8-bit / 255), depth [S,S,1] float32
    metric, fov, modelview float32 [4,4] math order}.  No GPU needed."""
    data = np.load(path, allow_pickle=True)["data"]
    out = []
    for d in data:
        color = _png_array(d["color"])
        s = color.shape[0]
        depth = np.frombuffer(np.ascontiguousarray(_png_array(d["depth"])), dtype=np.float32).reshape(s, s, 1)
        out.append({"color": color / 255, "depth": depth, "fov": d["fov"], "modelview": _from_glm(d["modelview"])})
    return out


def load_scene(path, atol=0.03, rtol=0.03, erode_rgb=3):
    """inference/utils.py:103-113, same return: (meshes, colors) with meshes = depth_to_mesh(depth, 32, fov, modelview,
    atol, rtol, erode_rgb, cal_normal=True) per stored view (built on the GPU) and colors = [S,S,3] float in [0,1]."""
    from .. import rgbd_3d
    views = read_scene(path)
    meshes = [rgbd_3d.utils.depth_to_mesh(v["depth"], 32, v["fov"], v["modelview"], atol=atol, rtol=rtol, erode_rgb=erode_rgb,
                                          cal_normal=True) for v in views]
    return meshes, [v["color"] for v in views]


def scene_to_renderer(renderer, scene, atol=0.03, rtol=0.03, erode_rgb=3):
    """The device-resident form of load_scene: rebuild the meshes of a decoded scene (read_scene) inside a WarpRenderer
    (batch 1) without a host round trip per mesh."""
    renderer.reset()
    for v in scene:
        rgbd = np.concatenate([v["color"].astype(np.float32), v["depth"].astype(np.float32)], axis=-1)
        t = torch.from_numpy(np.ascontiguousarray(rgbd.transpose(2, 0, 1)[None]))
        renderer.add_view(t, v["modelview"], v["fov"], atol=atol, rtol=rtol, erode_rgb=erode_rgb, padding=32, metric=True)


def write_video(path, frames, fps=30):
    """`imageio.mimsave(path.mp4, frames, fps=30)` of inference/render.py:87-88 without imageio: raw RGB frames piped into
    an `ffmpeg` executable (libx264, yuv420p) when one is on PATH; else imageio if it imports; else an animated GIF next to
    the requested name.  frames: uint8 [F,H,W,3].  Returns the path actually written."""
    frames = np.ascontiguousarray(np.asarray(frames, dtype=np.uint8))
    f, h, w, _ = frames.shape
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    exe = shutil.which("ffmpeg")
    if exe is not None:
        cmd = [exe, "-y", "-loglevel", "error", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{w}x{h}", "-r", str(fps), "-i", "-",
               "-an", "-vcodec", "libx264", "-pix_fmt", "yuv420p", path]
        p = subprocess.run(cmd, input=frames.tobytes(), capture_output=True)
        if p.returncode == 0:
            return path
        print("ffmpeg failed, falling back:", p.stderr.decode(errors="replace")[-300:])
    try:
        import imageio
        imageio.mimsave(path, list(frames), fps=fps)
        return path
    except ImportError:
        pass
    from PIL import Image
    gif = os.path.splitext(path)[0] + ".gif"
    imgs = [Image.fromarray(a) for a in frames]
    imgs[0].save(gif, save_all=True, append_images=imgs[1:], duration=int(round(1000 / fps)), loop=0)
    return gif
