"""Free-view fusion rendering of generated scenes — the reference's inference/render.py (:17-88) on the HIP warp kernels.

Same CLI flags and defaults.  Per scene: load_scene -> meshes rebuilt on the GPU (numeric padding 32) -> every frame of
the camera trajectory is one `WarpRenderer.render` at SSAA 5 (640^2, near 0.1 / far 200, render.py:62-64) -> 8-bit
quantise + Pillow LANCZOS to 128^2 and inferno-coloured projected depth, exactly the reference's post-processing
(:73-84).  Videos: `<scene>.mp4` / `<scene>_depth.mp4` at 30 fps like render.py:87-88 through an `ffmpeg` executable (or
imageio) when present; an animated GIF otherwise (utils.write_video)."""
import argparse
import glob
import os

import numpy as np

from .. import rgbd_3d
from ..rgbd_3d import camera
from .utils import colorize_depth, read_scene, scene_to_renderer, write_video


def trajectory(name, frames, num_scenes, rng=None):
    """Camera paths of render.py:41-61 (4x4 math-order modelview matrices)."""
    if name == "swing":
        ts = np.linspace(0, 2 * np.pi, frames)
        return [camera.orbit(0.6 * np.cos(t), 0.15 * np.sin(t)) for t in ts]
    if name == "random":
        rng = rng or np.random
        out = []
        for _ in range(num_scenes):
            yaw = np.clip(0.3 * rng.normal(), -0.6, 0.6)
            pitch = np.clip(0.15 * rng.normal(), -0.15, 0.15)
            out.append([camera.orbit(yaw, pitch)])
        return out
    raise NotImplementedError(name)


def render_scene(renderer, scene, modelviews, atol=0.03, rtol=0.03, erode_rgb=3, ssaa=5):
    """-> (colors uint8 [F,128,128,3], depths uint8 [F,128,128,3]) for one scene (render.py:66-84)."""
    from PIL import Image
    scene_to_renderer(renderer, scene, atol, rtol, erode_rgb)
    S, off = renderer.image_size, ssaa // 2
    colors, depths = [], []
    for mv in modelviews:
        res = renderer.render(mv)
        c8 = res.color8[0].cpu().numpy()                      # == (res['color'] * 255).astype(np.uint8), render.py:79
        colors.append(np.asarray(Image.fromarray(c8).resize((S, S), Image.Resampling.LANCZOS)))
        d = res.depth[0].cpu().numpy()[off::ssaa, off::ssaa]
        depths.append((colorize_depth(rgbd_3d.utils.project_depth(d), min=0, max=1) * 255).astype(np.uint8))
    return np.stack(colors), np.stack(depths)


def main(argv=None):
    from PIL import Image
    p = argparse.ArgumentParser()
    p.add_argument("--scene_dir", type=str, default="samples/imagenet128/viewset_3x9_steps_u1000_c50_guidance3.0")
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--frames", type=int, default=60)
    p.add_argument("--traj", type=str, default="swing")
    p.add_argument("--atol", type=float, default=0.03)
    p.add_argument("--rtol", type=float, default=0.03)
    p.add_argument("--erode_rgb", type=int, default=3)
    opt = p.parse_args(argv)
    if opt.output_dir is None:
        opt.output_dir = opt.scene_dir
    os.makedirs(os.path.join(opt.output_dir, "results"), exist_ok=True)
    os.makedirs(os.path.join(opt.output_dir, "videos"), exist_ok=True)
    scenes = sorted(glob.glob(os.path.join(opt.scene_dir, "scenes", "*.npz")))
    print(f"Found {len(scenes)} scenes.")
    mvs = trajectory(opt.traj, opt.frames, len(scenes))
    ssaa, renderer = 5, None
    for i, path in enumerate(scenes):
        scene = read_scene(path)
        S = scene[0]["color"].shape[0]
        if renderer is None or renderer.max_views < len(scene) or renderer.image_size != S:
            renderer = rgbd_3d.WarpRenderer(1, S, ssaa, max(27, len(scene)), near=0.1, far=200.0)
        colors, depths = render_scene(renderer, scene, mvs[i] if isinstance(mvs[0], list) else mvs, opt.atol, opt.rtol,
                                      opt.erode_rgb, ssaa)
        name = os.path.basename(path)[:-4]
        if opt.traj == "random":
            Image.fromarray(colors[0]).save(os.path.join(opt.output_dir, "results", f"{name}.png"))
        else:
            for arr, suffix in ((colors, ""), (depths, "_depth")):
                write_video(os.path.join(opt.output_dir, "videos", f"{name}{suffix}.mp4"), arr, fps=30)


if __name__ == "__main__":
    main()
