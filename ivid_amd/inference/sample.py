"""Iterative multiview sampling driver — the MI355X counterpart of /root/reference/inference/sample.py.

Same CLI flags and defaults (sample.py:242-263), same config JSONs / checkpoints, same by-name model creation
(`getattr(backbones, cfg.backbone.name)(**cfg.backbone.args)`, :183-192), same seed -> noise rule (:64-71), same
rank-strided partition over GPUs (:199-202).  What changes is where the work happens: the per-view conditioning
(depth -> mesh -> warp -> aggregate -> masks) is a few HIP launches for the whole batch instead of a serial CPU/OpenGL
loop over samples, nothing round-trips through host memory between views, and every rank gets its weights from one RCCL
broadcast.  Launch with `torchrun --nproc-per-node N -m ivid_amd.inference.sample ...` for N GPUs.
"""
import argparse
import json
import os
import threading

import numpy as np
import torch

from .. import parallel, rgbd_3d
from ..diffusion import backbones, frameworks, samplers
from ..utils import AttrDict
from .superres import super_resolve
from .utils import colorize_depth, parse_int_list, reorder, save_grid, save_png, save_scene


@torch.no_grad()
def sample_all(framework_uncond, framework_cond, seeds_or_num_samples, steps_uncond, steps_cond, modelviews, fov=45,
               near=0.6, far=5, atol=0.03, rtol=0.03, erode_rgb=2, classes=None, guidance=3.0, batchsize=10, noise_fn=None):
    """Generator over samples: yields (views [V,4,S,S] in [-1,1], conds {'color','depth'} [V-1,...] or None, renderer
    state is per batch).  Mirrors sample_all of the reference (sample.py:30-147)."""
    device = framework_uncond.backbone.device
    S = framework_uncond.backbone.image_size
    sampler_uncond = samplers.DdimSampler(framework_uncond) if steps_uncond < 1000 else samplers.DdpmSampler(framework_uncond)
    sampler_cond = samplers.DdimSampler(framework_cond) if framework_cond is not None else None
    num_samples = seeds_or_num_samples if not isinstance(seeds_or_num_samples, list) else len(seeds_or_num_samples)
    seeds = seeds_or_num_samples if isinstance(seeds_or_num_samples, list) else None
    is_cfg = isinstance(framework_uncond, frameworks.ClassifierFreeGuidance)
    renderers = {}
    for i in range(0, num_samples, batchsize):
        bs = min(batchsize, num_samples - i)
        s_modelviews = modelviews[i] if isinstance(modelviews[0], list) else modelviews   # sample.py:74 (batch shares cameras)
        if seeds is not None:
            noise = []
            for j in range(bs):
                torch.manual_seed(seeds[i + j])
                noise.append(torch.randn(1, 4, S, S, device=device))
            noise = torch.cat(noise, dim=0)
        else:
            noise = None
        b_classes = torch.tensor(classes[i:i + bs]).long().to(device) if classes is not None else None
        # Noise inside the chain (hole noise of InpaintCFG, eta > 0, DDPM).  The reference draws it from the process-wide
        # CUDA generator, so a sample's result depends on which other samples share its batch / GPU.  With seeds, every
        # sample gets its own generator instead: results are invariant under the rank / batch partition.
        if noise_fn is not None:
            extra = {"noise_fn": noise_fn}
        elif seeds is not None:
            gens = [torch.Generator(device=device).manual_seed(0x5EED0000 + int(seeds[i + j])) for j in range(bs)]
            extra = {"noise_fn": lambda shape, g=gens: torch.cat(
                [torch.randn((1,) + tuple(shape[1:]), device=device, generator=gj) for gj in g], dim=0)}
        else:
            extra = {}
        if os.environ.get("IVID_DEVICE_LOOP", "0") == "1":   # every chain as ONE C call (ivid_sample); bit-identical samples
            extra["device_loop"] = True
        if sampler_cond is not None:
            key = (bs, len(s_modelviews))
            if key not in renderers:
                renderers[key] = rgbd_3d.WarpRenderer(bs, S, 3, len(s_modelviews), device=device)
            renderer = renderers[key]
            renderer.reset()
        samples, conds = [], {"color": [], "depth": []}
        for j, modelview in enumerate(s_modelviews):
            if j == 0:
                kw = dict(strength=guidance) if is_cfg else {}
                res = sampler_uncond.sample(bs, noise=noise, classes=b_classes, steps=steps_uncond, verbose=False,
                                            keep_intermediates=False, **kw, **extra)
            else:
                c = renderer.conditions(modelview, fov, near, far, atol, rtol, erode_rgb)
                color, depth = c.color * 2 - 1, c.depth * 2 - 1                            # sample.py:102-103
                conds["color"].append(color)
                conds["depth"].append(depth)
                args = {                                                                   # sample.py:104-120
                    "y": torch.cat([color, depth], dim=1), "mask": c.mask, "mask_rgb": c.mask_rgb,
                    "replace_rgb": (0.1, color, c.mask_rgb), "replace_depth": (0.2, depth, c.mask),
                    "constrain_depth": (0.5, c.depth_convex * 2 - 1),
                }
                kw = dict(strength=guidance) if is_cfg else {}
                res = sampler_cond.sample(bs, classes=b_classes, steps=steps_cond, verbose=False, keep_intermediates=False,
                                          **kw, **args, **extra)
            samples.append(res.samples)
            if sampler_cond is not None:
                renderer.add_view(res.samples, modelview, fov, near, far, atol, rtol, erode_rgb)   # sample.py:128-139
        samples = torch.stack(samples, dim=1)
        cstack = {k: torch.stack(v, dim=1) for k, v in conds.items()} if conds["color"] else None
        for j in range(bs):
            yield samples[j], ({k: v[j] for k, v in cstack.items()} if cstack is not None else None)


def async_save(samples, conds, suffix, cfg, modelviews):
    """Writer thread with the reference's retry-and-swallow behaviour and output tree (sample.py:150-176): results/,
    grids/ (rgb + inferno depth, 3x9 order), conds/ and scenes/*.npz in the reference's scene wire format.  Tensors are
    complete: the caller synchronises before handing them over."""
    samples = samples.cpu()
    conds = {k: v.cpu() for k, v in conds.items()} if conds is not None else None

    def worker():
        for _ in range(10):
            try:
                out = cfg.output_dir
                scene = os.path.join(out, "scenes", f"scene_{suffix}.npz")
                if cfg.viewset == "uncond":
                    save_png(os.path.join(out, "results", f"rgb_{suffix}.png"), samples[0, :3])
                    save_scene(scene, samples, modelviews, cfg.fov, cfg.near, cfg.far)
                elif cfg.viewset == "random":
                    save_grid(os.path.join(out, "grids", f"rgb_{suffix}.png"), samples[:, :3], 2)
                    save_png(os.path.join(out, "conds", f"rgb_{suffix}.png"), samples[0, :3])
                    save_png(os.path.join(out, "results", f"rgb_{suffix}.png"), samples[1, :3])
                else:
                    save_grid(os.path.join(out, "grids", f"rgb_{suffix}.png"), reorder(samples[:, :3], cfg.viewset), 9)
                    save_grid(os.path.join(out, "grids", f"depth_{suffix}.png"),
                              reorder(colorize_depth(samples[:, 3:]), cfg.viewset), 9)
                    save_grid(os.path.join(out, "conds", f"rgb_cond_{suffix}.png"), reorder(conds["color"][:, :3], cfg.viewset), 9)
                    save_grid(os.path.join(out, "conds", f"depth_cond_{suffix}.png"),
                              reorder(colorize_depth(conds["depth"]), cfg.viewset), 9)
                    save_scene(scene, samples, modelviews, cfg.fov, cfg.near, cfg.far)
                break
            except Exception as e:  # noqa: BLE001
                print(e)
    t = threading.Thread(target=worker)
    t.start()
    return t


def build_model(cfg_model, ckpt_path, device, precision=None):
    """Model creation by name + checkpoint loaded on rank 0 and broadcast over RCCL (one message per model)."""
    args = dict(cfg_model["backbone"]["args"])
    if precision is not None:
        args["precision"] = precision
    backbone = getattr(backbones, cfg_model["backbone"]["name"])(**args)
    rank, _ = parallel.rank_world()
    schema = [(k, tuple(v.shape)) for k, v in backbone.state_dict().items()]
    sd = torch.load(ckpt_path, map_location="cpu") if rank == 0 else None
    backbone.load_state_dict(parallel.broadcast_state_dict(schema, sd, device=device))
    backbone = backbone.to(device)
    return getattr(frameworks, cfg_model["framework"]["name"])(backbone, **cfg_model["framework"]["args"])


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config_uncond", type=str, default="configs/rgbd_imagenet_adm_128_large_cfg.json")
    p.add_argument("--config_cond", type=str, default="configs/rgbd_imagenet_adm_128_large_cond.json")
    p.add_argument("--ckpt_uncond", type=str, default="ckpts/imagenet128_uncond.pt")
    p.add_argument("--ckpt_cond", type=str, default="ckpts/imagenet128_cond.pt")
    p.add_argument("--output_dir", type=str, default="samples/imagenet128")
    p.add_argument("--seeds", type=str, default="0-8")
    p.add_argument("--num_samples", type=int, default=None)
    p.add_argument("--classes", type=str, default="mod")
    p.add_argument("--viewset", type=str, default="3x9")
    p.add_argument("--steps_uncond", type=int, default=1000)
    p.add_argument("--steps_cond", type=int, default=50)
    p.add_argument("--guidance", type=float, default=3.0)
    p.add_argument("--batchsize", type=int, default=10)
    p.add_argument("--fov", type=float, default=45)
    p.add_argument("--near", type=float, default=0.6)
    p.add_argument("--far", type=float, default=5)
    p.add_argument("--atol", type=float, default=0.03)
    p.add_argument("--rtol", type=float, default=0.03)
    p.add_argument("--erode_rgb", type=int, default=3)
    p.add_argument("--precision", type=str, default=None, help="fp32 | bf16x3 | fp16s (fp16 MFMA, compensated trunk, trunk-critical layers in split precision: within 1e-3 of fp32 on "
                        "every input) | fp16sa (adaptive: fp16s without its first-level island at timesteps >= 150: +10 %%) | fp16sa3 (+ plain fp16cx "
                        "from t >= 500: unconditional 128^2 backbones) | fp16sx (strict ladder: both parity metrics under 1e-3) | fp16cx | "
                        "fp16c | fp16 | bf16; default: fp16sx if the config says use_fp16 else fp32")
    # 128 -> 256 super-resolution of every generated view (BASELINE config 5).  Not in the reference CLI: the reference ships
    # the SR model and SuperResCFG but no inference driver for them (only SuperResTrainer.sample, trainers/superres.py:97-134)
    p.add_argument("--config_sr", type=str, default=None, help="e.g. configs/rgbd_imagenet_adm_256_128_small_sr.json")
    p.add_argument("--ckpt_sr", type=str, default=None)
    p.add_argument("--steps_sr", type=int, default=50)
    p.add_argument("--guidance_sr", type=float, default=3.0)
    p.add_argument("--batchsize_sr", type=int, default=27, help="views per SR batch (27 = a whole 3x9 viewset: +3.5 %% per view over 16 + 11)")
    opt = p.parse_args(argv)
    cfg = AttrDict(vars(opt))
    with open(opt.config_uncond) as f:
        cfg_uncond = json.load(f)
    with open(opt.config_cond) as f:
        cfg_cond = json.load(f)
    cfg.output_dir = os.path.join(cfg.output_dir, f"viewset_{cfg.viewset}_steps_u{cfg.steps_uncond}_c{cfg.steps_cond}_guidance{cfg.guidance}")
    for d in ("scenes", "conds", "grids", "results"):
        os.makedirs(os.path.join(cfg.output_dir, d), exist_ok=True)

    rank, world = parallel.init_from_env()
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)

    if cfg.num_samples is not None:
        num_samples, seeds = cfg.num_samples, None
    else:
        seeds = parse_int_list(cfg.seeds)
        num_samples = len(seeds)
    # the reference draws random classes / cameras ONCE in the parent and passes the lists to every rank (sample.py:296-336):
    # here every rank draws them itself from one common seed, then takes its shard of the same lists
    # (a local generator: the caller's global numpy RNG state is left alone)
    draw = np.random.RandomState(parallel.common_draw_seed() % (1 << 32))
    classes = None
    num_classes = cfg_uncond["backbone"]["args"].get("num_classes")
    if num_classes is not None:
        if cfg.classes == "mod":
            classes = [seeds[i] % num_classes for i in range(num_samples)]
        elif cfg.classes == "random":
            classes = [int(draw.randint(num_classes)) for _ in range(num_samples)]
        elif cfg.classes == "uniform":
            classes = [i % num_classes for i in range(num_samples)]
        else:
            classes = parse_int_list(cfg.classes)
    modelviews = rgbd_3d.camera.viewset(cfg.viewset, num_samples, rng=draw)

    fw_uncond = build_model(cfg_uncond, cfg.ckpt_uncond, device, cfg.precision)
    fw_cond = build_model(cfg_cond, cfg.ckpt_cond, device, cfg.precision) if cfg.viewset != "uncond" else None
    fw_sr = None
    if cfg.config_sr is not None:
        with open(cfg.config_sr) as f:
            fw_sr = build_model(json.load(f), cfg.ckpt_sr, device, cfg.precision)
        os.makedirs(os.path.join(cfg.output_dir, "results_sr"), exist_ok=True)

    # rank-strided partition, identical to sample.py:199-202
    seeds_r = parallel.shard(seeds, rank, world)
    idx = np.arange(cfg.num_samples)[rank::world] if cfg.num_samples is not None else None
    classes_r = parallel.shard(classes, rank, world)
    views_r = parallel.shard_views(modelviews, rank, world)
    gen = sample_all(fw_uncond, fw_cond, seeds_r if seeds_r is not None else len(idx), cfg.steps_uncond, cfg.steps_cond, views_r,
                     classes=classes_r, guidance=cfg.guidance, batchsize=cfg.batchsize, fov=cfg.fov, near=cfg.near, far=cfg.far,
                     atol=cfg.atol, rtol=cfg.rtol, erode_rgb=cfg.erode_rgb)
    threads = []
    for i, (samples, conds) in enumerate(gen):
        parts = []
        if classes_r is not None:
            parts.append(f"class{classes_r[i]:03d}")
        parts.append(f"seed{seeds_r[i]:05d}" if seeds_r is not None else f"{idx[i]:05d}")
        if fw_sr is not None:   # the chain uncond -> warp/inpaint views -> SR stays on the GPU
            hi = super_resolve(fw_sr, samples, classes=classes_r[i] if classes_r is not None else None, steps=cfg.steps_sr,
                               strength=cfg.guidance_sr, batchsize=cfg.batchsize_sr)
            torch.cuda.synchronize(device)
            hi = hi.cpu()
            for v in range(hi.shape[0]):
                save_png(os.path.join(cfg.output_dir, "results_sr", f"rgb_{'_'.join(parts)}_view{v:02d}.png"), hi[v, :3])
        torch.cuda.synchronize(device)
        mv_i = views_r[i] if isinstance(views_r[0], list) else views_r
        threads.append(async_save(samples, conds, "_".join(parts), cfg, mv_i))
    for t in threads:
        t.join()


def cli(argv=None):
    """Entry point: one process per GPU like the reference's mp.spawn (sample.py:340-348) unless a launcher already
    made this process a rank."""
    import sys
    argv = sys.argv[1:] if argv is None else list(argv)
    rc = parallel.spawn_one_process_per_gpu("ivid_amd.inference.sample", argv, module=True)
    if rc is not None:
        raise SystemExit(rc)
    main(argv)


if __name__ == "__main__":
    cli()
