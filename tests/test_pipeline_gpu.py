"""GPU: the whole iterative multiview loop (uncond sample -> mesh -> warp -> conditional inpainting) at mini scale,
and the sharding invariant behind the sample-parallel multi-GPU mode: a sample's result does not depend on which
batch / rank it was generated in."""
import numpy as np
import pytest
import torch

import common as C

pytestmark = pytest.mark.gpu


def _frameworks():
    from ivid_amd.diffusion import frameworks
    from ivid_amd.diffusion.backbones import AdmUnet2d
    mu = AdmUnet2d(**C.MINI, precision="fp32"); mu.load_state_dict(C.synth_weights(C.MINI, 0)); mu = mu.cuda()
    mc = AdmUnet2d(**C.MINI_COND, precision="fp32"); mc.load_state_dict(C.synth_weights(C.MINI_COND, 2)); mc = mc.cuda()
    return (frameworks.ClassifierFreeGuidance(mu, timesteps=1000, beta_schedule="linear", p_uncond=0.1),
            frameworks.InpaintCFG(mc, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0))


def _run(fu, fc, seeds, classes, batchsize, views):
    from ivid_amd.inference.sample import sample_all
    out = list(sample_all(fu, fc, seeds, 4, 3, views, classes=classes, guidance=0.5, batchsize=batchsize, erode_rgb=1))
    return [o[0].cpu() for o in out], [o[1] for o in out]


def test_multiview_loop_runs_and_is_partition_invariant():
    from ivid_amd import parallel
    from ivid_amd.rgbd_3d import camera
    fu, fc = _frameworks()
    views = camera.viewset("3x9")[:3]
    seeds, classes = [0, 1, 2, 3, 4], [0, 1, 2, 3, 4]
    full, conds = _run(fu, fc, seeds, classes, 5, views)
    assert len(full) == 5 and full[0].shape == (3, 4, 32, 32) and all(torch.isfinite(s).all() for s in full)
    assert conds[0]["color"].shape == (2, 3, 32, 32) and conds[0]["depth"].shape == (2, 1, 32, 32)
    # two "ranks" with the reference's strided partition, different batch sizes
    for rank in range(2):
        part, _ = _run(fu, fc, parallel.shard(seeds, rank, 2), parallel.shard(classes, rank, 2), 2, views)
        for k, s in zip(parallel.shard(list(range(5)), rank, 2), part):
            assert torch.allclose(s, full[k], rtol=1e-4, atol=1e-4), (rank, k, float((s - full[k]).abs().max()))


def test_random_viewset_and_uncond_only():
    from ivid_amd.rgbd_3d import camera
    fu, fc = _frameworks()
    views = camera.viewset("random", 2, np.random.default_rng(0))
    out, conds = _run(fu, fc, [7, 8], [1, 2], 2, views)
    assert out[0].shape == (2, 4, 32, 32) and conds[0]["color"].shape[0] == 1
    out, conds = _run(fu, None, [7], [1], 1, camera.viewset("uncond"))
    assert out[0].shape == (1, 4, 32, 32) and conds[0] is None


def test_sample_all_chain_against_the_cpu_oracle_chain():
    """sample_all (inference/sample.py:30-147) end to end at mini scale vs the CPU oracle of the SAME chain: unconditional
    DDIM -> depth_to_mesh -> aggregate_conditions at the next camera -> the reference's conditioning wiring (`*2-1`, y /
    mask / mask_rgb, replace_rgb 0.1 / replace_depth 0.2 / constrain_depth 0.5, sample.py:99-120) -> conditional DDIM with
    InpaintCFG -> ... for three views.  Oracle = sampler_oracle + adm_oracle + warp_oracle (C rasteriser), noise streams
    injected identically on both sides."""
    import gpu_util as G
    import warp_common as WC
    from ivid_amd.inference.sample import sample_all
    from ivid_amd.rgbd_3d import camera
    from oracle import adm_oracle, sampler_oracle as SO, warp_oracle as W
    fu, fc = _frameworks()
    S, su, sc, g = 32, 4, 3, 0.5
    vs = camera.viewset("3x9")
    views = [vs[0], vs[3], vs[7]]                  # front, yaw +0.15, yaw -0.15 / pitch +0.15
    seeds, classes = [3, 4], [1, 2]
    near, far, atol, rtol, erode = 0.6, 5.0, 0.03, 0.03, 1
    cpu_noise = lambda shape: torch.randn(shape).cuda()
    torch.manual_seed(1234)
    out = list(sample_all(fu, fc, seeds, su, sc, views, classes=classes, guidance=g, batchsize=2, near=near, far=far, atol=atol,
                          rtol=rtol, erode_rgb=erode, noise_fn=cpu_noise))
    got = torch.stack([o[0] for o in out]).cpu()                                     # [B,V,4,S,S]
    gcond = {k: torch.stack([o[1][k] for o in out]).cpu() for k in ("color", "depth")}  # [B,V-1,C,S,S], already *2-1
    # ---- the oracle chain ----
    xs = []
    for s in seeds:                                                                  # sample.py:64-71: per-seed device noise
        torch.manual_seed(s)
        xs.append(torch.randn(1, 4, S, S, device="cuda").cpu())
    x_T = torch.cat(xs)
    sdu, sdc = C.synth_weights(C.MINI, 0), C.synth_weights(C.MINI_COND, 2)
    uu = lambda x, t, c: adm_oracle.unet_forward(sdu, C.MINI, x, t, c)
    uc = lambda x, t, c: adm_oracle.unet_forward(sdc, C.MINI_COND, x, t, c)
    cls = torch.tensor(classes)
    betas = SO.linear_betas(1000)
    # NOTE the per-seed torch.manual_seed above also reseeds the CPU generator (sample.py:66 does the same): the injected
    # noise stream of the chain starts from manual_seed(seeds[-1]) on both sides
    prev = [SO.ddim_sample(lambda x, t: SO.cfg_eps(uu, x, t, cls, g), x_T, su, betas)["samples"]]
    e0 = C.rel_l2(got[:, 0], prev[0])
    errs, cond_close = {"view0": e0}, {}
    for j in range(1, len(views)):
        cs = []
        for b in range(len(seeds)):
            meshes, cols = zip(*[WC.oracle_mesh(prev[k][b].numpy(), views[k]) for k in range(j)])
            # oracle_mesh uses erode_rgb 3; rebuild with this test's erode_rgb (flags differ)
            meshes = []
            for k in range(j):
                hw = prev[k][b].numpy().transpose(1, 2, 0) * 0.5 + 0.5
                meshes.append(W.depth_to_mesh(W.linearize_depth(hw[:, :, 3:], near, far), 45, views[k], atol, rtol, erode))
            cs.append(W.aggregate_conditions(list(meshes), list(cols), views[j], S, 3, 45, near, far, atol, rtol, erode))
        T = lambda k: torch.from_numpy(np.stack([c[k] for c in cs])).permute(0, 3, 1, 2).float()
        # (i) the conditioning tensors sample_all used = a fresh WarpRenderer fed with the chain's views (deterministic)
        from ivid_amd.rgbd_3d import WarpRenderer
        wr = WarpRenderer(len(seeds), S, 3, len(views))
        for k in range(j):
            wr.add_view(got[:, k].cuda(), views[k], 45, near, far, atol, rtol, erode)
        c = wr.conditions(views[j], 45, near, far, atol, rtol, erode)
        color, depth = (c.color * 2 - 1).cpu(), (c.depth * 2 - 1).cpu()                  # sample.py:102-103
        assert torch.equal(color, gcond["color"][:, j - 1]) and torch.equal(depth, gcond["depth"][:, j - 1])
        mask, mask_rgb, convex = c.mask.cpu(), c.mask_rgb.cpu(), (c.depth_convex * 2 - 1).cpu()
        # (ii) the oracle's warp (C rasteriser) of the same views: untrained models emit saturated, noise-like depth -- a
        # mesh made of discontinuity sheets -- and the two rasterisers still have to agree on it
        dc = (color - (T("color") * 2 - 1)).abs().amax(1)
        dd = (depth - (T("depth") * 2 - 1)).abs().amax(1)
        cond_close[f"cond{j}_color_frac_exact"] = float((dc < 1e-6).float().mean())
        cond_close[f"cond{j}_depth_frac_1e-4"] = float((dd < 1e-4).float().mean())
        cond_close[f"cond{j}_mask_agree"] = float((mask == T("mask")).float().mean())
        assert (dc < 1e-2).float().mean() > 0.99 and (dd < 1e-3).float().mean() > 0.99, (j, cond_close)
        assert cond_close[f"cond{j}_mask_agree"] > 0.99
        # (iii) the reference's wiring + conditional sampler on those tensors (sample.py:99-120)
        y = torch.cat([color, depth], dim=1)
        x2 = torch.randn(len(seeds), 4, S, S)                                        # the conditional chain's x_T (noise stream)
        res = SO.ddim_sample(lambda x, t: SO.inpaint_cfg_eps(uc, x, t, y, mask, cls, g, mask_rgb), x2, sc, betas,
                             replace_rgb=(0.1, color, mask_rgb), replace_depth=(0.2, depth, mask), constrain_depth=(0.5, convex))
        prev.append(res["samples"])
        errs[f"view{j}"] = C.rel_l2(got[:, j], prev[j])
    G.report("chain/sample_all_vs_oracle", **errs, **cond_close)
    assert all(e < 1e-3 for e in errs.values()), errs


def test_cli_main_writes_the_reference_output_tree(tmp_path):
    """inference/sample.py `main` (sample.py:179-338) end to end on one GPU at mini scale: reference-format config JSONs and
    checkpoints (bare state_dicts written with torch.save) in, the reference's output tree out (results/ grids/ conds/
    scenes/ under viewset_<..>_steps_u<..>_c<..>_guidance<..>, file names class<ccc>_seed<sssss>), scene files readable."""
    import json
    import os
    from ivid_amd.inference import sample as S_, utils as U
    cfgs = {}
    for tag, args, fw, fwargs, seed in (("uncond", C.MINI, "ClassifierFreeGuidance", {"p_uncond": 0.1}, 0),
                                        ("cond", C.MINI_COND, "InpaintCFG", {"p_uncond": 0.1, "p_uncond_img": 0}, 2)):
        cfg = {"backbone": {"name": "AdmUnet2d", "args": dict(args, use_fp16=True)},
               "framework": {"name": fw, "args": dict(timesteps=1000, beta_schedule="linear", **fwargs)}}
        cfgs[tag] = str(tmp_path / f"{tag}.json")
        json.dump(cfg, open(cfgs[tag], "w"))
        torch.save(C.synth_weights(args, seed), str(tmp_path / f"{tag}.pt"))
    out = str(tmp_path / "out")
    argv = ["--config_uncond", cfgs["uncond"], "--config_cond", cfgs["cond"], "--ckpt_uncond", str(tmp_path / "uncond.pt"),
            "--ckpt_cond", str(tmp_path / "cond.pt"), "--output_dir", out, "--seeds", "3,5-6", "--classes", "mod", "--viewset", "3x9",
            "--steps_uncond", "3", "--steps_cond", "2", "--guidance", "0.5", "--batchsize", "2", "--erode_rgb", "1"]
    S_.main(argv)
    root = os.path.join(out, "viewset_3x9_steps_u3_c2_guidance0.5")
    for seed in (3, 5, 6):
        name = f"class{seed % 10:03d}_seed{seed:05d}"
        for sub, f in (("grids", f"rgb_{name}.png"), ("grids", f"depth_{name}.png"), ("conds", f"rgb_cond_{name}.png"),
                       ("conds", f"depth_cond_{name}.png"), ("scenes", f"scene_{name}.npz")):
            assert os.path.isfile(os.path.join(root, sub, f)), (sub, f)
        scene = U.read_scene(os.path.join(root, "scenes", f"scene_{name}.npz"))
        assert len(scene) == 27 and scene[0]["color"].shape == (32, 32, 3) and scene[26]["modelview"].shape == (4, 4)
    from PIL import Image
    g = Image.open(os.path.join(root, "grids", "rgb_class003_seed00003.png"))
    assert g.size == (9 * 34 + 2, 3 * 34 + 2)               # torchvision grid: 3 rows x 9 columns, 2-px borders (sample.py:161-165)
    # the models came up in the reference's precision for use_fp16 configs
    assert S_.parse_int_list("3,5-6") == [3, 5, 6]


def test_sample_all_against_the_references_own_sample_all():
    """tests/golden/sample_all_ref.npz = /root/reference/inference/sample.py `sample_all` ITSELF, executed in the build container
    (tests/golden/make_golden_sample_all.py: host tensors, the reference's AggregationRenderer on real OpenGL through
    oracle/glshim, one seeded CPU noise stream): unconditional DDIM + CFG -> mesh -> aggregate_conditions -> the wiring of
    sample.py:99-120 -> conditional DDIM with InpaintCFG, three views of the `3x9` viewset at the reference's hard-wired
    128 x 128.  The product replays the same noise stream through `noise_fn` (same draws, same order -- checked) and must
    reproduce the views and the conditioning tensors.  Untrained weights saturate the samples (|x| ~ 700), so the depth
    maps are noise-like and every mesh is made of discontinuity sheets: the hardest input the warp can get."""
    import gpu_util as G
    from ivid_amd.diffusion import frameworks
    from ivid_amd.diffusion.backbones import AdmUnet2d
    from ivid_amd.inference.sample import sample_all
    from ivid_amd.rgbd_3d import camera
    g = C.load_golden("sample_all_ref")
    mu = AdmUnet2d(**C.MINI128, precision="fp32"); mu.load_state_dict(C.synth_weights(C.MINI128, 0)); mu = mu.cuda()
    mc = AdmUnet2d(**C.MINI128_COND, precision="fp32"); mc.load_state_dict(C.synth_weights(C.MINI128_COND, 2)); mc = mc.cuda()
    fu = frameworks.ClassifierFreeGuidance(mu, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    fc = frameworks.InpaintCFG(mc, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(g["noise_seed"]))
    expected = [tuple(int(v) for v in str(d).split("x")) for d in g["draws"]]
    drawn = []

    def noise_fn(shape):
        shape = tuple(int(v) for v in shape)
        assert len(drawn) < len(expected) and shape == expected[len(drawn)], (len(drawn), shape)
        drawn.append(shape)
        return torch.randn(shape, generator=gen).cuda()
    vs = camera.viewset("3x9")
    views = [vs[int(k)] for k in g["view_ids"]]
    su, sc, erode = (int(v) for v in g["cfg"])
    out = list(sample_all(fu, fc, 1, su, sc, views, classes=[int(c) for c in g["classes"]], guidance=float(g["guidance"]),
                          batchsize=1, erode_rgb=erode, noise_fn=noise_fn))
    assert drawn == expected                                  # the same random draws in the same order as the reference
    samples, conds = out[0][0].cpu(), {k: v.cpu() for k, v in out[0][1].items()}
    errs = {f"view{j}": C.rel_l2(samples[j], g["samples"][j]) for j in range(3)}
    for j in range(2):
        dc = (conds["color"][j] - torch.from_numpy(g["cond_color"][j])).abs().amax(0)
        dd = (conds["depth"][j] - torch.from_numpy(g["cond_depth"][j])).abs().amax(0)
        errs[f"cond{j + 1}_color_frac_within_1_255"] = float((dc < 2.1 / 255).float().mean())     # [-1,1] scale: 1/255 -> 2/255
        errs[f"cond{j + 1}_depth_frac_1e-3"] = float((dd < 1e-3).float().mean())
    G.report("chain/sample_all_vs_reference_sample_all", **errs)
    print("vs reference sample_all", errs)
    # measured: 1.2e-6 / 2.2e-6 / 2.3e-6 on the three views, every conditioning pixel within tolerance
    assert errs["view0"] < 1e-4, errs
    assert all(errs[k] > 0.999 for k in errs if k.startswith("cond")), errs
    assert errs["view1"] < 1e-3 and errs["view2"] < 1e-3, errs           # north_star's bar on the whole chain


@pytest.mark.parametrize("precision", ["fp32", "fp16s"])
def test_sample_all_scene_fixture_exercises_the_conditioning_against_the_references_sample_all(monkeypatch, precision):
    """tests/golden/sample_all_scene_ref.npz (make_golden_sample_all.py scene): the reference's own `sample_all` on inputs that
    make its generated views WELL-FORMED scenes -- `out.2` of both synthetic checkpoints scaled by 4e-6 and the three x_T draws
    replaced by sqrt(alpha_bar_T) * (smooth synthetic RGBD) + 4e-6 * (stream draw) -- so that the next views' conditioning has
    86-89 % mask / mask_rgb coverage (the generator asserts >= 50 % / 30 %): replace_rgb, replace_depth, mask_rgb and the
    convex-hull constraint (sample.py:99-120, ddim.py:86-95) all act end to end.  Compared: the three views, and per conditional
    view the colour / depth conditioning, mask, mask_rgb and depth_convex that aggregate_conditions handed to the sampler."""
    import gpu_util as G
    import warp_common as WC
    from ivid_amd import rgbd_3d
    from ivid_amd.diffusion import frameworks
    from ivid_amd.diffusion.backbones import AdmUnet2d
    from ivid_amd.inference.sample import sample_all
    from ivid_amd.rgbd_3d import camera
    g = C.load_golden("sample_all_scene_ref")

    def model(args, seed):
        sd = C.synth_weights(args, seed)
        sd["out.2.weight"] = sd["out.2.weight"] * float(g["out_scale"])
        sd["out.2.bias"] = sd["out.2.bias"] * float(g["out_scale"])
        m = AdmUnet2d(**args, precision=precision)     # fp32 and the headline 16-bit mode
        m.load_state_dict(sd)
        return m.cuda()
    fu = frameworks.ClassifierFreeGuidance(model(C.MINI128, 0), timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    fc = frameworks.InpaintCFG(model(C.MINI128_COND, 2), timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
    assert abs(float(fu.alphas_cumprod[-1]) ** 0.5 - float(g["xt_scale"])) < 1e-9
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(g["noise_seed"]))
    expected = [tuple(int(v) for v in str(d).split("x")) for d in g["draws"]]
    base = {int(d): float(g["xt_scale"]) * torch.from_numpy(WC.synthetic_rgbd(128, int(sd_), smooth_color=True))
            for d, sd_ in zip(g["xt_draws"], g["scene_seeds"])}
    drawn = []

    def noise_fn(shape):
        shape = tuple(int(v) for v in shape)
        assert len(drawn) < len(expected) and shape == expected[len(drawn)], (len(drawn), shape)
        t = torch.randn(shape, generator=gen)
        if len(drawn) in base:
            t = base[len(drawn)] + float(g["xt_noise"]) * t
        drawn.append(shape)
        return t.cuda()
    captured = []
    orig = rgbd_3d.WarpRenderer.conditions

    def conditions(self, *a, **kw):
        c = orig(self, *a, **kw)
        captured.append({k: getattr(c, k).detach().cpu() for k in ("color", "depth", "mask", "mask_rgb", "depth_convex")})
        return c
    monkeypatch.setattr(rgbd_3d.WarpRenderer, "conditions", conditions)
    vs = camera.viewset("3x9")
    views = [vs[int(k)] for k in g["view_ids"]]
    su, sc, erode = (int(v) for v in g["cfg"])
    out = list(sample_all(fu, fc, 1, su, sc, views, classes=[int(c) for c in g["classes"]], guidance=float(g["guidance"]),
                          batchsize=1, erode_rgb=erode, noise_fn=noise_fn))
    assert drawn == expected and len(captured) == 2
    samples = out[0][0].cpu()
    errs = {f"view{j}": C.rel_l2(samples[j], g["samples"][j]) for j in range(3)}
    nhwc = lambda t: t[0].permute(1, 2, 0).numpy()
    for j, c in enumerate(captured):
        m_ref, mr_ref = g["cond_mask"][j].astype(bool), g["cond_mask_rgb"][j].astype(bool)
        errs[f"cond{j + 1}_mask_coverage"] = float(m_ref.mean())
        errs[f"cond{j + 1}_mask_rgb_coverage"] = float(mr_ref.mean())
        errs[f"cond{j + 1}_mask_mismatch"] = int((nhwc(c["mask"]).astype(bool) != m_ref).sum())
        errs[f"cond{j + 1}_mask_rgb_mismatch"] = int((nhwc(c["mask_rgb"]).astype(bool) != mr_ref).sum())
        dc = np.abs(nhwc(c["color"]) * 2 - 1 - g["cond_color"][j].transpose(1, 2, 0)).max(-1)
        errs[f"cond{j + 1}_color_frac_within_1_255"] = float((dc < 2.1 / 255).mean())
        errs[f"cond{j + 1}_depth_frac_1e-3"] = float((np.abs(nhwc(c["depth"]) * 2 - 1 - g["cond_depth"][j].transpose(1, 2, 0)) < 1e-3).mean())
        errs[f"cond{j + 1}_depth_convex_frac_1e-3"] = float((np.abs(nhwc(c["depth_convex"]) - g["cond_depth_convex"][j]) < 5e-4).mean())
    G.report("chain/sample_all_scene_vs_reference_sample_all" + ("" if precision == "fp32" else "_" + precision), **errs)
    print("scene fixture vs reference sample_all", errs)
    for j in (1, 2):
        assert errs[f"cond{j}_mask_coverage"] >= 0.5 and errs[f"cond{j}_mask_rgb_coverage"] >= 0.3      # the fixture is not vacuous
        # a mask pixel may flip where a rasterised edge passes within float round-off of a sample point (real OpenGL vs the kernel)
        assert errs[f"cond{j}_mask_mismatch"] <= 8 and errs[f"cond{j}_mask_rgb_mismatch"] <= 8, errs
        assert errs[f"cond{j}_color_frac_within_1_255"] > 0.995 and errs[f"cond{j}_depth_frac_1e-3"] > 0.995, errs
        assert errs[f"cond{j}_depth_convex_frac_1e-3"] > 0.995, errs
    assert errs["view0"] < (1e-4 if precision == "fp32" else 1e-3) and errs["view1"] < 1e-3 and errs["view2"] < 1e-3, errs


def test_config4_rank_shard_at_full_size_batch_32_then_the_ragged_batch_of_2():
    """BASELINE config 4 (10 000 samples, 8 ranks, batches of 32): every rank ends on a ragged batch of 2 next to its 39 batches of
    32 (parallel.shard_plan; sample.py:56-58,199-202).  The first hardware run must not die on plumbing: at FULL model size (large
    cfg backbone, the headline precision mode) the two stacked-CFG plans of a rank -- bs 32 and bs 2 -- are built side by side,
    both run (announced low-t and high-t timesteps: every tier of the adaptive mode), a sample's forward does not depend on the
    batch it sits in (bitwise), and a plan budget that holds only one of the arenas evicts instead of failing."""
    from ivid_amd import parallel
    from ivid_amd.diffusion.backbones import AdmUnet2d
    sp = parallel.shard_plan(10000, 8, 32)
    assert all(r["samples"] == 1250 and r["batches"] == 40 and r["last_batch"] == 2 for r in sp)
    m = AdmUnet2d(**C.LARGE128, precision="fp16sa")
    m.load_state_dict(C.synth_weights(C.LARGE128, 4), strict=True)
    m = m.cuda().eval()
    x = C.seeded_randn(11, 32, 4, 128, 128).cuda()
    cls = (torch.arange(32) * 31 % 1000).cuda()
    for t in (20, 999):
        tt = torch.full((32,), t, dtype=torch.long).cuda()
        m.note_timestep(t)
        ec, eu = [v.clone() for v in m.forward_cfg(x, tt, cls)]
        m.note_timestep(t)
        rc, ru = m.forward_cfg(x[:2], tt[:2], cls[:2])                     # the ragged last batch: its own plan, same bits
        assert torch.isfinite(ec).all() and torch.equal(rc, ec[:2]) and torch.equal(ru, eu[:2]), t
    assert len(m._plans) == 4                                               # (32 | 2) x (tier 0 | tier 1), side by side
    big = max(p.arena.total_bytes() for p in m._plans.values())
    import warnings
    m.max_plan_bytes = int(0.5 * big) + sum(w.nbytes() for w in m._packed_tiers.values())     # neither bs-32 arena fits any more
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tt = torch.full((3,), 999, dtype=torch.long).cuda()
        m.note_timestep(999)
        e3, _ = m.forward_cfg(x[:3], tt, cls[:3])                           # a new shape under the tight budget: evicts, then runs
    assert torch.equal(e3[:2], ec[:2])                                      # (ec: the t = 999 pass of the loop above)
    assert (3, True, 1) in m._plans and not any(k[0] == 32 for k in m._plans), sorted(m._plans)
