"""CPU: the oracle restatement reproduces the golden fixtures generated from the live reference."""
import numpy as np
import pytest
import torch

import common as C
from oracle import adm_oracle, sampler_oracle


def _fwd(name, args, seed, batch):
    g = C.load_golden(name)
    S = args["image_size"]
    x = C.seeded_randn(100 + seed, batch, args["in_channels"], S, S)
    assert abs(float(x.double().sum()) - float(g["x_checksum"])) < 1e-6, "torch CPU RNG stream changed"
    if "x" in g:
        assert np.array_equal(g["x"], x.numpy())
    t = torch.full((batch,), int(g["t"]), dtype=torch.long)
    cls = torch.from_numpy(g["classes"]) if "classes" in g else None
    sd = C.synth_weights(args, seed)
    out = adm_oracle.unet_forward(sd, args, x, t, cls)
    assert C.rel_l2(out, g["eps"]) < 1e-5
    if "eps_uncond" in g:
        assert C.rel_l2(adm_oracle.unet_forward(sd, args, x, t, None), g["eps_uncond"]) < 1e-5


@pytest.mark.parametrize("name,args,seed,batch", [
    ("mini_fwd", C.MINI, 0, 2),
    ("mini_unclass_fwd", C.MINI_UNCLASS, 1, 2),
    ("mini_cond_fwd", C.MINI_COND, 2, 2),
    ("small128_fwd", C.SMALL128, 3, 1),
])
def test_unet_oracle_matches_reference_golden(name, args, seed, batch):
    _fwd(name, args, seed, batch)


def test_large_unet_oracle_matches_reference_golden():
    _fwd("large128_fwd", C.LARGE128, 4, 1)


def test_fresh_reference_style_model_is_not_all_zero():
    # the synthetic checkpoint must re-randomise the zero_module()'d tensors
    sd = C.synth_weights(C.MINI, 0)
    for k in ("out.2.weight", "middle_block.1.proj_out.weight", "input_blocks.1.0.out_layers.3.weight"):
        assert float(sd[k].abs().max()) > 0


def test_ddim_cfg_chain_matches_reference_golden():
    g = C.load_golden("mini_ddim_cfg")
    sd = C.synth_weights(C.MINI, 0)
    cls = torch.from_numpy(g["classes"])
    um = lambda a, b, c: adm_oracle.unet_forward(sd, C.MINI, a, b, c)
    torch.manual_seed(5)
    out = sampler_oracle.ddim_sample(lambda x, t: sampler_oracle.cfg_eps(um, x, t, cls, 0.5), torch.from_numpy(g["x_T"]),
                                     5, sampler_oracle.linear_betas(1000), eta=0.5)
    assert C.rel_l2(out["samples"], g["samples"]) < 1e-5
    assert C.rel_l2(out["pred_x_0"][0], g["x0_first"]) < 1e-5


def test_ddim_inpaint_chain_matches_reference_golden():
    g = C.load_golden("mini_ddim_inpaint")
    sd = C.synth_weights(C.MINI_COND, 2)
    T = torch.from_numpy
    y, mask, mask_rgb, convex, cls = T(g["y"]), T(g["mask"]), T(g["mask_rgb"]), T(g["convex"]), T(g["classes"])
    um = lambda a, b, c: adm_oracle.unet_forward(sd, C.MINI_COND, a, b, c)
    torch.manual_seed(7)
    out = sampler_oracle.ddim_sample(
        lambda x, t: sampler_oracle.inpaint_cfg_eps(um, x, t, y, mask, cls, 3.0, mask_rgb), T(g["x_T"]), 4,
        sampler_oracle.linear_betas(1000), replace_rgb=(0.1, y[:, :3], mask_rgb), replace_depth=(0.2, y[:, 3:], mask),
        constrain_depth=(0.5, convex))
    assert C.rel_l2(out["samples"], g["samples"]) < 1e-5
    assert C.rel_l2(out["pred_x_0"][-1], g["x0_last"]) < 1e-5


def test_ddpm_chain_matches_reference_golden():
    g = C.load_golden("mini_ddpm")
    sd = C.synth_weights(C.MINI_UNCLASS, 1)
    torch.manual_seed(9)
    out = sampler_oracle.ddpm_sample(lambda x, t: adm_oracle.unet_forward(sd, C.MINI_UNCLASS, x, t, None),
                                     torch.from_numpy(g["x_T"]), sampler_oracle.linear_betas(100))
    assert C.rel_l2(out["samples"], g["samples"]) < 1e-5


def test_oracle_chain_matches_the_references_own_sample_all():
    """tests/golden/sample_all_ref.npz: the reference's `sample_all` itself (inference/sample.py:30-147), run end to end in the
    build container incl. its renderer on real OpenGL (tests/golden/make_golden_sample_all.py).  The oracle's pieces --
    UNet, CFG / InpaintCFG, DDIM, depth_to_mesh, C rasteriser + aggregation + resolve -- chained with the reference's wiring
    (sample.py:99-120) and fed the same seeded noise stream (the draw order of the reference) must reproduce its three views
    and its conditioning tensors.  The product is checked against the same file on the GPU (tests/test_pipeline_gpu.py)."""
    import warp_common as WC
    from oracle import warp_oracle as W
    g = C.load_golden("sample_all_ref")
    S, near, far, atol, rtol = 128, 0.6, 5.0, 0.03, 0.03
    su, sc, erode = (int(v) for v in g["cfg"])
    guid, cls = float(g["guidance"]), torch.from_numpy(g["classes"])
    vs = WC.viewset_3x9()
    views = [WC.orbit(*vs[int(k)]) for k in g["view_ids"]]
    sdu, sdc = C.synth_weights(C.MINI128, 0), C.synth_weights(C.MINI128_COND, 2)
    uu = lambda x, t, c: adm_oracle.unet_forward(sdu, C.MINI128, x, t, c)
    uc = lambda x, t, c: adm_oracle.unet_forward(sdc, C.MINI128_COND, x, t, c)
    betas = sampler_oracle.linear_betas(1000)
    torch.manual_seed(int(g["noise_seed"]))        # the default generator now yields the stream the reference drew from
    x_T = torch.randn(1, 4, S, S)
    prev = [sampler_oracle.ddim_sample(lambda x, t: sampler_oracle.cfg_eps(uu, x, t, cls, guid), x_T, su, betas)["samples"]]
    assert C.rel_l2(prev[0][0], g["samples"][0]) < 1e-5
    for j in (1, 2):
        meshes, cols = [], []
        for k in range(j):
            hw = prev[k][0].numpy().transpose(1, 2, 0) * 0.5 + 0.5                       # sample.py:83
            meshes.append(W.depth_to_mesh(W.linearize_depth(hw[:, :, 3:], near, far), 45, views[k], atol, rtol, erode))
            cols.append(np.ascontiguousarray(hw[:, :, :3]))
        c = W.aggregate_conditions(meshes, cols, views[j], S, 3, 45, near, far, atol, rtol, erode)
        T = lambda k: torch.from_numpy(np.asarray(c[k], np.float32)).permute(2, 0, 1)[None]
        color, depth, mask, mask_rgb, convex = T("color") * 2 - 1, T("depth") * 2 - 1, T("mask"), T("mask_rgb"), T("depth_convex") * 2 - 1
        dc = (color[0] - torch.from_numpy(g["cond_color"][j - 1])).abs().amax(0)
        dd = (depth[0] - torch.from_numpy(g["cond_depth"][j - 1])).abs().amax(0)
        assert (dc < 2.1 / 255).float().mean() > 0.999 and (dd < 1e-3).float().mean() > 0.999, (j, float((dc < 2.1 / 255).float().mean()))
        y = torch.cat([color, depth], dim=1)
        x2 = torch.randn(1, 4, S, S)                                                     # ddim.py:151: the conditional chain's x_T
        res = sampler_oracle.ddim_sample(lambda x, t: sampler_oracle.inpaint_cfg_eps(uc, x, t, y, mask, cls, guid, mask_rgb), x2, sc, betas,
                                         replace_rgb=(0.1, color, mask_rgb), replace_depth=(0.2, depth, mask), constrain_depth=(0.5, convex))
        prev.append(res["samples"])
        assert C.rel_l2(prev[j][0], g["samples"][j]) < 1e-3, (j, C.rel_l2(prev[j][0], g["samples"][j]))


def test_oracle_reproduces_the_representative_forward_set_small_and_large_rows():
    """tests/golden/*_fwd_set.npz (round 4, make_golden_fwd_set.py): the seeded recipe rebuilds the generator's inputs and the
    oracle reproduces the live reference's outputs -- all 12 rows of the small set, the hardest and an easy row of the large one
    on both guidance branches (the generator checked every one of the 36 on the spot: manifest.json, rel-L2 0.0)."""
    import json
    import os
    man = json.load(open(os.path.join(C.GOLDEN, "manifest.json")))
    assert man["large128_fwd_set"]["oracle_vs_reference"]["rel_l2_max"] == 0.0
    assert man["small128_fwd_set"]["oracle_vs_reference"]["rel_l2_max"] == 0.0
    for gname, args, seed, keep in (("small128_fwd_set", C.SMALL128, 3, None), ("large128_fwd_set", C.LARGE128, 4, ("smooth_t20", "layers_t999"))):
        g = C.load_golden(gname)
        sd = C.synth_weights(args, seed)
        ins = C.fwd_set_inputs(args["in_channels"], args["image_size"])
        assert len(ins) == 12 and len({k for k, *_ in ins}) == 12
        for key, x, t, cls in ins:
            assert abs(float(x.double().sum()) - float(g[key + "_xsum"])) < 1e-3 * max(1.0, abs(float(g[key + "_xsum"]))), key
            if keep is not None and key not in keep:
                continue
            tt = torch.tensor([t])
            if args["num_classes"] is not None:
                assert C.rel_l2(adm_oracle.unet_forward(sd, args, x, tt, torch.tensor([cls])), g[key + "_c"]) < 1e-5, key
            assert C.rel_l2(adm_oracle.unet_forward(sd, args, x, tt, None), g[key + "_u"]) < 1e-5, key


def test_oracle_reproduces_the_mid_t_forward_sets():
    """tests/golden/{large,small}128_fwd_set_mid.npz (make_golden_fwd_set.py largemid smallmid): the rows between the low-noise
    timesteps of the main sets (t = 50, 100, 150, 350), where the adaptive precision mode switches plans -- recipe and oracle
    against the live reference's outputs (all 8 rows of the small set, the t = 50 smooth row of the large one, the t = 250 smooth row
    of the conditional model's set -- its 10-channel input re-drawn by the recipe of InpaintCFG.make_cond_inputs)."""
    import json
    import os
    man = json.load(open(os.path.join(C.GOLDEN, "manifest.json")))
    assert man["sr256_fwd_set_mid"]["oracle_vs_reference"]["rel_l2_max"] == 0.0      # checked by the generator (a 256^2 forward: 15 s each)
    for gname, args, seed, keep in (("small128_fwd_set_mid", C.SMALL128, 3, None), ("large128_fwd_set_mid", C.LARGE128, 4, ("smooth_t50",)),
                                    ("largecond128_fwd_set_mid", C.LARGE128_COND, 2, ("smooth_t250",))):
        assert man[gname]["oracle_vs_reference"]["rel_l2_max"] == 0.0
        g = C.load_golden(gname)
        sd = C.synth_weights(args, seed)
        ins = C.FWD_SETS[gname.replace("_fwd_set", "")][3]()
        assert [t for _, _, t, _ in ins] == list(C.FWD_SET_T_MID_MORE if "cond" in gname else C.FWD_SET_T_MID) * 2
        for key, x, t, cls in ins:
            assert abs(float(x.double().sum()) - float(g[key + "_xsum"])) < 1e-3 * max(1.0, abs(float(g[key + "_xsum"]))), key
            if keep is not None and key not in keep:
                continue
            tt = torch.tensor([t])
            if args["num_classes"] is not None:
                assert C.rel_l2(adm_oracle.unet_forward(sd, args, x, tt, torch.tensor([cls])), g[key + "_c"]) < 1e-5, key
            assert C.rel_l2(adm_oracle.unet_forward(sd, args, x, tt, None), g[key + "_u"]) < 1e-5, key


def test_teacher_forced_steps_golden_is_consistent_with_the_chain_golden():
    """large128_ddim50_cfg_steps.npz records what the reference's framework saw at sample_once calls 1, 10, 25, 49 of the config-2
    chain: the timesteps must be those of the 50-step DDIM schedule (ddim.py:157: t - 1 of (1000 - 20 k)) and the oracle's guided
    eps on the recorded input of step 49 must reproduce the recorded answer."""
    g = C.load_golden("large128_ddim50_cfg_steps")
    for k in (1, 10, 25, 49):
        assert int(g[f"t_step{k}"]) == 1000 - 20 * k - 1
        assert g[f"x_step{k}"].shape == g[f"eps_step{k}"].shape == (2, 4, 128, 128)
    sd = C.synth_weights(C.LARGE128, 4)
    cls = torch.from_numpy(g["classes"])
    x, t = torch.from_numpy(g["x_step49"])[:1], torch.tensor([int(g["t_step49"])])
    um = lambda a, b, c: adm_oracle.unet_forward(sd, C.LARGE128, a, b, c)
    eps = sampler_oracle.cfg_eps(um, x, t, cls[:1], 0.5)
    assert C.rel_l2(eps, g["eps_step49"][:1]) < 1e-5


def test_philox_oracle_reproduces_the_published_known_answer_vectors():
    """oracle/philox_oracle.py (the checker of ivid_randn) against the Random123 known-answer vectors of Philox4x32-10
    (Salmon et al., SC'11, kat_vectors): zero, all-ones and the pi-digits counter / key."""
    import numpy as np
    from oracle import philox_oracle as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = P.philox4x32_10(np.array(ctr, dtype=np.uint32), key)
        assert tuple(int(v) for v in got) == want
    z = P.randn(1234, 5, 1 << 18).astype(np.float64)
    assert abs(z.mean()) < 6e-3 and abs(z.var() - 1) < 1e-2 and abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3) < 5e-2
    assert not np.array_equal(P.randn(1234, 5, 64), P.randn(1234, 6, 64)) and not np.array_equal(P.randn(1234, 5, 64), P.randn(1235, 5, 64))
