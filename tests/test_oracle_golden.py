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
