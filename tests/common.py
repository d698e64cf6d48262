"""Shared test scaffolding: the mini model configs, seeded inputs, synthetic checkpoints, error metrics."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle.synth import synth_state_dict  # noqa: E402

# A 3-level, 32x32 ADM UNet that exercises every code path of the big ones in < 1 s on CPU:
# attention at T=256 and T=64, skip-conv ResBlocks (64->128), down/up ResBlocks, skip concat whose
# GroupNorm groups straddle the seam (192 = 128+64 channels -> 6-channel groups), class embedding
# with null class.
MINI = dict(image_size=32, in_channels=4, out_channels=4, model_channels=64, num_res_blocks=2,
            num_classes=10, has_null_class=True, channel_mult=[1, 2, 2], attention_resolutions=[16, 8],
            num_groups=32, num_heads=None, num_head_channels=64, dropout=0.0, use_fp16=False)
MINI_COND = dict(MINI, in_channels=10)
MINI_UNCLASS = dict(MINI, num_classes=None, has_null_class=False)

# backbone.args of the reference configs (configs/*.json) with use_fp16 forced False for fp32 parity
SMALL128 = dict(image_size=128, in_channels=4, out_channels=4, model_channels=128, num_res_blocks=2,
                num_classes=None, has_null_class=False, channel_mult=[1, 1, 2, 3, 4],
                attention_resolutions=[32, 16, 8], num_groups=32, num_heads=None, num_head_channels=64,
                dropout=0.0, use_fp16=False)
LARGE128 = dict(image_size=128, in_channels=4, out_channels=4, model_channels=256, num_res_blocks=2,
                num_classes=1000, has_null_class=True, channel_mult=[1, 1, 2, 3, 4],
                attention_resolutions=[32, 16, 8], num_groups=32, num_heads=None, num_head_channels=64,
                dropout=0.0, use_fp16=False)


# a class-conditional variant of the small-128 backbone (null class included): CFG chains at a CPU-affordable size
SMALL128_CFG = dict(SMALL128, num_classes=1000, has_null_class=True)

# rgbd_imagenet_adm_256_128_small_sr.json backbone (BASELINE config 5), fp32
SR256 = dict(image_size=256, in_channels=8, out_channels=4, model_channels=128, num_res_blocks=2, num_classes=1000,
             has_null_class=True, channel_mult=[1, 1, 2, 3, 4], attention_resolutions=[64, 32, 16], num_groups=32,
             num_heads=None, num_head_channels=64, dropout=0.0, use_fp16=False)
MINI_SR = dict(MINI, image_size=64, in_channels=8, attention_resolutions=[32, 16])
# the mini model at the reference sample_all's hard-wired 128 x 128 (inference/sample.py:50,68), attention at the 32^2 level
MINI128 = dict(MINI, image_size=128, attention_resolutions=[32])
MINI128_COND = dict(MINI128, in_channels=10)


def schema_for(args):
    from ivid_amd.diffusion.backbones.spec import build_spec
    return [(n, s) for n, s, _ in build_spec(**args).schema]


def synth_weights(args, seed=0):
    return synth_state_dict(schema_for(args), seed)


def seeded_randn(seed, *shape):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


# ---- the representative forward set (round 4): inputs a sampling chain actually feeds the network ----
FWD_SET_T = (0, 20, 250, 500, 750, 999)
# (tag, synthetic_rgbd kwargs, class): one band-limited scene and one layered scene with white-noise colours
FWD_SET_SCENES = (("smooth", dict(seed=0, smooth_color=True, layers=False), 7),
                  ("layers", dict(seed=1, smooth_color=False, layers=True), 416))


def fwd_set_inputs(in_channels=4, S=128):
    """[(key, x_t [1,C,S,S] fp32, t, class)] with x_t = sqrt(abar_t) * x0 + sqrt(1 - abar_t) * n (gaussian_diffusion.py:45-56,
    linear betas, 1000 timesteps), x0 = tests/warp_common.synthetic_rgbd in [-1, 1], n = seeded_randn(7000 + 100*scene + index of t).
    The same recipe runs in tests/golden/make_golden_fwd_set.py (live reference) and on the GPU box."""
    import warp_common as WC
    betas = np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)
    abar = np.cumprod(1.0 - betas)
    out = []
    for si, (tag, kw, cls) in enumerate(FWD_SET_SCENES):
        x0 = torch.from_numpy(WC.synthetic_rgbd(S, **kw)).float()
        if in_channels > 4:   # conditional / SR models: the remaining channels carry the clean scene (y, mask-like planes)
            x0 = torch.cat([x0, x0[:, : in_channels - 4]], 1)
        for ti, t in enumerate(FWD_SET_T):
            n = seeded_randn(7000 + 100 * si + ti, 1, in_channels, S, S)
            x = float(np.sqrt(abar[t])) * x0 + float(np.sqrt(1.0 - abar[t])) * n
            if in_channels > 4:
                x[:, 4:] = x0[:, 4:]
            out.append((f"{tag}_t{t}", x.contiguous(), t, cls))
    return out
