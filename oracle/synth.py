"""Deterministic synthetic checkpoints (TEST INFRASTRUCTURE).

No ivid checkpoint is available offline and a freshly constructed reference model outputs exactly
0 (every ResBlock out-conv, every proj_out and the final conv are zero_module()'d —
/root/reference/diffusion/backbones/adm.py:182,278,486).  `synth_state_dict`
fills EVERY tensor of a state_dict schema with well-scaled seeded values, so the same recipe gives
bit-identical weights here (golden generation against the live reference) and on the GPU box.
The result is a bare state_dict in the reference checkpoint format (inference/sample.py:186-187).
"""
import math

import torch


def synth_state_dict(schema, seed=0, variant=None):
    """schema: iterable of (name, shape) in state_dict order.  Returns {name: fp32 tensor}.
    variant "trained" (round 5: is a tolerance claim seed luck?): the same draw, then the statistics a TRAINED ADM checkpoint
    differs by from a fresh one -- GroupNorm gains far from 1 (gamma ~ U(0.2, 3), beta ~ U(-0.5, 0.5)) and FiLM projections four
    times larger (emb_layers gain 1.2 instead of 0.3: (1 + scale) swings between ~ -1 and 3) -- drawn from a second generator
    so that variant=None is bit-identical to what every earlier fixture was made from."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd = {}
    for name, shape in schema:
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "freqs":  # PosEncoding buffer, adm.py:28
            half = shape[0]
            sd[name] = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
            continue
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        is_norm = any(s in name for s in (".in_layers.0.", ".out_layers.0.", ".norm.")) or name.startswith("out.0.")
        if is_norm:
            sd[name] = (1.0 + 0.1 * n) if leaf == "weight" else 0.1 * n
        elif name == "label_emb.weight":
            sd[name] = 0.5 * n
        elif len(shape) == 1:
            sd[name] = 0.05 * n
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 0.3 if ".emb_layers." in name else 1.0  # keep FiLM scale/shift moderate
            sd[name] = n * (gain / math.sqrt(fan_in))
    if variant == "trained":
        g2 = torch.Generator(device="cpu")
        g2.manual_seed(1000003 + seed)
        for name, shape in schema:
            leaf = name.rsplit(".", 1)[-1]
            is_norm = any(s in name for s in (".in_layers.0.", ".out_layers.0.", ".norm.")) or name.startswith("out.0.")
            if is_norm:
                u = torch.rand(tuple(shape), generator=g2, dtype=torch.float32)
                sd[name] = (0.2 + 2.8 * u) if leaf == "weight" else (u - 0.5)
            elif ".emb_layers." in name and leaf == "weight":
                sd[name] = sd[name] * 4.0
    elif variant is not None:
        raise ValueError(f"unknown synthetic-checkpoint variant {variant!r}")
    return sd


def schema_of(module_or_sd):
    sd = module_or_sd if isinstance(module_or_sd, dict) else module_or_sd.state_dict()
    return [(k, tuple(v.shape)) for k, v in sd.items()]
