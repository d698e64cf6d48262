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


def synth_state_dict(schema, seed=0):
    """schema: iterable of (name, shape) in state_dict order.  Returns {name: fp32 tensor}."""
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
    return sd


def schema_of(module_or_sd):
    sd = module_or_sd if isinstance(module_or_sd, dict) else module_or_sd.state_dict()
    return [(k, tuple(v.shape)) for k, v in sd.items()]
