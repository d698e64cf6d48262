"""Schedule tables shared by the samplers (reference: diffusion/samplers/utils.py:7-23 `extract`).

The reference gathers float64 table entries and rounds them to fp32 AFTER the gather
(`torch.from_numpy(arr)[t].float()`), then does all step math in fp32 torch.  Timesteps are
batch-uniform inside a sampler call (ddim.py:157-158), so here the gather happens on the host:
`f32(table, i)` is that rounded scalar, and derived coefficients are computed with numpy float32
arithmetic in the reference's operation order before being passed to the fused step kernel.
"""
import numpy as np
import torch


def f32(table, i):
    return np.float32(table[int(i)])


def uniform_timestep(t):
    """Host value of a batch-uniform timestep tensor / int (device tensors cost one sync)."""
    if isinstance(t, torch.Tensor):
        if t.numel() > 1 and not bool((t == t.flatten()[0]).all()):
            raise NotImplementedError("ivid_amd samplers require a batch-uniform timestep (as sample() issues)")
        return int(t.flatten()[0])
    return int(t)


def as_f32(t):
    return t.float().contiguous()
