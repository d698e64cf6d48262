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


_CANON = None


def equivalent_timestep(framework, t_model):
    """The adaptive precision modes' tier thresholds are timesteps of the schedule every reference config uses (1000 steps, linear
    betas: configs/*.json `framework.args`).  A framework with another schedule announces the canonical timestep of the SAME noise
    level instead of its own index: the largest canonical t whose signal share abar_t is not below this step's (so the tier picked
    is at least as accurate as the one the canonical schedule would pick for an input this clean).  Identity for the canonical
    schedule; frameworks without a `betas` table are passed through."""
    global _CANON
    betas = getattr(framework, "betas", None)
    if betas is None or t_model is None:
        return t_model
    if _CANON is None:
        cb = np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)
        _CANON = (cb, np.cumprod(1.0 - cb))
    b = np.asarray(betas, dtype=np.float64)
    if b.shape == _CANON[0].shape and np.allclose(b, _CANON[0], rtol=1e-12, atol=0.0):
        return int(t_model)
    abar = float(np.cumprod(1.0 - b)[int(t_model)])
    n_cleaner_or_equal = int(np.searchsorted(-_CANON[1], -abar, side="right"))      # canonical steps with abar_t >= abar
    return max(0, n_cleaner_or_equal - 1)


def announce_timestep(framework, t_model):
    """Tell the backbone the host-side value of the timestep tensor the next `model_inference` call will carry (the samplers build
    that tensor from a Python int; the backbone would have to synchronise to read it back): AdmUnet2d.note_timestep, which only the
    adaptive precision modes look at -- as the equivalent timestep of the canonical 1000-step linear schedule (equivalent_timestep).
    None withdraws the announcement.  Other backbones are left alone."""
    bb = getattr(framework, "backbone", None)
    bb = getattr(bb, "module", bb)
    if hasattr(bb, "note_timestep"):
        bb.note_timestep(equivalent_timestep(framework, t_model))


def framework_eps(framework, x_t, t_model, classes, kwargs):
    """-> (eps_cond, eps_uncond | None, strength) for the fused step kernels.  Frameworks of this package expose `eps_branches`
    (both guidance branches out of ONE stacked forward, combined inside the step kernel).  Any other object with the
    reference's framework contract -- `model_inference(x, t, classes=..., **kwargs) -> eps` (gaussian_diffusion.py:56-70,
    classifier_free_guidance.py:23-42) -- is called exactly as the reference's samplers call it (ddim.py:81, ddpm.py:101) and
    its already combined eps passes through the kernel unchanged."""
    if hasattr(framework, "eps_branches"):
        return framework.eps_branches(x_t, t_model, classes=classes, **kwargs)
    kw = {k: v for k, v in kwargs.items() if k != "noise_fn"}      # noise_fn is this package's sampler option, not the framework's
    return framework.model_inference(x_t, t_model, classes=classes, **kw), None, 0.0
