"""CPU restatement of ivid's frameworks + samplers — TEST INFRASTRUCTURE (the oracle).

Follows /root/reference/diffusion/frameworks/{utils,gaussian_diffusion,classifier_free_guidance,
inpaint_cfg}.py and /root/reference/diffusion/samplers/{ddim,ddpm,utils}.py.  `eps_model(x, t,
classes)` is any callable returning the UNet's epsilon (the UNet oracle, or the live reference).
Noise comes from torch's CPU generator in the reference's draw order, so seeding
`torch.manual_seed(s)` before a call reproduces the reference's noise stream exactly.
"""
import numpy as np
import torch


def linear_betas(T):
    """frameworks/utils.py:22-28."""
    s = 1000 / T
    return np.linspace(s * 0.0001, s * 0.02, T, dtype=np.float64)


def _x(table, idx, like):
    """samplers/utils.py:20 — float64 table gathered, THEN rounded to fp32, broadcast over the batch."""
    return torch.from_numpy(np.asarray(table))[idx].float().view(-1, *([1] * (like.dim() - 1)))


def cfg_eps(eps_model, x, t, classes, strength):
    """classifier_free_guidance.py:39-42."""
    e = (1 + strength) * eps_model(x, t, classes)
    if strength > 0:
        e = e - strength * eps_model(x, t, None)
    return e


def inpaint_inputs(x, y, mask, mask_rgb=None):
    """inpaint_cfg.py:24-49 (noise: rgb first, then depth)."""
    parts = [x]
    mr = mask
    if mask_rgb is not None:
        parts.append(mask_rgb)
        mr = mask_rgb
    y_rgb, y_d = y[:, :3], y[:, 3:]
    parts.append(y_rgb * mr + torch.randn_like(y_rgb) * (1 - mr))
    parts.append(y_d * mask + torch.randn_like(y_d) * (1 - mask))
    parts.append(mask)
    return torch.cat(parts, dim=1)


def inpaint_cfg_eps(eps_model, x, t, y, mask, classes, strength, mask_rgb=None):
    """inpaint_cfg.py:60-83."""
    ci = inpaint_inputs(x, y, mask, mask_rgb)
    if classes is None:
        return eps_model(ci, t, None)
    e = (1 + strength) * eps_model(ci, t, classes)
    if strength > 0:
        e = e - strength * eps_model(ci, t, None)
    return e


@torch.no_grad()
def ddim_sample(eps_fn, x_T, steps, betas, eta=0.0, clip_denoised=False, replace_rgb=None, replace_depth=None,
                constrain_depth=None):
    """ddim.py:47-165.  eps_fn(x_t, t_index[B]) -> eps, called with t-1 like ddim.py:81."""
    T = len(betas)
    ac = np.cumprod(1.0 - betas)
    ac_prev = np.append(1.0, ac[:-1])
    sr, srm1 = np.sqrt(1.0 / ac), np.sqrt(1.0 / ac - 1)
    B = x_T.shape[0]
    img = x_T
    jump = T // steps
    out = {"pred_x_t": [], "pred_x_0": []}
    for i in reversed(range(steps)):
        t = torch.full((B,), jump * (i + 1), dtype=torch.long)
        tp = torch.full((B,), jump * i, dtype=torch.long)
        eps = eps_fn(img, t - 1)
        x0 = _x(sr, t - 1, img) * img - _x(srm1, t - 1, img) * eps
        nz = (tp != 0).float().view(-1, 1, 1, 1)
        if clip_denoised:
            x0 = x0.clamp(-1.0, 1.0)
        if replace_rgb is not None:
            w, rgb, m = replace_rgb
            x0[:, :3] = (1 - nz) * x0[:, :3] + nz * ((w * rgb + (1 - w) * x0[:, :3]) * m + x0[:, :3] * (1 - m))
        if replace_depth:
            w, d, m = replace_depth
            x0[:, 3:] = (w * d + (1 - w) * x0[:, 3:]) * m + x0[:, 3:] * (1 - m)
            if constrain_depth:
                cw, convex = constrain_depth
                x0[:, 3:] = x0[:, 3:] * m + (cw * torch.maximum(x0[:, 3:], convex) + (1 - cw) * x0[:, 3:]) * (1 - m)
        eps2 = (_x(sr, t - 1, img) * img - x0) / _x(srm1, t - 1, img)
        ab, abp = _x(ac, t - 1, img), _x(ac_prev, tp, img)
        sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
        mean = torch.sqrt(abp) * x0 + torch.sqrt(1 - abp - sigma ** 2) * eps2
        noise = torch.randn_like(img)
        img = mean + nz * sigma * noise
        out["pred_x_t"].append(img)
        out["pred_x_0"].append(x0)
    out["samples"] = img
    return out


@torch.no_grad()
def ddpm_sample(eps_fn, x_T, betas, clip_denoised=False, t_start=None, t_stop=0):
    """ddpm.py:43-187 (fixed-small variance, clipped log-variance).  `t_start` / `t_stop` restrict the
    chain to steps t_start ... t_stop (default T-1 ... 0) for single-step tests."""
    T = len(betas)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas)
    ac_prev = np.append(1.0, ac[:-1])
    sr, srm1 = np.sqrt(1.0 / ac), np.sqrt(1.0 / ac - 1)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    plv = np.log(np.append(pv[1], pv[1:]))
    c1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
    c2 = (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)
    B = x_T.shape[0]
    img = x_T
    out = {"pred_x_t": [], "pred_x_0": []}
    for i in range(T - 1 if t_start is None else t_start, t_stop - 1, -1):
        t = torch.full((B,), i, dtype=torch.long)
        eps = eps_fn(img, t)
        x0 = _x(sr, t, img) * img - _x(srm1, t, img) * eps
        if clip_denoised:
            x0 = x0.clamp(-1, 1)
        mean = _x(c1, t, img) * x0 + _x(c2, t, img) * img
        noise = torch.randn_like(img)
        nz = (t != 0).float().view(-1, 1, 1, 1)
        img = mean + nz * torch.exp(0.5 * _x(plv, t, img)) * noise
        out["pred_x_t"].append(img)
        out["pred_x_0"].append(x0)
    out["samples"] = img
    return out
