"""The whole sampling loop as ONE C call (`ivid_sample`, include/ivid_hip.h): what `DdimSampler.sample` /
`DdpmSampler.sample` do step by step from Python (reference: diffusion/samplers/ddim.py:150-163, ddpm.py:172-185) — build the
model input, run the UNet program of the step's precision tier, apply the fused update — enqueued by the library in one go, with
no host code between the steps.  Opt-in through the samplers' `device_loop=True`; results are bit-identical to the host-driven loop
(same programs, same kernels, same order, same noise draws: tests/test_sample_loop_gpu.py).

What it covers is what the sampling drivers issue (inference/sample.py:75-139): GaussianDiffusion / ClassifierFreeGuidance on x_t
itself, InpaintCFG with `replace_rgb` / `replace_depth` / `constrain_depth`, SuperResCFG on a low-resolution `y`; anything else (a foreign framework or backbone, a
guidance strength <= 0 with classes, further framework arguments) raises NotImplementedError — the caller keeps the host loop
for those.  Intermediates are not kept: `pred_x_t` comes back empty and `pred_x_0` holds the last step's prediction.
`device_noise_seed=<int>`: every draw of the loop (step noise, InpaintCFG's hole noise) comes from the library's counter-based
generator (`ivid_randn`: stream 2 * step [+ 1] under that seed) instead of torch's -- no noise buffers, the same chain whatever the
loop is cut into; x_T is still the caller's.
"""
import ctypes as C

import torch

from ... import _lib
from ...utils import AttrDict, default_noise
from .utils import as_f32, equivalent_timestep


def _backbone_of(framework):
    bb = getattr(framework, "backbone", None)
    bb = getattr(bb, "module", bb)
    if not (hasattr(bb, "plan") and hasattr(bb, "tier_of") and hasattr(bb, "forward_cfg")):
        raise NotImplementedError("device_loop needs an ivid_amd AdmUnet2d backbone")
    return bb


def run(sampler, kind, img, steps, classes, kwargs, chunk=64):
    """steps: list of (t_model, coef_factory(strength, w_rgb, w_depth, w_con) -> ctypes coefficient struct, draws_noise) in sampling
    order.  Returns AttrDict(samples, pred_x_t=[], pred_x_0=[last pred_x_0])."""
    from ..frameworks import ClassifierFreeGuidance, GaussianDiffusion, InpaintCFG, SuperResCFG
    fw = sampler.framework
    bb = _backbone_of(fw)
    kwargs = dict(kwargs)
    injected = kwargs.pop("noise_fn", None)
    dev_seed = kwargs.pop("device_noise_seed", None)
    chunk = int(kwargs.pop("device_loop_chunk", chunk))
    if dev_seed is not None and injected is not None:
        raise ValueError("device_loop: noise_fn and device_noise_seed exclude each other")
    noise_fn = injected or (lambda shape: default_noise(shape, img.device))
    strength = kwargs.pop("strength", 3.0 if type(fw) in (ClassifierFreeGuidance, InpaintCFG, SuperResCFG) else 0.0)
    inpaint = type(fw) is InpaintCFG
    if type(fw) not in (GaussianDiffusion, ClassifierFreeGuidance, InpaintCFG, SuperResCFG):
        raise NotImplementedError(f"device_loop: framework {type(fw).__name__} is driven from the host")
    x = as_f32(img).clone()
    b, c, h, w = x.shape
    if c != 4:
        raise NotImplementedError("device_loop: RGBD (4-channel) samples only")
    hw = h * w
    # which program a step runs: the frameworks' own rules (gaussian_diffusion.py cfg_branches)
    if type(fw) is GaussianDiffusion:
        stacked, strength = False, 0.0
    elif classes is None:
        stacked, strength = False, 0.0
    elif strength > 0:
        stacked = True
    else:
        raise NotImplementedError("device_loop: guidance strength <= 0 with classes is driven from the host")
    cond = _lib.SampleCond()
    keepalive = []

    def dev(t, ch):
        t = as_f32(t.to(x.device).expand(b, ch, h, w))
        keepalive.append(t)
        return t.data_ptr()
    w_rgb = w_dep = w_con = -1.0
    if inpaint:
        cond.y, cond.mask = dev(kwargs.pop("y"), 4), dev(kwargs.pop("mask"), 1)
        mr = kwargs.pop("mask_rgb", None)
        cond.mask_rgb = dev(mr, 1) if mr is not None else None
    if type(fw) is SuperResCFG:
        y = as_f32(kwargs.pop("y").to(x.device))
        y = as_f32(y.expand(b, -1, -1, -1))
        keepalive.append(y)
        cond.sr_y, cond.sr_channels, cond.sr_size = y.data_ptr(), y.shape[1], y.shape[-1]
    if kind == _lib.SAMPLE_DDIM:
        rr, rd, cd = kwargs.pop("replace_rgb", None), kwargs.pop("replace_depth", None), kwargs.pop("constrain_depth", None)
        if rr is not None:                   # `is not None` / truthiness exactly as DdimSampler.sample_once (ddim.py:86-95)
            w_rgb, cond.rgb, cond.rgb_mask = float(rr[0]), dev(rr[1], 3), dev(rr[2], 1)
        if rd:
            w_dep, cond.depth, cond.depth_mask = float(rd[0]), dev(rd[1], 1), dev(rd[2], 1)
            if cd:
                w_con, cond.convex = float(cd[0]), dev(cd[1], 1)
    if kwargs:
        raise NotImplementedError(f"device_loop: unsupported sampler arguments {sorted(kwargs)}")
    if classes is not None:
        bb._check_labels(classes)
        classes = classes.to(device=x.device, dtype=torch.int64).contiguous()
    # programs of the tiers the schedule visits
    tiers = [bb.tier_of(equivalent_timestep(fw, t), strength if stacked else None) for (t, _, _) in steps]
    order = sorted(set(tiers))
    plans = [bb.plan(b, stacked, k) for k in order]
    for p in plans:
        if p.program is None:
            raise NotImplementedError("device_loop needs C-side launch programs (IVID_PROGRAM=0 disables them)")
    handles = (C.c_void_p * len(plans))(*[getattr(p.program, "value", p.program) for p in plans])
    # the loop runs on the first plan's own stream (a program captures its hipGraph on the stream it is launched on, which must not
    # be the legacy default stream), ordered behind the caller's stream and every plan's earlier work
    cur = torch.cuda.current_stream(x.device)
    ls = plans[0].stream
    x0 = torch.empty_like(x)
    ls.wait_stream(cur)
    for p in plans[1:]:
        ls.wait_stream(p.stream)

    def on_loop_stream(*tensors):       # caching-allocator bookkeeping: these blocks are in use on `ls`
        for tns in tensors:
            if tns is not None:
                tns.record_stream(ls)
    on_loop_stream(x, x0, classes, *keepalive)
    coef_t = _lib.DdimCoef if kind == _lib.SAMPLE_DDIM else _lib.DdpmCoef
    for s0 in range(0, len(steps), chunk):
        part = steps[s0:s0 + chunk]
        n = len(part)
        coefs = (coef_t * n)(*[f(strength if stacked else 0.0, w_rgb, w_dep, w_con) for (_, f, _) in part])
        tm = (C.c_longlong * n)(*[int(t) for (t, _, _) in part])
        eng = (C.c_int * n)(*[order.index(k) for k in tiers[s0:s0 + n]])
        # noise in the host loop's draw order: per step [hole rgb, hole depth,] step noise (drawn whenever the host loop draws it)
        # (the kernel takes a step's hole noise as two dense tensors: rgb [B,3,HW], then depth [B,1,HW])
        gen = dev_seed is not None
        hole = torch.empty(n, b * 4 * hw, dtype=torch.float32, device=x.device) if (inpaint and not gen) else None
        any_noise = any(d for (_, _, d) in part) and not gen
        stepn = torch.empty(n, b, 4, h, w, dtype=torch.float32, device=x.device) if any_noise else None
        for i, (_, _, draws) in enumerate(part if not gen else ()):
            if inpaint:
                hole[i, :b * 3 * hw].view(b, 3, h, w).copy_(noise_fn((b, 3, h, w)))
                hole[i, b * 3 * hw:].view(b, 1, h, w).copy_(noise_fn((b, 1, h, w)))
            if draws or injected:
                z = noise_fn((b, 4, h, w))
                if draws:
                    stepn[i] = z
        if inpaint:
            cond.hole_noise = hole.data_ptr() if hole is not None else None
        plan = _lib.SamplePlan(kind, n, hw, tm, C.cast(coefs, C.c_void_p), eng, int(gen), s0, int(dev_seed) & (2 ** 64 - 1) if gen else 0)
        use_cond = C.byref(cond) if (inpaint or cond.sr_y or w_rgb >= 0 or w_dep >= 0) else None
        need = _lib.load().ivid_sample_scratch_bytes(handles, len(plans), C.byref(plan), use_cond)
        if need < 0:
            _lib.check(-1, "ivid_sample_scratch_bytes")
        scratch = torch.empty(int(need), dtype=torch.uint8, device=x.device)
        on_loop_stream(hole, stepn, scratch)
        ls.wait_stream(cur)             # the noise of this call was drawn on the caller's stream
        _lib.call("ivid_sample", handles, len(plans), C.byref(plan), _lib.ptr(classes), use_cond, _lib.ptr(x), _lib.ptr(stepn),
                  _lib.ptr(x0), _lib.ptr(scratch), int(need), C.c_void_p(ls.cuda_stream))
    cur.wait_stream(ls)
    for p in plans[1:]:
        p.stream.wait_stream(ls)
    return AttrDict({"samples": x, "pred_x_t": [], "pred_x_0": [x0]})


def write_plan_file(path, kind, hw, batch, t_model, engine_of_step, coefs, classes=None, step_noise=None, n_engines=None, noise_seed=None):
    """The host tables of one `ivid_sample` call as a file a C host reads (examples/sample_loop_host.c documents the layout):
    `coefs` = the samplers' `_coef(...)` structs in sampling order, `classes` an int64 tensor / sequence or None, `step_noise` a
    float32 tensor [n_steps, batch, 4, H, W] or None; `noise_seed` (instead of step_noise): the host draws with ivid_randn."""
    import struct

    import numpy as np
    n = len(t_model)
    assert len(engine_of_step) == n and len(coefs) == n
    ne = n_engines if n_engines is not None else max(engine_of_step) + 1
    with open(path, "wb") as f:
        assert step_noise is None or noise_seed is None
        f.write(struct.pack("<8i", 0x50535649, kind, n, hw, batch, int(classes is not None),
                            2 if noise_seed is not None else int(step_noise is not None), ne))
        f.write(struct.pack("<Q", int(noise_seed or 0) & (2 ** 64 - 1)))
        f.write(np.asarray([int(t) for t in t_model], dtype="<i8").tobytes())
        f.write(np.asarray([int(e) for e in engine_of_step], dtype="<i4").tobytes())
        for k in coefs:
            f.write(bytes(k))
        if classes is not None:
            f.write(np.asarray(classes.cpu() if hasattr(classes, "cpu") else classes, dtype="<i8").tobytes())
        if step_noise is not None:
            f.write(np.ascontiguousarray(step_noise.float().cpu().numpy(), dtype="<f4").tobytes())
