"""`ivid_sample` (include/ivid_hip.h): the whole DDIM / DDPM sampling loop as ONE C call over UNet programs (SURVEY.md §8(b);
reference loops: diffusion/samplers/ddim.py:150-163, ddpm.py:172-185).  The bar is BIT-identity with the host-driven loop of the
same package on the same noise stream (same programs, same kernels, same order) -- which the other tests hold against the live
reference's chains -- plus the reference chain goldens directly, and refusal of malformed plans before anything is enqueued."""
import ctypes as C_
import os
import subprocess

import numpy as np
import pytest
import torch

import common as C
import gpu_util as G
from test_unet_gpu import build

pytestmark = pytest.mark.gpu


def _cpu_noise_fn():
    return lambda shape: torch.randn(shape).cuda()   # torch's CPU generator = the reference's stream


def _both(make, seed, **kw):
    """The same chain driven from the host and as one C call, each on a fresh model and the same CPU noise stream."""
    out = []
    for dev in (False, True):
        smp, args = make()
        torch.manual_seed(seed)
        out.append(smp.sample(*args, verbose=False, noise_fn=_cpu_noise_fn(), keep_intermediates=True, device_loop=dev, **kw))
    return out


def test_ddim_cfg_chain_device_loop_is_bit_identical_and_matches_the_reference_golden():
    from ivid_amd.diffusion import frameworks, samplers
    g = C.load_golden("mini_ddim_cfg")

    def make():
        m, _ = build(C.MINI, 0, "fp32")
        fw = frameworks.ClassifierFreeGuidance(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
        return samplers.DdimSampler(fw), (2,)
    host, dev = _both(make, 5, noise=torch.from_numpy(g["x_T"]).cuda(), classes=torch.from_numpy(g["classes"]).cuda(), steps=5,
                      strength=0.5, eta=0.5)
    assert torch.equal(host.samples, dev.samples)
    assert torch.equal(host.pred_x_0[-1], dev.pred_x_0[-1]) and dev.pred_x_t == []
    e = C.rel_l2(dev.samples.cpu(), g["samples"])
    G.report("device_loop/mini_ddim_cfg", samples=e)
    assert e < 1e-3


def test_ddim_inpaint_chain_with_replacement_and_depth_constraint_device_loop_is_bit_identical():
    from ivid_amd.diffusion import frameworks, samplers
    g = C.load_golden("mini_ddim_inpaint")
    T = lambda k: torch.from_numpy(g[k]).cuda()
    y, mask, mask_rgb, convex = T("y"), T("mask"), T("mask_rgb"), T("convex")

    def make():
        m, _ = build(C.MINI_COND, 2, "fp32")
        fw = frameworks.InpaintCFG(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
        return samplers.DdimSampler(fw), (2,)
    host, dev = _both(make, 7, noise=T("x_T"), classes=T("classes"), steps=4, strength=3.0, y=y, mask=mask, mask_rgb=mask_rgb,
                      replace_rgb=(0.1, y[:, :3], mask_rgb), replace_depth=(0.2, y[:, 3:], mask), constrain_depth=(0.5, convex))
    assert torch.equal(host.samples, dev.samples)
    assert torch.equal(host.pred_x_0[-1], dev.pred_x_0[-1])
    e = C.rel_l2(dev.samples.cpu(), g["samples"])
    G.report("device_loop/mini_ddim_inpaint", samples=e)
    assert e < 1e-3


def test_ddpm_chain_device_loop_in_several_calls_is_bit_identical():
    """100 ancestral steps of an unconditional model (plain programs, classes = NULL), the loop cut into calls of 64 + 36 steps."""
    from ivid_amd.diffusion import frameworks, samplers
    g = C.load_golden("mini_ddpm")

    def make():
        m, _ = build(C.MINI_UNCLASS, 1, "fp32")
        fw = frameworks.GaussianDiffusion(m, timesteps=100, beta_schedule="linear")
        return samplers.DdpmSampler(fw), (2,)
    host, dev = _both(make, 9, noise=torch.from_numpy(g["x_T"]).cuda())
    assert torch.equal(host.samples, dev.samples)
    assert C.rel_l2(dev.samples.cpu(), g["samples"]) < 1e-3


@pytest.mark.parametrize("precision", ["fp16sx", "fp16sa3"])
def test_device_loop_walks_the_precision_ladder_like_the_host_loop(precision):
    """A ladder mode: every step must run the program of the tier its timestep AND the guidance strength select (strength 3: the
    first, pure-noise step in the guidance-aware bf16x3 tier), i.e. the chain is bit-identical to the host loop, which announces
    both per step."""
    from ivid_amd.diffusion import frameworks, samplers
    g = C.load_golden("mini_ddim_cfg")

    def make():
        m, _ = build(C.MINI, 0, precision)
        fw = frameworks.ClassifierFreeGuidance(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
        return samplers.DdimSampler(fw), (2,)
    host, dev = _both(make, 5, noise=torch.from_numpy(g["x_T"]).cuda(), classes=torch.from_numpy(g["classes"]).cuda(), steps=10,
                      strength=3.0, eta=0.0)
    assert torch.equal(host.samples, dev.samples)
    smp, _ = make()
    bb = smp.framework.backbone
    tiers = {bb.tier_of(t - 1, 3.0) for t in range(100, 1001, 100)}
    assert len(tiers) >= 3, tiers      # the 10-step schedule really visits several programs


def test_device_loop_refuses_what_it_does_not_cover():
    from ivid_amd.diffusion import frameworks, samplers
    m, _ = build(C.MINI, 0, "fp32")
    fw = frameworks.ClassifierFreeGuidance(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    smp = samplers.DdimSampler(fw)
    cls = torch.tensor([1, 2]).cuda()
    with pytest.raises(NotImplementedError):
        smp.sample(2, classes=cls, steps=2, strength=0.0, verbose=False, device_loop=True)       # scaled single branch: host loop
    with pytest.raises(NotImplementedError):
        smp.sample(2, classes=cls, steps=2, verbose=False, device_loop=True, some_framework_kwarg=1)


def test_ivid_sample_validates_the_plan_before_enqueueing_anything():
    from ivid_amd import _lib
    m, _ = build(C.MINI, 0, "fp32")
    lib = _lib.load()
    plan_obj = m.plan(2, True)
    S = C.MINI["image_size"]
    h = (C_.c_void_p * 1)(plan_obj.program.value)
    x = torch.zeros(2, 4, S, S, device="cuda")
    before = x.clone()
    stream = torch.cuda.Stream()             # the programs capture their hipGraph on it: not the legacy default stream
    torch.cuda.synchronize()

    def call(n_steps=2, eng=(0, 0), hw=S * S, sigma=0.0, noise=None, scratch_bytes=None, w_rgb=-1.0):
        coefs = (_lib.DdimCoef * n_steps)()
        for k in coefs:
            k.sqrt_recip_ac, k.sqrt_recipm1_ac, k.sqrt_ac_prev, k.dir_coef, k.nonzero = 1.1, 0.5, 0.9, 0.4, 1.0
            k.sigma, k.replace_rgb_w, k.replace_depth_w, k.constrain_w = sigma, w_rgb, -1.0, -1.0
        tm = (C_.c_longlong * n_steps)(*([5] * n_steps))
        e = (C_.c_int * n_steps)(*eng)
        plan = _lib.SamplePlan(_lib.SAMPLE_DDIM, n_steps, hw, tm, C_.cast(coefs, C_.c_void_p), e)
        need = lib.ivid_sample_scratch_bytes(h, 1, C_.byref(plan), None)
        sc = torch.empty(max(int(need), 256) if scratch_bytes is None else scratch_bytes, dtype=torch.uint8, device="cuda")
        return lib.ivid_sample(h, 1, C_.byref(plan), None, None, x.data_ptr(), noise, None, sc.data_ptr(), sc.numel(),
                               stream.cuda_stream), need
    assert call(eng=(0, 1))[0] != 0 and b"engine_of_step" in lib.ivid_last_error()
    assert call(hw=S * S + 1)[0] != 0 and b"image size" in lib.ivid_last_error()
    assert call(sigma=0.5)[0] != 0 and b"step_noise" in lib.ivid_last_error()
    assert call(w_rgb=0.1)[0] != 0 and b"missing" in lib.ivid_last_error()
    assert call(scratch_bytes=256)[0] != 0 and b"scratch" in lib.ivid_last_error()
    torch.cuda.synchronize()
    assert torch.equal(x, before)            # nothing ran
    st, need = call()
    torch.cuda.synchronize()
    assert st == 0 and need > 0 and torch.isfinite(x).all()


def test_c_host_samples_a_guided_ddim_chain_from_engine_files_without_python(tmp_path):
    """examples/sample_loop_host.c: engine files of every precision tier the schedule visits + the plan tables -> ONE ivid_sample
    call from a C program.  Must equal the Python-driven chain of the same model bit for bit."""
    from ivid_amd import _lib
    from ivid_amd import build as B
    from ivid_amd.diffusion import frameworks, samplers
    from ivid_amd.diffusion.samplers import device_loop
    from ivid_amd.diffusion.samplers.utils import equivalent_timestep
    if not os.path.exists(B.LOOP_BIN):
        raise RuntimeError(f"{B.LOOP_BIN} missing: python -m ivid_amd.build")
    args, bs, steps, strength = C.MINI128, 2, 10, 3.0
    m, _ = build(args, 0, "fp16sx")
    fw = frameworks.ClassifierFreeGuidance(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    smp = samplers.DdimSampler(fw)
    S = args["image_size"]
    xT = C.seeded_randn(21, bs, 4, S, S).cuda()
    cls = torch.tensor([3, 7]).cuda()
    want = smp.sample(bs, noise=xT, classes=cls, steps=steps, strength=strength, eta=0.0, verbose=False).samples.cpu().numpy()
    jump = 1000 // steps
    pairs = [(jump * (i + 1), jump * i) for i in reversed(range(steps))]
    tiers = [m.tier_of(equivalent_timestep(fw, t - 1), strength) for t, _ in pairs]
    order = sorted(set(tiers))
    assert len(order) == 3
    engines = []
    for k in order:
        engines.append(str(tmp_path / f"tier{k}.eng"))
        m.export_engine(bs, True, path=engines[-1], high_t=k)
    device_loop.write_plan_file(str(tmp_path / "plan.bin"), _lib.SAMPLE_DDIM, S * S, bs, [t - 1 for t, _ in pairs],
                                [order.index(k) for k in tiers],
                                [smp._coef(t, tp, 0.0, strength, False, -1.0, -1.0, -1.0) for t, tp in pairs], classes=cls)
    xT.cpu().numpy().tofile(tmp_path / "xT.bin")
    # the same schedule with eta = 0.5 and the library's own noise (the plan file carries the seed, no noise tensor)
    want_eta = smp.sample(bs, noise=xT, classes=cls, steps=steps, strength=strength, eta=0.5, verbose=False, device_loop=True,
                          device_noise_seed=11).samples.cpu().numpy()
    device_loop.write_plan_file(str(tmp_path / "plan_eta.bin"), _lib.SAMPLE_DDIM, S * S, bs, [t - 1 for t, _ in pairs],
                                [order.index(k) for k in tiers],
                                [smp._coef(t, tp, 0.5, strength, False, -1.0, -1.0, -1.0) for t, tp in pairs], classes=cls, noise_seed=11)
    del m, fw, smp
    torch.cuda.synchronize()
    r = subprocess.run([B.LOOP_BIN, str(tmp_path / "plan.bin"), str(tmp_path / "xT.bin"), str(tmp_path / "out.bin")] + engines,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "one call" in r.stdout
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(want.shape)
    assert np.array_equal(got, want)
    r = subprocess.run([B.LOOP_BIN, str(tmp_path / "plan_eta.bin"), str(tmp_path / "xT.bin"), str(tmp_path / "out_eta.bin")] + engines,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(tmp_path / "out_eta.bin", dtype=np.float32).reshape(want_eta.shape), want_eta)
    assert not np.array_equal(want_eta, want)
    # an engine count that does not match the plan is refused by the host
    r = subprocess.run([B.LOOP_BIN, str(tmp_path / "plan.bin"), str(tmp_path / "xT.bin"), str(tmp_path / "o2.bin")] + engines[:2],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "engine" in r.stderr


def test_sample_all_with_the_device_loop_is_bit_identical_to_the_host_loop(monkeypatch):
    """The sampling driver end to end (inference/sample.py:75-139: unconditional DDIM + CFG -> mesh -> warp -> InpaintCFG with the
    reference's conditioning wiring, three views) with every chain as ONE ivid_sample call (IVID_DEVICE_LOOP=1): the same views and
    conditioning tensors as the host-driven loops, bit for bit, on per-seed device noise."""
    from ivid_amd.diffusion import frameworks
    from ivid_amd.diffusion.backbones import AdmUnet2d
    from ivid_amd.inference.sample import sample_all
    from ivid_amd.rgbd_3d import camera
    views = camera.viewset("3x9")[:3]
    outs = []
    for env in ("0", "1"):
        monkeypatch.setenv("IVID_DEVICE_LOOP", env)
        mu = AdmUnet2d(**C.MINI, precision="fp16sx"); mu.load_state_dict(C.synth_weights(C.MINI, 0)); mu = mu.cuda()
        mc = AdmUnet2d(**C.MINI_COND, precision="fp16sx"); mc.load_state_dict(C.synth_weights(C.MINI_COND, 2)); mc = mc.cuda()
        fu = frameworks.ClassifierFreeGuidance(mu, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
        fc = frameworks.InpaintCFG(mc, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
        outs.append(list(sample_all(fu, fc, [0, 1, 2], 6, 4, views, classes=[3, 4, 5], guidance=3.0, batchsize=3, erode_rgb=1)))
    for (sa, ca), (sb, cb) in zip(*outs):
        assert torch.equal(sa, sb)
        assert all(torch.equal(ca[k], cb[k]) for k in ca)


def test_superres_chain_device_loop_is_bit_identical_and_matches_the_reference_golden(monkeypatch):
    """SuperResCFG + DDIM (sr_cfg.py:23-60 through `super_resolve`): the per-step bilinear conditioning inside the one C call."""
    from ivid_amd.diffusion import frameworks
    from ivid_amd.inference.superres import super_resolve
    g = C.load_golden("mini_superres")
    low, cls2 = torch.from_numpy(g["low"]).cuda(), torch.from_numpy(g["classes"]).cuda()
    outs = []
    for env in ("0", "1"):
        monkeypatch.setenv("IVID_DEVICE_LOOP", env)
        ms, _ = build(C.MINI_SR, 7, "fp32")
        fw = frameworks.SuperResCFG(ms, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
        torch.manual_seed(3)
        outs.append(super_resolve(fw, low, classes=cls2, steps=4, strength=3.0, noise_fn=_cpu_noise_fn()))
    assert torch.equal(outs[0], outs[1])
    assert C.rel_l2(outs[1].cpu(), g["samples"]) < 1e-3


def test_ivid_randn_matches_the_philox_oracle_and_is_a_function_of_seed_stream_and_index():
    from ivid_amd import _lib
    from oracle import philox_oracle as P
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for seed, sid, n in ((0, 0, 4096), (1234, 5, 1003), (2 ** 63 + 17, 2 ** 40 + 3, 8190)):
        out = torch.full((n + 8,), 7.0, device="cuda")
        _lib.check(lib.ivid_randn(seed, sid, out.data_ptr(), n, st), "randn")
        got = out.cpu().numpy()
        assert np.all(got[n:] == 7.0)                                    # the tail of a partial block is not overrun
        ref = P.randn(seed, sid, n)
        assert np.abs(got[:n] - ref).max() < 4e-6, float(np.abs(got[:n] - ref).max())
    big = torch.empty(1 << 22, device="cuda")
    _lib.check(lib.ivid_randn(99, 1, big.data_ptr(), big.numel(), st), "randn")
    z = big.double()
    assert abs(float(z.mean())) < 2e-3 and abs(float(z.var()) - 1) < 3e-3 and abs(float(((z - z.mean()) ** 4).mean() / z.var() ** 2) - 3) < 2e-2
    # a prefix of a longer draw is the shorter draw (index-addressed), other streams / seeds differ
    small = torch.empty(1024, device="cuda")
    _lib.check(lib.ivid_randn(99, 1, small.data_ptr(), 1024, st), "randn")
    assert torch.equal(small, big[:1024])
    _lib.check(lib.ivid_randn(99, 2, small.data_ptr(), 1024, st), "randn")
    assert not torch.equal(small, big[:1024])
    assert lib.ivid_randn(1, 1, big.data_ptr() + 4, 16, st) != 0             # misaligned output is refused


def test_device_loop_with_library_noise_is_reproducible_and_independent_of_how_the_loop_is_cut():
    """`device_noise_seed`: every draw (DDPM step noise; InpaintCFG hole noise + DDIM eta noise) from ivid_randn streams addressed by
    the GLOBAL step index -- the same chain whether the 100 steps go in one call, in calls of 64 + 36 or of 7."""
    from ivid_amd.diffusion import frameworks, samplers
    g = C.load_golden("mini_ddpm")
    xT = torch.from_numpy(g["x_T"]).cuda()
    outs = []
    for chunk in (128, 64, 7, 64):
        m, _ = build(C.MINI_UNCLASS, 1, "fp32")
        fw = frameworks.GaussianDiffusion(m, timesteps=100, beta_schedule="linear")
        res = samplers.DdpmSampler(fw).sample(2, noise=xT, verbose=False, device_loop=True, device_noise_seed=4242, device_loop_chunk=chunk)
        outs.append(res.samples)
    assert all(torch.equal(outs[0], o) for o in outs[1:]) and torch.isfinite(outs[0]).all()
    m, _ = build(C.MINI_UNCLASS, 1, "fp32")
    fw = frameworks.GaussianDiffusion(m, timesteps=100, beta_schedule="linear")
    other = samplers.DdpmSampler(fw).sample(2, noise=xT, verbose=False, device_loop=True, device_noise_seed=4243).samples
    assert not torch.equal(other, outs[0])
    # the noise the chain used IS the documented stream: replaying it through noise_fn on the host loop gives the same samples
    from ivid_amd import _lib
    lib = _lib.load()
    step = [0]

    def replay(shape):
        z = torch.empty(shape, device="cuda")
        _lib.check(lib.ivid_randn(4242, 2 * step[0], z.data_ptr(), z.numel(), torch.cuda.current_stream().cuda_stream), "randn")
        step[0] += 1
        return z
    m, _ = build(C.MINI_UNCLASS, 1, "fp32")
    fw = frameworks.GaussianDiffusion(m, timesteps=100, beta_schedule="linear")
    host = samplers.DdpmSampler(fw).sample(2, noise=xT, verbose=False, noise_fn=replay).samples
    assert torch.equal(host, outs[0])
    # InpaintCFG: hole noise on stream 2 * step + 1
    gi = C.load_golden("mini_ddim_inpaint")
    T = lambda k: torch.from_numpy(gi[k]).cuda()
    outs = []
    for chunk in (64, 3):
        mc, _ = build(C.MINI_COND, 2, "fp32")
        fc = frameworks.InpaintCFG(mc, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
        outs.append(samplers.DdimSampler(fc).sample(2, noise=T("x_T"), classes=T("classes"), steps=8, strength=3.0, eta=0.5, verbose=False,
                                                    y=T("y"), mask=T("mask"), mask_rgb=T("mask_rgb"), device_loop=True, device_noise_seed=7,
                                                    device_loop_chunk=chunk).samples)
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all()
