"""CPU: host-side logic and the C-ABI surface (no compute calls — there is no GPU here)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import common as C


def test_library_loads_and_exports_every_declared_symbol():
    from ivid_amd import _lib
    hdr = open(os.path.join(C.ROOT, "include", "ivid_hip.h")).read()
    declared = set(re.findall(r"\b(ivid_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ivid_ddim_coef", "ivid_ddpm_coef"}
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ivid_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == declared
    assert lib.ivid_version() >= 1


def test_ctypes_structs_match_header_layout():
    from ivid_amd import _lib
    assert ctypes.sizeof(_lib.DdimCoef) == 11 * 4
    assert ctypes.sizeof(_lib.DdpmCoef) == 7 * 4


def test_product_has_no_cpu_fallback():
    from ivid_amd.diffusion.backbones import AdmUnet2d
    from ivid_amd._lib import IvidHipError
    m = AdmUnet2d(**C.MINI)
    with pytest.raises(IvidHipError):
        m(torch.zeros(1, 4, 32, 32), torch.zeros(1, dtype=torch.long), None)


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(C.ROOT, "ivid_amd")):
        for f in fs:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dp, f)).read(), re.M):
                bad.append(f)
    assert not bad, f"product files import the oracle: {bad}"


@pytest.mark.parametrize("args", [C.MINI, C.MINI_COND, C.MINI_UNCLASS, C.SMALL128, C.LARGE128])
def test_state_dict_schema_and_strict_load(args):
    from ivid_amd.diffusion.backbones import AdmUnet2d
    m = AdmUnet2d(**args)
    sd = C.synth_weights(args, 0)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.state_dict()["out.2.weight"], sd["out.2.weight"])
    # fresh model reproduces the reference's zero-init semantics (adm.py:182,278,486)
    fresh = AdmUnet2d(**args).state_dict()
    assert float(fresh["out.2.weight"].abs().max()) == 0.0
    assert float(fresh["middle_block.1.proj_out.weight"].abs().max()) == 0.0


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference not mounted")
def test_reference_config_json_loads_unchanged_and_schema_matches_reference():
    import sys
    sys.path.insert(0, "/root/reference")
    from diffusion.backbones import AdmUnet2d as Ref
    from ivid_amd.diffusion.backbones import AdmUnet2d
    for fn in sorted(os.listdir("/root/reference/configs")):
        cfg = json.load(open(os.path.join("/root/reference/configs", fn)))
        mine, ref = AdmUnet2d(**cfg["backbone"]["args"]), Ref(**cfg["backbone"]["args"])
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a.keys()) == list(b.keys()), fn
        assert all(a[k].shape == b[k].shape for k in a), fn
        assert torch.equal(a["time_embed.0.freqs"], b["time_embed.0.freqs"])
    sys.path.remove("/root/reference")
    for k in [k for k in sys.modules if k == "diffusion" or k.startswith("diffusion.")]:
        del sys.modules[k]


def test_spec_topology_large():
    from ivid_amd.diffusion.backbones.spec import Attn, Res, build_spec
    sp = build_spec(**C.LARGE128)
    ops = [o for st in sp.stages for o in st.ops]
    assert sum(isinstance(o, Res) for o in ops) == 35 and sum(isinstance(o, Attn) for o in ops) == 16
    assert sp.emb_total == sum(2 * o.cout for o in ops if isinstance(o, Res))
    ups = [o for o in ops if isinstance(o, Res) and o.mode == "up"]
    downs = [o for o in ops if isinstance(o, Res) and o.mode == "down"]
    assert len(ups) == 4 and len(downs) == 4
    assert [o.prefix for o in ups] == ["output_blocks.2.2", "output_blocks.5.2", "output_blocks.8.2", "output_blocks.11.1"]
    assert max(o.cin for o in ops if isinstance(o, Res)) == 2048
    assert {(o.c, o.heads, o.res ** 2) for o in ops if isinstance(o, Attn)} == {(512, 8, 1024), (768, 12, 256), (1024, 16, 64)}


def test_framework_tables_and_sampler_coefficients_match_oracle_formulas():
    from ivid_amd.diffusion import frameworks, samplers
    from oracle import sampler_oracle

    class Dummy:
        image_size, out_channels = 32, 4
        def forward(self, x, times, classes=None):
            return x
    fw = frameworks.ClassifierFreeGuidance(Dummy(), timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    assert np.array_equal(fw.betas, sampler_oracle.linear_betas(1000)) and fw.betas.dtype == np.float64
    assert set(fw.backbone_args.keys()) == {"x", "times", "classes"}
    s = samplers.DdimSampler(fw)
    ac = np.cumprod(1 - fw.betas)
    k = s._coef(1000, 980, 0.5, 0.5, False, -1, -1, -1)
    ab, abp = np.float32(ac[999]), np.float32(ac[979])
    sigma = np.float32(0.5) * np.sqrt((1 - abp) / (1 - ab)) * np.sqrt(1 - ab / abp)
    assert k.sigma == pytest.approx(float(sigma), rel=1e-6)
    assert k.sqrt_recip_ac == np.float32(np.sqrt(1 / ac[999]))
    assert k.nonzero == 1.0 and s._coef(20, 0, 0.0, 0.0, False, -1, -1, -1).nonzero == 0.0
    assert s._coef(20, 0, 0.7, 0.0, False, -1, -1, -1).sigma == 0.0  # alpha_bar_prev[0] = 1
    d = samplers.DdpmSampler(fw)
    assert d._coef(0, 0.0, False).std == 0.0
    assert d.posterior_log_variance_clipped[0] == d.posterior_log_variance_clipped[1]
    cos = frameworks.GaussianDiffusion(Dummy(), timesteps=50, beta_schedule="cosine")
    assert cos.betas.shape == (50,) and cos.betas.max() <= 0.999


def test_sampler_signatures_match_reference_positional_order():
    import inspect
    from ivid_amd.diffusion import samplers
    assert list(inspect.signature(samplers.DdimSampler.sample).parameters)[:9] == [
        "self", "num", "image_size", "noise", "classes", "steps", "clip_denoised", "eta", "verbose"]
    assert list(inspect.signature(samplers.DdpmSampler.sample).parameters)[:8] == [
        "self", "num", "steps", "image_size", "noise", "classes", "clip_denoised", "verbose"]
    assert list(inspect.signature(samplers.DdimSampler.sample_once).parameters)[:10] == [
        "self", "x_t", "t", "t_prev", "classes", "clip_denoised", "eta", "replace_rgb", "replace_depth", "constrain_depth"]


def test_install_aliases_reference_package_names():
    import sys
    import ivid_amd
    saved = {k: v for k, v in sys.modules.items() if k == "diffusion" or k.startswith("diffusion.") or k == "rgbd_3d"}
    try:
        ivid_amd.install()
        import diffusion.backbones as b
        import diffusion.samplers as s
        assert b.AdmUnet2d.__module__.startswith("ivid_amd.") and s.DdimSampler.__module__.startswith("ivid_amd.")
    finally:
        for k in [k for k in sys.modules if k == "diffusion" or k.startswith("diffusion.") or k == "rgbd_3d"]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_unsupported_attention_geometry_is_refused_at_construction():
    """csrc/attn.hip is written for head dim 64 and T % 64 == 0: any other configuration must fail loudly at build_spec
    time instead of striding the qkv tensor wrongly."""
    from ivid_amd.diffusion.backbones import AdmUnet2d
    with pytest.raises(NotImplementedError):
        AdmUnet2d(**dict(C.MINI, num_head_channels=-1, num_heads=1))         # the constructor defaults: head dim = C
    with pytest.raises(NotImplementedError):
        AdmUnet2d(**dict(C.MINI, num_head_channels=32))
    with pytest.raises(NotImplementedError):
        AdmUnet2d(**dict(C.MINI, image_size=24, attention_resolutions=[6]))  # T = 36
    AdmUnet2d(**C.MINI)


def test_label_check_accepts_inference_mode_tensors():
    """The once-per-tensor label validation reads the tensor's version counter; inference-mode tensors have none (reading
    `_version` raises): they are validated on every call instead (round-5 review)."""
    from ivid_amd.diffusion.backbones import AdmUnet2d
    m = AdmUnet2d(**dict(C.MINI, num_classes=10))
    with torch.inference_mode():
        cls = torch.tensor([1, 2])
    m._check_labels(cls)
    m._check_labels(cls)
    assert m._labels_ok is None
    with torch.inference_mode():
        bad = torch.tensor([1, 10])
    with pytest.raises(IndexError):
        m._check_labels(bad)
    ok = torch.tensor([3, 4])
    m._check_labels(ok)
    assert m._labels_ok is not None and m._labels_ok[0]() is ok


def test_adaptive_mode_bookkeeping_of_the_announced_timestep():
    """AdmUnet2d.note_timestep (host logic): only "fp16sa" has a high-t mode; an announcement selects it from adaptive_t upwards, is
    consumed by ONE query, and IVID_ADAPTIVE_T moves the threshold."""
    from ivid_amd.diffusion.backbones import AdmUnet2d
    m = AdmUnet2d(**C.MINI, precision="fp16sa")
    assert (m._base_precision, m._high_t_precision, m.adaptive_t) == ("fp16s", "fp16cs", 150)
    assert m._take_high_t() is False
    for t, want in ((149, False), (150, True), (999, True), (0, False)):
        m.note_timestep(t)
        assert m._take_high_t() is want and m._take_high_t() is False
    m.set_precision("fp16s")
    m.note_timestep(999)
    assert m._high_t_precision is None and m._take_high_t() is False
    os.environ["IVID_ADAPTIVE_T"] = "600"
    try:
        m.set_precision("fp16sa")
        m.note_timestep(599)
        assert m._take_high_t() is False
        m.note_timestep(600)
        assert m._take_high_t() is True
    finally:
        del os.environ["IVID_ADAPTIVE_T"]
    # three tiers ("fp16sa3"): the LAST tier whose t_min <= t; None withdraws an announcement; IVID_ADAPTIVE_T2 moves tier 2
    m.set_precision("fp16sa3")
    assert m._tiers == [("fp16s", 0), ("fp16cs", 150), ("fp16cx", 500)]
    assert [m.tier_of(t) for t in (None, 0, 149, 150, 499, 500, 999)] == [0, 0, 0, 1, 1, 2, 2]
    m.note_timestep(700)
    m.note_timestep(None)
    assert m._take_tier() == 0
    m.note_timestep(700)
    assert m._take_tier() == 2 and m._take_tier() == 0
    os.environ["IVID_ADAPTIVE_T2"] = "800"
    try:
        m.set_precision("fp16sa3")
        assert m.tier_of(700) == 1 and m.tier_of(800) == 2
    finally:
        del os.environ["IVID_ADAPTIVE_T2"]
    with pytest.raises(ValueError):
        AdmUnet2d(**C.MINI, precision="fp16sa").plan(1, False, high_t=3)     # refused before anything touches a device
    with pytest.raises(ValueError):
        AdmUnet2d(**C.MINI, precision="fp16s").plan(1, False, high_t=1)      # a single rung has no other tier
    # the overrides name the RUNG whose threshold they move, in any ladder (IVID_ADAPTIVE_T: fp16cs, _T2: fp16cx, _TS: fp16s), and a
    # ladder that no longer ascends is refused with a message instead of a bare assertion
    os.environ["IVID_ADAPTIVE_T"] = "700"
    try:
        m.set_precision("fp16sx")
        assert m._tiers == [("bf16x3", 0), ("fp16s", 250), ("fp16cs", 700)]
        with pytest.raises(ValueError, match="ascend"):
            m.set_precision("fp16sa3")                                       # fp16cs at 700 behind fp16cx at 500
    finally:
        del os.environ["IVID_ADAPTIVE_T"]
    # the guidance-aware tier: a forward announced with strength s, 1 + 2 s > 3, at t >= 990 runs the exact mode -- tier 0 of the
    # strict ladder, one more plan set of the others; consumed with the timestep; no effect on a single rung
    from ivid_amd import _lib
    m.set_precision("fp16sx")
    assert m._guided_tier == 0 and m._tier_modes == ["bf16x3", "fp16s", "fp16cs"]
    assert [m.tier_of(t, 3.0) for t in (None, 0, 600, 989, 990, 999)] == [0, 0, 2, 2, 0, 0]
    assert [m.tier_of(t, 0.5) for t in (990, 999)] == [2, 2] and m.tier_of(999, 1.0) == 2 and m.tier_of(999, 1.01) == 0
    m.set_precision("fp16sa3")
    assert m._tiers == [("fp16s", 0), ("fp16cs", 150), ("fp16cx", 500)] and m._tier_modes[-1] == _lib.GUIDED_MODE == "bf16x3"
    assert m.tier_of(999, 3.0) == 3 and m.tier_of(999, None) == 2 and m.tier_of(989, 3.0) == 2
    m.note_timestep(999); m.note_guidance(3.0)
    assert m._take_tier() == 3 and m._take_tier() == 0
    m.note_guidance(3.0)                                                     # a strength without a timestep selects nothing
    assert m._take_tier() == 0
    m.set_precision("fp16s")
    m.note_timestep(999); m.note_guidance(3.0)
    assert m._guided_tier is None and m._take_tier() == 0
    # the samplers announce through the framework: backbones without note_timestep are left alone
    from ivid_amd.diffusion.samplers.utils import announce_timestep

    class FW:
        backbone = m
    m.set_precision("fp16sa")
    announce_timestep(FW, 700)
    assert m._take_high_t() is True
    announce_timestep(type("X", (), {"backbone": object()}), 5)
    announce_timestep(object(), 5)
    # a framework with ANOTHER schedule announces the canonical (1000-step linear) timestep of the same noise level, rounded toward
    # the cleaner side; the canonical schedule announces itself unchanged
    import numpy as np
    from ivid_amd.diffusion.frameworks.utils import get_betas_by_name
    from ivid_amd.diffusion.samplers.utils import equivalent_timestep
    canon = type("F", (), {"betas": get_betas_by_name("linear", 1000)})
    assert [equivalent_timestep(canon, t) for t in (0, 249, 250, 999)] == [0, 249, 250, 999]
    ab1000 = np.cumprod(1 - canon.betas)
    for name, n in (("linear", 250), ("linear", 4000), ("cosine", 1000)):
        f = type("F", (), {"betas": get_betas_by_name(name, n)})
        ab = np.cumprod(1 - f.betas)
        prev = -1
        for t in range(0, n, max(1, n // 50)):
            te = equivalent_timestep(f, t)
            assert 0 <= te <= 999 and te >= prev and (ab1000[te] >= ab[t] or te == 0), (name, n, t, te)     # never noisier than the real input
            assert te == 999 or ab1000[te + 1] < ab[t]                                                       # ... and the largest such
            prev = te
    f250 = type("F", (), {"betas": get_betas_by_name("linear", 250), "backbone": m})
    announce_timestep(f250, 124)                       # the middle of a 250-step schedule is canonical t ~ 500, not 124
    assert 480 <= m._t_hint <= 520
    m.note_timestep(None)


def test_precision_names_and_reference_fp16_api():
    from ivid_amd import _lib
    from ivid_amd.diffusion.backbones import AdmUnet2d
    assert _lib.PRECISIONS == {"fp32": 0, "bf16": 1, "fp16": 2, "bf16x3": 3, "fp16c": 2, "fp16cx": 2, "fp16s": 2, "fp16cs": 2, "fp16sa": 2,
                               "fp16sa3": 2, "fp16sx": 2}
    assert _lib.COMPENSATED == {"fp16c": 1, "fp16cx": 2, "fp16s": 3, "fp16cs": 3, "fp16sa": 3, "fp16sa3": 3, "fp16sx": 3}
    assert _lib.NO_ISLAND == {"fp16cs"}
    assert _lib.ADAPTIVE == {"fp16sa": (("fp16s", 0), ("fp16cs", 150)), "fp16sa3": (("fp16s", 0), ("fp16cs", 150), ("fp16cx", 500)),
                             "fp16sx": (("bf16x3", 0), ("fp16s", 250), ("fp16cs", 500))}
    hdr = open(os.path.join(C.ROOT, "include", "ivid_hip.h")).read()
    for name, code in (("IVID_F32", 0), ("IVID_BF16", 1), ("IVID_F16", 2), ("IVID_BF16X3", 3)):
        assert re.search(rf"#define {name} {code}\b", hdr)
    m = AdmUnet2d(**dict(C.MINI, use_fp16=True))
    assert m.precision == _lib.DEFAULT_FP16 == "fp16sx" and m.dtype == torch.float16   # adm.py:333: the reference's attribute
    assert m._base_precision == "bf16x3"                                # an unannounced forward runs the most accurate tier
    m.convert_to_fp32(); assert m.precision == "fp32"
    m.convert_to_fp16(); assert m.precision == "fp16sx"
    m.set_precision("bf16x3"); assert m.precision == "bf16x3"
    with pytest.raises(ValueError):
        m.set_precision("int8")


def test_split_pack_layout_is_hi_then_lo_per_eight_channels():
    """IVID_BF16X3 weight operand (include/ivid_hip.h): 32 bytes per 8 K values = 8 x bf16 hi, 8 x bf16 lo; hi + lo
    reproduces the fp32 weight to 2^-17."""
    from ivid_amd.diffusion.backbones.plan import split_pack
    w = C.seeded_randn(3, 5, 32) * 0.37
    p = split_pack(w)
    assert p.dtype == torch.bfloat16 and p.shape == (5, 64)
    q = p.view(5, 4, 2, 8).float()
    hi, lo = q[:, :, 0].reshape(5, 32), q[:, :, 1].reshape(5, 32)
    assert torch.equal(hi, w.bfloat16().float())
    assert float(((hi + lo) - w).abs().max() / w.abs().max()) < 2 ** -16


def test_cfg_branches_follow_the_reference_for_every_sign_of_strength():
    """classifier_free_guidance.py:39-42 with a CPU stand-in backbone: s > 0 two branches, s = 0 plain eps_c, s < 0 scaled
    eps_c and still one forward; classes None -> unconditional."""
    from ivid_amd.diffusion.frameworks.gaussian_diffusion import cfg_branches, cfg_combine
    calls = []

    def backbone(x, t, classes=None):
        calls.append(classes is not None)
        return x * (2.0 if classes is not None else 3.0)
    x, t, cls = torch.ones(2, 1), torch.zeros(2), torch.tensor([1, 2])
    for s, want, ncalls in ((1.5, (1 + 1.5) * 2.0 - 1.5 * 3.0, 2), (0.0, 2.0, 1), (-0.25, 0.75 * 2.0, 1)):
        calls.clear()
        out = cfg_combine(*cfg_branches(backbone, x, t, cls, s))
        assert torch.allclose(out, torch.full((2, 1), want)) and len(calls) == ncalls, s
    calls.clear()
    assert torch.allclose(cfg_combine(*cfg_branches(backbone, x, t, None, 3.0)), torch.full((2, 1), 3.0)) and calls == [False]


def test_up4_weights_turn_upsample_plus_conv3x3_into_four_2x2_phase_convolutions():
    """include/ivid_hip.h ivid_conv3x3_up: Upsample2d (nearest x2) -> Conv2d 3x3 pad 1 (adm.py:70-83, 203-206) equals, per
    output phase (py, px), a 2x2 convolution of the zero-padded SOURCE with the phase-summed weights (exact in fp64)."""
    import torch.nn.functional as F
    from ivid_amd.diffusion.backbones.plan import up4_weights
    w = C.seeded_randn(5, 6, 4, 3, 3).double()
    x = C.seeded_randn(6, 2, 4, 5, 7).double()
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    w4 = up4_weights(w)
    assert w4.shape == (4 * 6, 4 * 4)
    w4 = w4.reshape(2, 2, 6, 2, 2, 4)                       # [py][px][cout][a][b][cin]
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros_like(ref)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):                           # source pixel (y + py - 1 + a, x + px - 1 + b)
                    out[:, :, py::2, px::2] += torch.einsum("nchw,oc->nohw", xp[:, :, py + a:py + a + 5, px + b:px + b + 7],
                                                            w4[py, px, :, a, b, :])
    assert float((out - ref).abs().max()) < 1e-12


def test_launch_program_object_records_ops_without_a_gpu():
    """csrc/program.hip: the C-side owner of a planned forward.  Creating, filling and destroying it needs no device; the
    slot packing follows each entry point's signature (floats as double, everything else as int64)."""
    import ctypes as CT
    from ivid_amd import _lib
    lib = _lib.load()
    h = CT.c_void_p()
    _lib.call("ivid_program_create", CT.byref(h))
    args = (0x1000, 8, 2, 0x2000, 4, 8, 2, 256, 32, 1e-5, 0x3000, 0x4000, None, 0, 0, 0x5000)    # ivid_gn_finalize2
    arr = _lib.pack_args("ivid_gn_finalize2", args)
    assert arr[9].f == pytest.approx(1e-5) and arr[0].i == 0x1000 and arr[12].i == 0
    _lib.call("ivid_program_add", h, _lib.OP_CODES["ivid_gn_finalize2"], CT.cast(arr, CT.c_void_p), len(args))
    assert lib.ivid_program_num_ops(h) == 1 and lib.ivid_program_has_graph(h) == 0
    with pytest.raises(_lib.IvidHipError):
        _lib.call("ivid_unet_forward", h, None, None, None, None, 1, None)       # no boundary bound yet
    # an argument list that is not the entry point's is refused (run_op reads fixed slots), and so is an unknown code
    for bad_op, bad_n in ((_lib.OP_CODES["ivid_gn_finalize2"], len(args) - 1), (_lib.OP_CODES["ivid_copy"], len(args)), (999, 3)):
        assert lib.ivid_program_add(h, bad_op, CT.cast(arr, CT.c_void_p), bad_n) != 0
    assert lib.ivid_program_num_ops(h) == 1
    _lib.call("ivid_program_destroy", h)
    hdr = open(os.path.join(C.ROOT, "include", "ivid_hip.h")).read()
    for name, code in _lib.OP_CODES.items():
        assert re.search(rf"#define IVID_OP_{name[5:].upper()} {code}\b", hdr), name
        # the library's arity table (what ivid_program_add / ivid_unet_load check) is the ctypes signature's length
        assert lib.ivid_program_op_arity(code) == len(_lib.SIGNATURES[name][1]) - 1, name
    assert lib.ivid_program_op_arity(0) == -1 and lib.ivid_program_op_arity(max(_lib.OP_CODES.values()) + 1) == -1
    assert re.search(rf"#define IVID_ENGINE_ABI {_lib.ENGINE_ABI}\b", hdr)


@pytest.mark.parametrize("cfg", ["MINI", "LARGE128", "SMALL128", "SR256"])
@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_launch_plan_builds_without_a_gpu_and_every_launch_matches_its_c_signature(cfg, precision, monkeypatch):
    """The launch plan is host logic: on torch's `meta` device it can be built without a GPU.  Every recorded launch must
    pack against the ctypes signature of its entry point (argument count and kinds), the four `up` ResBlocks of the
    five-level models go through ivid_conv3x3_up (two in the three-level mini model), and statistics come fused
    (no ivid_gn_partial pass)."""
    import collections
    from ivid_amd import _lib
    from ivid_amd.diffusion.backbones import plan as P
    from ivid_amd.diffusion.backbones.spec import build_spec

    class FakeStream:
        def __init__(self, device=None):
            self.cuda_stream = 0
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setenv("IVID_PY_LAUNCH", "1")           # no C-side program: binding it needs real device pointers
    args = getattr(C, cfg)
    spec = build_spec(**args)
    sd = {k: torch.zeros(s) for k, s in C.schema_for(args)}
    dt = _lib.PRECISIONS[precision]
    pl = P.UNetPlan(spec, P.PackedWeights(spec, sd, "meta", dt), "meta", 2, True)
    cnt = collections.Counter(name for _, name, _ in pl.launches)
    for _fn, name, a in pl.launches:
        _lib.pack_args(name, a)                          # asserts the argument count
        assert name in _lib.OP_CODES, name
    n_up = sum(1 for op in spec.res_ops() if op.mode == "up")
    assert cnt["ivid_conv3x3_up"] == n_up == len(args["channel_mult"]) - 1
    assert cnt["ivid_gn_partial"] == 0 and cnt["ivid_attention"] == sum(1 for st in spec.stages for op in st.ops if hasattr(op, "heads"))
    for _fn, name, a in pl.launches:
        if name == "ivid_conv3x3_up":                    # source side, channels: the activated low-resolution tensor
            assert a[9] == a[10] and (a[9] * a[10]) % 64 == 0 and a[11] > 32
    # stacked CFG forward (2 x 2 rows here): the class-independent in_layers convolution of the first ResBlock runs on one
    # half of the batch (N = 2); its out_layers convolution is one launch PER HALF on that shared result (round 5: no copies),
    # whenever the block takes the fused kernel
    fused = [a for _fn, name, a in pl.launches if name in ("ivid_conv3x3_gn", "ivid_conv3x3_gn_skip")]
    assert cnt["ivid_copy"] == 0
    if cfg == "LARGE128" or precision == "bf16":
        assert [a[12] for a in fused[:4]] == [2, 2, 2, 4] and pl.n == 4, [a[12] for a in fused[:4]]
        assert fused[1][1] == fused[2][1] and fused[1][9] != fused[2][9]        # the two halves read the SAME h1, write different outputs


def test_bench_flop_accounting_adds_up_to_the_reference_count(monkeypatch):
    """bench.py prices every recorded launch with the reference's FLOP count (2 x MACs; the phase-form up-convolution at its
    9-tap count).  Summed over a planned forward of the large model this must reproduce BASELINE's 613.78 GFLOP per
    sample-forward -- minus the one convolution the stacked CFG forward shares between its halves."""
    import importlib.util
    from ivid_amd import _lib
    from ivid_amd.diffusion.backbones import plan as P
    from ivid_amd.diffusion.backbones.spec import build_spec

    class FakeStream:
        def __init__(self, device=None):
            self.cuda_stream = 0
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setenv("IVID_PY_LAUNCH", "1")
    spec_b = importlib.util.spec_from_file_location("ivid_bench", os.path.join(C.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec_b)
    spec_b.loader.exec_module(bench)
    spec = build_spec(**C.LARGE128)
    sd = {k: torch.zeros(s) for k, s in C.schema_for(C.LARGE128)}
    bsrc = 2
    pl = P.UNetPlan(spec, P.PackedWeights(spec, sd, "meta", _lib.BF16), "meta", bsrc, True)
    prof = [(name, a, 1.0) for _fn, name, a in pl.launches]
    fam, _other = bench.kernel_table(prof, "bf16")
    launched = sum(f["flop"] for f in fam.values())
    first = next(op for op in spec.res_ops())
    shared = 2.0 * bsrc * first.res_out ** 2 * first.cout * 9 * first.cin       # the half that is copied instead of computed
    ref = pl.n * bench.GFLOP_PER_SAMPLE_FWD["large"] * 1e9
    assert abs((launched + shared) / ref - 1.0) < 5e-3, (launched + shared) / ref
    up = sum(bench.up_flops(a) for name, a, _ in prof if name == "ivid_conv3x3_up")
    executed = (launched - up * 5.0 / 9.0) / ref
    assert 0.92 < executed < 0.95, executed
    # fp16s (the headline mode): the split-precision correction passes of the 1x1 skip convolutions and the three MFMA passes
    # of the island are EXECUTED work, not algorithmic work -- the planned forward still adds up to the reference's count
    pl = P.UNetPlan(spec, P.PackedWeights(spec, sd, "meta", _lib.F16, comp=3), "meta", bsrc, True)
    prof = [(name, a, 1.0) for _fn, name, a in pl.launches]
    names = [n for n, _, _ in prof]
    # island hand-over: all three tensors leave as fp16 twins written by their own producers (no conversion pass)
    assert names.count("ivid_f32_to_hilo") == 0 and names.count("ivid_conv2d_o16") == 1     # the stem writes its own twin too
    # (stacked CFG plan: the first block's out_layers is one launch per half of the batch on the shared in_layers result)
    assert names.count("ivid_conv3x3_gn_o16") == 3 and names.count("ivid_conv3x3_gn_skip_s") >= 9
    o16 = [a for n, a, _ in prof if n == "ivid_conv3x3_gn_o16"]
    assert o16[0][7] is not None and o16[1][7] is not None and o16[2][7] is None   # only the first block's fp32 form has a reader (the second block)
    assert o16[0][12] == o16[1][12] == bsrc and o16[2][12] == 2 * bsrc
    fam, other = bench.kernel_table(prof, "fp16s")
    launched = sum(f["flop"] for f in fam.values())
    assert abs((launched + shared) / ref - 1.0) < 5e-3, (launched + shared) / ref
    assert "ivid_f32_to_hilo" not in other
    # fp16cs (the high-t half of the adaptive mode): the same plan WITHOUT the island -- no bf16x3 launch, no fp16 twins, the stem
    # in its split 16-bit form, the split-precision skip convolutions all there
    plc = P.UNetPlan(spec, P.PackedWeights(spec, sd, "meta", _lib.F16, comp=3, island=False), "meta", bsrc, True)
    cn = [n for _fn, n, _a in plc.launches]
    assert cn.count("ivid_conv3x3_gn_o16") == 0 and cn.count("ivid_conv2d_o16") == 0 and cn.count("ivid_f32_to_hilo") == 0
    assert cn.count("ivid_stem_im2col_split") == 1 and cn.count("ivid_conv3x3_gn_skip_s") == names.count("ivid_conv3x3_gn_skip_s")
    dts = {a[0] for _fn, n, a in plc.launches if n in ("ivid_conv3x3_gn_skip_c", "ivid_conv3x3_gn_skip_s", "ivid_conv2d_c", "ivid_conv2d")}
    assert dts == {_lib.F32, _lib.F16}, dts            # fp32: the embedding MLP (nn.Linear is never cast, backbones/utils.py:6-13)
    profc = [(n, a, 1.0) for _fn, n, a in plc.launches]
    famc, _ = bench.kernel_table(profc, "fp16cs")
    launched_c = sum(f["flop"] for f in famc.values())
    assert abs((launched_c + shared) / ref - 1.0) < 5e-3


def test_bench_merges_the_adaptive_modes_two_kernel_tables_by_schedule_share():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(C.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lo = ({"conv3x3_fused_kernel": dict(ms=70.0, n=35, flop=7e13, byt=1e11), "attn_kernel": dict(ms=2.0, n=16, flop=1e12, byt=1e9)}, {"ivid_gn_apply": 3.0})
    hi = ({"conv3x3_fused_kernel": dict(ms=60.0, n=35, flop=7e13, byt=9e10), "attn_kernel": dict(ms=2.0, n=16, flop=1e12, byt=1e9)}, {"ivid_gn_apply": 3.0, "ivid_copy": 1.0})
    fam, other = bench.merge_kernel_tables([lo, hi], [0.24, 0.76])
    f = fam["conv3x3_fused_kernel"]
    assert abs(f["ms"] - (0.24 * 70 + 0.76 * 60)) < 1e-9 and abs(f["n"] - 35) < 1e-9 and abs(f["flop"] - 7e13) < 1
    assert abs(other["ivid_gn_apply"] - 3.0) < 1e-9 and abs(other["ivid_copy"] - 0.76) < 1e-9
    fam3, _ = bench.merge_kernel_tables([lo, hi, hi], [0.24, 0.26, 0.5])                       # three tiers
    assert abs(fam3["conv3x3_fused_kernel"]["ms"] - f["ms"]) < 1e-9
    # what the bench line says about an adaptive run: tiers, their share of the 50-step schedule, the timed steps each served
    from ivid_amd.diffusion.backbones import AdmUnet2d
    m = AdmUnet2d(**C.MINI, precision="fp16sa3")
    pairs = [(20 * (i + 1), 20 * i) for i in reversed(range(50))]
    rec = bench.adaptive_record(m, pairs, [999, 499, 19, 259])
    assert [(t["mode"], t["t_min"], t["share_over_the_50_step_schedule"], t["timed_steps"]) for t in rec["tiers"]] == \
        [("fp16s", 0, 0.14, 1), ("fp16cs", 150, 0.36, 2), ("fp16cx", 500, 0.5, 1)]
    e = bench.roofline_entry("conv3x3_fused_kernel", f, 2500.0, 70.0, 128)
    assert e["launches_per_forward"] == 35.0 and abs(e["avg_launch_ms"] - round(f["ms"] / 35, 4)) < 1e-9


@pytest.mark.parametrize("mod", ["ivid_amd.inference.sample", "ivid_amd.inference.render"])
def test_command_line_help_renders(mod):
    """`--help` of the two CLIs (README: `python -m ivid_amd.inference.sample --help`): argparse %-formats every help string."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", mod, "--help"], capture_output=True, text=True, cwd=C.ROOT, timeout=300)
    assert r.returncode == 0 and "usage:" in r.stdout, r.stderr[-400:]
    if mod.endswith("sample"):
        assert "--precision" in r.stdout and "fp16s" in r.stdout


def test_bench_unet_share_walks_the_real_schedules_and_refuses_an_inconsistent_share():
    """bench.py --config c3 | c4 | c5: the UNet share of a batch is the sum over the steps of the REAL schedules, each timed in
    the tier its announced timestep + guidance strength select (round 5 timed one unannounced forward -- the costliest tier --
    for all of them, reported more UNet seconds than the batch took and clamped the share to 0).  Host logic only."""
    import bench
    from ivid_amd.diffusion.backbones import AdmUnet2d
    from ivid_amd.diffusion.frameworks.utils import get_betas_by_name
    m = AdmUnet2d(**C.MINI, precision="fp16sx")
    fw = type("F", (), {"betas": get_betas_by_name("linear", 1000)})
    # DDPM-1000 at strength 3.0: the 10 pure-noise steps in the guidance-aware tier (= tier 0 of the strict ladder), then fp16cs down
    # to t = 500, fp16s down to 250, bf16x3 below
    tiers = bench.schedule_tiers(m, fw, "ddpm", 1000, 3.0)
    assert len(tiers) == 1000 and tiers[:10] == [0] * 10 and tiers[10] == 2 and tiers[-1] == 0
    assert (tiers.count(0), tiers.count(1), tiers.count(2)) == (10 + 250, 250, 490)
    # DDIM-50 (t = 999, 979, ..., 19): one guided step, then 24 / 13 / 12
    tiers = bench.schedule_tiers(m, fw, "ddim", 50, 3.0)
    assert tiers[0] == 0 and (tiers.count(0), tiers.count(1), tiers.count(2)) == (1 + 12, 13, 24)
    assert bench.schedule_tiers(m, fw, "ddim", 50, 0.5)[0] == 2            # no amplification at strength 0.5
    m.set_precision("fp16s")
    assert set(bench.schedule_tiers(m, fw, "ddim", 50, 3.0)) == {0}        # a single rung has one tier
    counts = {0: 13, 1: 13, 2: 24}
    assert abs(bench.unet_seconds(counts, {0: 200.0, 1: 100.0, 2: 90.0}) - (13 * 0.2 + 13 * 0.1 + 24 * 0.09)) < 1e-12
    with pytest.raises(ValueError):
        bench.unet_seconds(counts, {0: 200.0, 1: 100.0})                   # a tier that serves steps must have been timed
    assert abs(bench.share_outside(9.0, 0.5, 10.0) - 0.05) < 1e-12
    assert bench.share_outside(10.2, 0.0, 10.0) < 0                        # timing noise of a loop that is all forwards: reported as is
    with pytest.raises(ValueError):
        bench.share_outside(56.3, 0.0, 51.9)                               # round 5's figures: refused, not clamped to 0


def test_sample_plan_file_layout_matches_what_the_c_host_parses(tmp_path):
    """samplers/device_loop.write_plan_file -> examples/sample_loop_host.c: header of eight int32, then the tables in the order and
    sizes the C host computes its expected file size from (ctypes mirrors of ivid_ddim_coef / ivid_ddpm_coef have the C sizes)."""
    import ctypes
    import struct

    import numpy as np
    import torch
    from ivid_amd import _lib
    from ivid_amd.diffusion.samplers import device_loop
    assert ctypes.sizeof(_lib.DdimCoef) == 44 and ctypes.sizeof(_lib.DdpmCoef) == 28
    n, b, hw = 3, 2, 16
    coefs = []
    for i in range(n):
        k = _lib.DdimCoef()
        k.sigma, k.cfg_strength, k.clip_denoised = 0.25 * i, 3.0, 1
        coefs.append(k)
    noise = torch.arange(n * b * 4 * hw, dtype=torch.float32).reshape(n, b, 4, 4, 4)
    path = tmp_path / "plan.bin"
    device_loop.write_plan_file(str(path), _lib.SAMPLE_DDIM, hw, b, [899, 499, 99], [0, 1, 2], coefs, classes=torch.tensor([5, 7]), step_noise=noise)
    blob = path.read_bytes()
    assert struct.unpack("<8i", blob[:32]) == (0x50535649, 0, n, hw, b, 1, 1, 3) and struct.unpack("<Q", blob[32:40]) == (0,)
    assert len(blob) == 40 + 8 * n + 4 * n + 44 * n + 8 * b + 4 * n * b * 4 * hw
    off = 40
    assert list(np.frombuffer(blob, "<i8", n, off)) == [899, 499, 99]; off += 8 * n
    assert list(np.frombuffer(blob, "<i4", n, off)) == [0, 1, 2]; off += 4 * n
    k1 = _lib.DdimCoef.from_buffer_copy(blob[off + 44:off + 88])
    assert (k1.sigma, k1.cfg_strength, k1.clip_denoised) == (0.25, 3.0, 1); off += 44 * n
    assert list(np.frombuffer(blob, "<i8", b, off)) == [5, 7]; off += 8 * b
    assert np.array_equal(np.frombuffer(blob, "<f4", -1, off), noise.numpy().ravel())
    device_loop.write_plan_file(str(path), _lib.SAMPLE_DDIM, hw, b, [899, 499, 99], [0, 1, 2], coefs, noise_seed=2 ** 63 + 5)
    blob = path.read_bytes()
    assert struct.unpack("<8i", blob[:32])[5:7] == (0, 2) and struct.unpack("<Q", blob[32:40]) == (2 ** 63 + 5,)
    assert len(blob) == 40 + 8 * n + 4 * n + 44 * n
