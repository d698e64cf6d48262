"""GPU parity of the compensated-storage entry points (precision mode fp16c; include/ivid_hip.h ivid_conv2d_c,
ivid_conv3x3_gn_skip_c, ivid_gn_apply_c, ivid_conv3x3_gn_out_c, ivid_stem_im2col_split) through the C ABI against fp64
torch of the same op.

A tensor with a lo plane stands for the fp32 value hi + lo: the references below use the UNROUNDED residual / input where
the kernel reads hi + lo, and 16-bit-rounded operands only where the kernel feeds an MFMA with the hi plane alone.  An
output with a lo plane must reproduce the fp32 result of the kernel's own arithmetic to ~2^-21 (bar 4e-6) instead of the
2^-11 of a plain fp16 store (2^-17 for a bf16 pair); the split stem / head must reproduce the UNROUNDED layer (bar 2e-5).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import common
import gpu_util as G

pytestmark = pytest.mark.gpu
DT16 = [1, 2]   # IVID_BF16, IVID_F16
PAIR_BAR = {1: 1.5e-5, 2: 4e-6}   # hi + lo vs the fp32 result: 2 x 8 mantissa bits (bf16), 2 x 11 (fp16; fp32 summation order left)


def planes(x, dtype):
    """fp32 NCHW (cpu) -> (hi, lo) NHWC planes on the GPU + the fp32 value they stand for."""
    t = G.tdt(dtype)
    hi = x.to(t)
    lo = (x - hi.float()).to(t)
    val = hi.float() + lo.float()
    return (hi.permute(0, 2, 3, 1).contiguous().cuda(), lo.permute(0, 2, 3, 1).contiguous().cuda(), val)


def joined(hi, lo):
    return (hi.float() + lo.float()).permute(0, 3, 1, 2).contiguous().cpu()


def res_for(mode, s, N, Cout, H, W):
    if mode == 1:
        return common.seeded_randn(s, N, Cout, H, W)
    if mode == 2:
        return common.seeded_randn(s, N, Cout, H // 2, W // 2)
    if mode == 3:
        return common.seeded_randn(s, N, Cout, 2 * H, 2 * W)
    return None


def add_res(y, r, mode):
    if mode == 1:
        return y + r.double()
    if mode == 2:
        return y + F.interpolate(r.double(), scale_factor=2, mode="nearest")
    if mode == 3:
        return y + F.avg_pool2d(r.double(), 2)
    return y


CONV_C = [
    # name, N, H, W, C0, C1, Cout, k, res_mode, tile_cfg, res has a lo plane
    ("1x1_res_same_tile1", 2, 16, 16, 128, 0, 256, 1, 1, 1, True),
    ("3x3_res_up_tile5", 2, 16, 16, 128, 0, 192, 3, 2, 5, True),
    ("3x3_res_down_concat", 2, 8, 8, 64, 64, 128, 3, 3, 1, True),
    ("3x3_bigtile_res", 2, 32, 32, 128, 0, 512, 3, 1, 2, True),
    ("3x3_tile128x384_res_plain", 2, 16, 16, 128, 0, 768, 3, 1, 6, False),   # lo output, plain residual
    ("1x1_nores_tile4", 2, 32, 32, 128, 0, 128, 1, 0, 4, False),
    ("3x3_mtail_cout_tail", 3, 8, 8, 64, 0, 72, 3, 1, 1, True),
]


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("case", CONV_C, ids=[c[0] for c in CONV_C])
def test_conv2d_c_lo_planes(case, dtype):
    name, N, H, W, C0, C1, Cout, k, res_mode, tile, res_lo = case
    L = G.lib()
    s = sum(map(ord, name)) % 1000
    x0 = common.seeded_randn(s, N, C0, H, W)
    x1 = common.seeded_randn(s + 1, N, C1, H, W) if C1 else None
    w = common.seeded_randn(s + 2, Cout, C0 + C1, k, k) / np.sqrt((C0 + C1) * k * k)
    b = common.seeded_randn(s + 3, Cout) * 0.1
    res = res_for(res_mode, s + 4, N, Cout, H, W)
    x = G.rounded(x0 if x1 is None else torch.cat([x0, x1], 1), dtype)
    ref = F.conv2d(x.double(), G.rounded(w, dtype).double(), b.double(), padding=k // 2)
    rh = rl = None
    if res is not None:
        if res_lo:
            rh, rl, rval = planes(res, dtype)
        else:
            rh, rval = G.to_nhwc(res, dtype), G.rounded(res, dtype)
        ref = add_res(ref, rval, res_mode)
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=G.tdt(dtype))
    out_lo = torch.full_like(out, float("nan"))
    blk = L.load().ivid_conv2d_stats_block(N, H, W, Cout, tile)
    stats = torch.full((N * H * W // blk, Cout, 2), float("nan"), device="cuda") if (H * W) % blk == 0 else None
    d0, d1 = G.to_nhwc(x0, dtype), (G.to_nhwc(x1, dtype) if x1 is not None else None)
    wp = G.pack_w(w.permute(0, 2, 3, 1).reshape(Cout, -1), dtype)
    bd = b.cuda()                     # device tensors stay referenced until the launch has finished
    L.call("ivid_conv2d_c", dtype, L.ptr(d0), C0, L.ptr(d1), C1, L.ptr(wp), L.ptr(bd), L.ptr(out), L.ptr(out_lo), L.ptr(rh),
           L.ptr(rl), res_mode, 0, N, H, W, Cout, k * k, tile, L.ptr(stats), G.stream())
    torch.cuda.synchronize()
    got = joined(out, out_lo)
    e = common.rel_l2(got, ref.float())
    e_hi = common.rel_l2(G.from_nhwc(out), ref.float())
    G.report(f"conv_c/{name}/{G.DN[dtype]}", rel_l2=e, rel_l2_hi_plane_alone=e_hi)
    assert torch.isfinite(got).all()
    assert e < PAIR_BAR[dtype], f"{name}: hi + lo is {e} from the fp32 result"
    assert e_hi > 20 * e                         # the hi plane alone carries the usual 16-bit rounding
    assert float((out_lo.float().abs() / out.float().abs().clamp_min(1e-6)).max()) <= 2.0 ** (-8 if dtype == 1 else -11)  # |lo| <= ulp(hi)/2
    if stats is not None:                        # GroupNorm partials describe hi + lo
        o = (out.float() + out_lo.float()).reshape(-1, blk, Cout)
        sref = torch.stack([o.sum(1), (o * o).sum(1)], -1)
        assert float((stats - sref).abs().max() / sref.abs().max()) < 1e-5


def test_conv2d_c_with_null_planes_equals_conv2d():
    L = G.lib()
    N, H, W, Cin, Cout = 2, 16, 16, 64, 128
    x = G.to_nhwc(common.seeded_randn(1, N, Cin, H, W), 2)
    w = G.pack_w((common.seeded_randn(2, Cout, Cin, 3, 3) / 24).permute(0, 2, 3, 1).reshape(Cout, -1), 2)
    res = G.to_nhwc(common.seeded_randn(4, N, Cout, H, W), 2)
    a = torch.empty((N, H, W, Cout), device="cuda", dtype=torch.float16)
    b = torch.empty_like(a)
    L.call("ivid_conv2d", 2, L.ptr(x), Cin, None, 0, L.ptr(w), None, L.ptr(a), L.ptr(res), 1, 0, N, H, W, Cout, 9, 0, None, G.stream())
    L.call("ivid_conv2d_c", 2, L.ptr(x), Cin, None, 0, L.ptr(w), None, L.ptr(b), None, L.ptr(res), None, 1, 0, N, H, W, Cout, 9, 0, None,
           G.stream())
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # lo planes are refused where they cannot exist
    lo = torch.empty_like(a)
    assert L.load().ivid_conv2d_c(0, L.ptr(x), Cin, None, 0, L.ptr(w), None, L.ptr(b), L.ptr(lo), None, None, 0, 0, N, H, W, Cout, 9, 0,
                                  None, G.stream()) != 0


FUSED_C = [
    # name, N, H, W, C0, C1, Cout, res_mode, skip channels (S0, S1)
    ("same_res_c256", 2, 32, 32, 64, 0, 256, 1, (0, 0)),
    ("up_res_c192", 2, 32, 32, 64, 0, 192, 2, (0, 0)),
    ("down_res_avgpool_c160", 2, 16, 32, 64, 0, 160, 3, (0, 0)),
    ("concat_nores_c320", 1, 8, 64, 64, 64, 320, 0, (0, 0)),
    ("skip_cat_128+64_c192", 2, 16, 32, 64, 0, 192, 0, (128, 64)),
    # Cout <= 128 and H % 16 == 0: the 16x32x128 variant (csrc/conv3x3_fused128.hip)
    ("n128_same_res_c128", 2, 32, 32, 128, 0, 128, 1, (0, 0)),
    ("n128_up_res_c96tail", 2, 32, 64, 64, 0, 96, 2, (0, 0)),
    ("n128_down_res_avgpool_c64", 3, 16, 32, 64, 0, 64, 3, (0, 0)),
    ("n128_skip_cat_128+128_c128", 2, 32, 32, 128, 0, 128, 0, (128, 128)),
]


@pytest.mark.parametrize("in_lo", [False, True], ids=["in_hi", "in_hi+lo"])
@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("case", FUSED_C, ids=[c[0] for c in FUSED_C])
def test_conv3x3_gn_skip_c_lo_planes(case, dtype, in_lo):
    """in_lo: the first convolution input carries a lo plane (the halo transform starts from hi + lo); a concat partner stays a
    plain tensor, which exercises the weight-0 stand-in of a source without a lo plane."""
    name, N, H, W, C0, C1, Cout, res_mode, (S0, S1) = case
    L = G.lib()
    s = sum(map(ord, name)) % 1000
    Cc = C0 + C1
    x0 = common.seeded_randn(s, N, C0, H, W)
    x1 = common.seeded_randn(s + 1, N, C1, H, W) if C1 else None
    a = 0.5 + 0.5 * torch.rand(N, Cc, generator=torch.Generator().manual_seed(s))
    b = 0.3 * common.seeded_randn(s + 2, N, Cc)
    w = common.seeded_randn(s + 3, Cout, Cc, 3, 3) / np.sqrt(Cc * 9)
    bias = common.seeded_randn(s + 4, Cout) * 0.1
    res = res_for(res_mode, s + 5, N, Cout, H, W)
    x0 = x0 * 3.0                                    # spread the values over several binades: the lo plane matters
    if in_lo:
        d0, d0l, x0v = planes(x0, dtype)
    else:
        d0, d0l, x0v = G.to_nhwc(x0, dtype), None, G.rounded(x0, dtype)
    x = x0v if x1 is None else torch.cat([x0v, G.rounded(x1, dtype)], 1)
    act = G.rounded(F.silu(x * a[:, :, None, None] + b[:, :, None, None]), dtype)
    ref = F.conv2d(act.double(), G.rounded(w, dtype).double(), bias.double(), padding=1)
    sk0 = sk1 = wsk = None
    if S0:
        k0 = common.seeded_randn(s + 6, N, S0, H, W)
        k1 = common.seeded_randn(s + 7, N, S1, H, W) if S1 else None
        wk = common.seeded_randn(s + 8, Cout, S0 + S1, 1, 1) / np.sqrt(S0 + S1)
        ref = ref + F.conv2d(G.rounded(k0 if k1 is None else torch.cat([k0, k1], 1), dtype).double(), G.rounded(wk, dtype).double())
        sk0, sk1 = G.to_nhwc(k0, dtype), (G.to_nhwc(k1, dtype) if k1 is not None else None)
        wsk = G.pack_w(wk.reshape(Cout, -1), dtype)
    rh = rl = None
    if res is not None:
        rh, rl, rval = planes(res, dtype)
        ref = add_res(ref, rval, res_mode)
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=G.tdt(dtype))
    out_lo = torch.full_like(out, float("nan"))
    stats = torch.full((N * H * W // 128, Cout, 2), float("nan"), device="cuda")
    d1 = G.to_nhwc(x1, dtype) if x1 is not None else None
    ab = torch.stack([a, b], -1).contiguous().cuda()
    wp = G.pack_w(w.permute(0, 2, 3, 1).reshape(Cout, -1), dtype)
    bd = bias.cuda()
    L.call("ivid_conv3x3_gn_skip_c", dtype, L.ptr(d0), L.ptr(d0l), C0, L.ptr(d1), None, C1, L.ptr(ab), 0, L.ptr(wp), L.ptr(bd), L.ptr(out),
           L.ptr(out_lo), L.ptr(rh), L.ptr(rl), res_mode, N, H, W, Cout, L.ptr(stats), L.ptr(sk0), S0, L.ptr(sk1), S1, L.ptr(wsk),
           G.stream())
    torch.cuda.synchronize()
    got = joined(out, out_lo)
    e = common.rel_l2(got, ref.float())
    e_hi = common.rel_l2(G.from_nhwc(out), ref.float())
    G.report(f"conv3x3_gn_c/{name}/{G.DN[dtype]}/{'in_lo' if in_lo else 'in_hi'}", rel_l2=e, rel_l2_hi_plane_alone=e_hi)
    assert torch.isfinite(got).all()
    # the halo transform's hardware exp / rcp move a few activations across a 16-bit rounding boundary (~1e-5 on the output)
    assert e < (2e-4 if dtype == 1 else 6e-5), f"{name}: hi + lo is {e} from the fp32 result"
    assert e_hi > 4 * e
    o = (out.float() + out_lo.float()).reshape(N, H // 4, 4, W // 32, 32, Cout).permute(0, 1, 3, 2, 4, 5).reshape(-1, 128, Cout)
    sref = torch.stack([o.sum(1), (o * o).sum(1)], -1)
    assert float((stats - sref).abs().max() / sref.abs().max()) < 1e-5


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("resample,act", [(0, 1), (0, 0), (1, 1), (2, 1)])
def test_gn_apply_c_reads_hi_plus_lo(dtype, resample, act):
    L = G.lib()
    N, H, W, C0, C1 = 2, 16, 16, 64, 128
    x0, x1 = common.seeded_randn(1, N, C0, H, W) * 3, common.seeded_randn(2, N, C1, H, W)
    a = 0.5 + 0.5 * torch.rand(N, C0 + C1, generator=torch.Generator().manual_seed(3))
    b = 0.3 * common.seeded_randn(4, N, C0 + C1)
    h0, l0, v0 = planes(x0, dtype)
    d1 = G.to_nhwc(x1, dtype)                                   # second source without a lo plane
    y = torch.cat([v0, G.rounded(x1, dtype)], 1).double() * a[:, :, None, None].double() + b[:, :, None, None].double()
    if act:
        y = F.silu(y)
    if resample == 1:
        y = F.interpolate(y, scale_factor=2, mode="nearest")
    elif resample == 2:
        y = F.avg_pool2d(y, 2)
    Ho, Wo = y.shape[2:]
    out = torch.full((N, Ho, Wo, C0 + C1), float("nan"), device="cuda", dtype=G.tdt(dtype))
    ab = torch.stack([a, b], -1).contiguous().cuda()
    L.call("ivid_gn_apply_c", dtype, L.ptr(h0), L.ptr(l0), C0, L.ptr(d1), None, C1, L.ptr(ab), L.ptr(out), N, H, W, resample, act,
           G.stream())
    torch.cuda.synchronize()
    got = G.from_nhwc(out)
    # exact up to the output rounding: compare against the fp64 result rounded the same way
    want = G.rounded(y.float(), dtype)
    ulp = 2.0 ** (-7 if dtype == 1 else -10)
    assert float(((got - want).abs() / want.abs().clamp_min(1e-3)).max()) <= 1.01 * ulp
    assert common.rel_l2(got[:, :C0], y.float()[:, :C0]) < (3e-3 if dtype == 1 else 4e-4)
    # ... and the lo plane matters: the hi plane alone is measurably farther from the fp64 result before rounding
    out2 = torch.empty_like(out)
    L.call("ivid_gn_apply", dtype, L.ptr(h0), C0, L.ptr(d1), C1, L.ptr(ab), L.ptr(out2), N, H, W, resample, act, G.stream())
    torch.cuda.synchronize()
    assert not torch.equal(out, out2)


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("shape", [(2, 32, 32, 128, 4, True), (1, 16, 64, 256, 4, False), (2, 8, 32, 64, 7, True)])
def test_conv3x3_gn_out_c_split_head(dtype, shape):
    """Split output head: GN-apply + SiLU + conv3x3 evaluated to ~2^-21 on hi + lo input, against the UNROUNDED fp64 layer."""
    N, H, W, Cc, Cout, with_lo = shape
    L = G.lib()
    x = common.seeded_randn(11, N, Cc, H, W) * 2
    a = 0.5 + 0.5 * torch.rand(N, Cc, generator=torch.Generator().manual_seed(5))
    b = 0.3 * common.seeded_randn(12, N, Cc)
    w = common.seeded_randn(13, Cout, Cc, 3, 3) / np.sqrt(Cc * 9)
    bias = common.seeded_randn(14, Cout) * 0.1
    if with_lo:
        xh, xl, xv = planes(x, dtype)
    else:
        xh, xl, xv = G.to_nhwc(x, dtype), None, G.rounded(x, dtype)
    ref = F.conv2d(F.silu(xv.double() * a[:, :, None, None].double() + b[:, :, None, None].double()), w.double(), bias.double(),
                   padding=1).float()
    t = G.tdt(dtype)
    w2 = w.permute(0, 2, 3, 1).reshape(Cout, -1)
    whi = w2.to(t)
    wlo = (w2 - whi.float()).to(t)
    out = torch.full((N, Cout, H, W), float("nan"), device="cuda")
    ab = torch.stack([a, b], -1).contiguous().cuda()
    whd, wld, bd = whi.cuda(), wlo.cuda(), bias.cuda()
    L.call("ivid_conv3x3_gn_out_c", dtype, L.ptr(xh), L.ptr(xl), Cc, L.ptr(ab), L.ptr(whd), L.ptr(wld), L.ptr(bd),
           L.ptr(out), N, H, W, Cout, G.stream())
    torch.cuda.synchronize()
    e = common.rel_l2(out.cpu(), ref)
    out1 = torch.empty_like(out)
    L.call("ivid_conv3x3_gn_out", dtype, L.ptr(xh), Cc, L.ptr(ab), L.ptr(whd), L.ptr(bd), L.ptr(out1), N, H, W, Cout,
           G.stream())
    torch.cuda.synchronize()
    e1 = common.rel_l2(out1.cpu(), ref)
    G.report(f"conv3x3_gn_out_c/{G.DN[dtype]}/{N}x{H}x{W}x{Cc}", rel_l2_split=e, rel_l2_single_product=e1)
    assert torch.isfinite(out).all()
    assert e < (2e-4 if dtype == 1 else 2e-5), e          # bf16 hi + lo = 16 bits, fp16 hi + lo = 22 bits
    assert e1 > 10 * e


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("cin", [4, 10])
def test_stem_split_reproduces_the_unrounded_convolution(dtype, cin):
    """ivid_stem_im2col_split + ivid_conv2d(taps = 1) with [w_hi | w_hi | w_lo] rows vs F.conv2d in fp64 on UNROUNDED operands."""
    L = G.lib()
    N, Bsrc, H, W, Cout = 4, 2, 32, 32, 128
    x = common.seeded_randn(21, Bsrc, cin, H, W)
    w = common.seeded_randn(22, Cout, cin, 3, 3) / np.sqrt(cin * 9)
    bias = common.seeded_randn(23, Cout) * 0.1
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1).float().repeat(N // Bsrc, 1, 1, 1)
    t = G.tdt(dtype)
    K9 = 9 * cin
    Kpad = (3 * K9 + 63) // 64 * 64
    w2 = w.permute(0, 2, 3, 1).reshape(Cout, K9)
    whi = w2.to(t)
    wlo = (w2 - whi.float()).to(t)
    w3 = F.pad(torch.cat([whi, whi, wlo], 1), (0, Kpad - 3 * K9)).contiguous().cuda()
    col = torch.full((N, H, W, Kpad), float("nan"), device="cuda", dtype=t)
    xd, bd = x.cuda(), bias.cuda()
    L.call("ivid_stem_im2col_split", dtype, L.ptr(xd), Bsrc, N, cin, H, W, Kpad, L.ptr(col), G.stream())
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=t)
    out_lo = torch.full_like(out, float("nan"))
    L.call("ivid_conv2d_c", dtype, L.ptr(col), Kpad, None, 0, L.ptr(w3), L.ptr(bd), L.ptr(out), L.ptr(out_lo), None, None, 0, 0,
           N, H, W, Cout, 1, 0, None, G.stream())
    torch.cuda.synchronize()
    assert torch.isfinite(col.float()).all()
    e = common.rel_l2(joined(out, out_lo), ref)
    G.report(f"stem_split/{G.DN[dtype]}/cin{cin}", rel_l2=e)
    assert e < (1e-4 if dtype == 1 else 5e-6), e


SKIP_S = [
    # name, N, H, W, C0 (conv input), Cout, res... none; skip channels (S0, S1); in_lo
    ("skip_cat_256+256_c256_like_ob12", 2, 32, 32, 128, 256, (256, 256), True),
    ("skip_single_192_c192", 1, 16, 64, 64, 192, (192, 0), False),
    ("skip_cat_128+64_c320_two_ntiles", 2, 8, 32, 64, 320, (128, 64), True),
]


@pytest.mark.parametrize("case", SKIP_S, ids=[c[0] for c in SKIP_S])
def test_conv3x3_gn_skip_s_split_precision_skip_phase(case):
    """ivid_conv3x3_gn_skip_s (precision mode fp16s): the 1x1 skip_connection term of the output (adm.py:190,222) must reproduce
    conv1x1(UNROUNDED x = hi + lo, UNROUNDED w) to ~2^-20 -- x_hi.w_hi + x_lo.w_hi + x_hi.w_lo -- where ivid_conv3x3_gn_skip_c
    (one MFMA pass on x_hi, w_hi) is 2^-11 off.  The 3x3 part keeps its single fp16 pass in both (same operands in the reference)."""
    name, N, H, W, C0, Cout, (S0, S1), in_lo = case
    dtype = 2
    L = G.lib()
    s = sum(map(ord, name)) % 1000
    x0 = common.seeded_randn(s, N, C0, H, W) * 3.0
    a = 0.5 + 0.5 * torch.rand(N, C0, generator=torch.Generator().manual_seed(s))
    b = 0.3 * common.seeded_randn(s + 2, N, C0)
    w = common.seeded_randn(s + 3, Cout, C0, 3, 3) / np.sqrt(C0 * 9) * 0.05          # small 3x3 branch: the skip term dominates
    bias = common.seeded_randn(s + 4, Cout) * 0.1
    d0, d0l, x0v = planes(x0, dtype) if in_lo else (G.to_nhwc(x0, dtype), None, G.rounded(x0, dtype))
    act = G.rounded(F.silu(x0v * a[:, :, None, None] + b[:, :, None, None]), dtype)
    ref3 = F.conv2d(act.double(), G.rounded(w, dtype).double(), bias.double(), padding=1)
    k0 = common.seeded_randn(s + 6, N, S0, H, W) * 4.0
    k1 = common.seeded_randn(s + 7, N, S1, H, W) * 2.0 if S1 else None
    wk = common.seeded_randn(s + 8, Cout, S0 + S1, 1, 1) / np.sqrt(S0 + S1)
    k0h, k0l, k0v = planes(k0, dtype)
    k1h, k1l, k1v = planes(k1, dtype) if k1 is not None else (None, None, None)
    kv = k0v if k1 is None else torch.cat([k0v, k1v], 1)
    ref_exact = ref3 + F.conv2d(kv.double(), wk.double())                                               # unrounded skip operands
    ref_single = ref3 + F.conv2d(G.rounded(kv, dtype).double(), G.rounded(wk, dtype).double())          # one fp16 pass
    t = G.tdt(dtype)
    wk2 = wk.reshape(Cout, -1)
    wkh = wk2.to(t)
    wkl = (wk2 - wkh.float()).to(t)
    wkhd, wkld = wkh.cuda(), wkl.cuda()
    ab = torch.stack([a, b], -1).contiguous().cuda()
    wp = G.pack_w(w.permute(0, 2, 3, 1).reshape(Cout, -1), dtype)
    bd = bias.cuda()
    outs = {}
    for split in (True, False):
        out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=t)
        out_lo = torch.full_like(out, float("nan"))
        stats = torch.full((N * H * W // 128, Cout, 2), float("nan"), device="cuda")
        L.call("ivid_conv3x3_gn_skip_s", dtype, L.ptr(d0), L.ptr(d0l), C0, None, None, 0, L.ptr(ab), 0, L.ptr(wp), L.ptr(bd), L.ptr(out),
               L.ptr(out_lo), None, None, 0, N, H, W, Cout, L.ptr(stats), L.ptr(k0h), S0, L.ptr(k1h), S1, L.ptr(wkhd),
               L.ptr(k0l) if split else None, L.ptr(k1l) if split else None, L.ptr(wkld) if split else None, G.stream())
        torch.cuda.synchronize()
        outs[split] = joined(out, out_lo)
        assert torch.isfinite(outs[split]).all()
        o = (out.float() + out_lo.float()).reshape(N, H // 4, 4, W // 32, 32, Cout).permute(0, 1, 3, 2, 4, 5).reshape(-1, 128, Cout)
        sref = torch.stack([o.sum(1), (o * o).sum(1)], -1)
        assert float((stats - sref).abs().max() / sref.abs().max()) < 1e-5
    e_split, e_single = common.rel_l2(outs[True], ref_exact.float()), common.rel_l2(outs[False], ref_exact.float())
    e_single_own = common.rel_l2(outs[False], ref_single.float())
    G.report(f"conv3x3_gn_skip_s/{name}", rel_l2_split_vs_exact=e_split, rel_l2_single_pass_vs_exact=e_single,
             rel_l2_single_pass_vs_its_own_operands=e_single_own)
    assert e_single_own < 2e-5            # skip_weight_lo == NULL is ivid_conv3x3_gn_skip_c
    assert e_split < 1.5e-5, e_split      # w_lo is an fp16 subnormal for small weights: ~2^-20 instead of 2^-22
    assert e_single > 10 * e_split, (e_single, e_split)
    # refused where the split phase does not exist
    assert L.load().ivid_conv3x3_gn_skip_s(dtype, L.ptr(d0), None, C0, None, None, 0, L.ptr(ab), 0, L.ptr(wp), L.ptr(bd), L.ptr(out), L.ptr(out_lo),
                                           None, None, 0, N, H, W, Cout, None, L.ptr(k0h), S0, L.ptr(k1h), S1, L.ptr(wkhd), None, None,
                                           L.ptr(wkld), G.stream()) != 0      # lo weights without lo planes of the sources


@pytest.mark.parametrize("dtype", DT16)
def test_f32_to_hilo_planes(dtype):
    L = G.lib()
    x = (common.seeded_randn(5, 3, 17, 8, 64) * 7.0).cuda()
    t = G.tdt(dtype)
    hi = torch.full(x.shape, float("nan"), device="cuda", dtype=t)
    lo = torch.full_like(hi, float("nan"))
    L.call("ivid_f32_to_hilo", dtype, L.ptr(x), L.ptr(hi), L.ptr(lo), x.numel(), G.stream())
    torch.cuda.synchronize()
    assert torch.equal(hi, x.to(t)) and torch.equal(lo, (x - x.to(t).float()).to(t))
    assert common.rel_l2(hi.float() + lo.float(), x) < (1.5e-5 if dtype == 1 else 1e-6)
    assert L.load().ivid_f32_to_hilo(0, L.ptr(x), L.ptr(hi), L.ptr(lo), x.numel(), G.stream()) != 0
    assert L.load().ivid_f32_to_hilo(dtype, L.ptr(x), L.ptr(hi), L.ptr(lo), 12, G.stream()) != 0


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("with_lo", [True, False])
def test_gn_apply_p_also_emits_the_pooled_raw_input(dtype, with_lo):
    """ivid_gn_apply_p: out = avg_pool(silu(x*a+b)) exactly as ivid_gn_apply_c, plus x_upd(x) = avg_pool(x) (adm.py:205-208) as hi
    [+ lo] planes -- the residual of a `down` ResBlock."""
    L = G.lib()
    N, H, W, C0 = 2, 16, 32, 128
    x0 = common.seeded_randn(31, N, C0, H, W) * 3
    a = 0.5 + 0.5 * torch.rand(N, C0, generator=torch.Generator().manual_seed(7))
    b = 0.3 * common.seeded_randn(32, N, C0)
    if with_lo:
        h0, l0, v0 = planes(x0, dtype)
    else:
        h0, l0, v0 = G.to_nhwc(x0, dtype), None, G.rounded(x0, dtype)
    t = G.tdt(dtype)
    ab = torch.stack([a, b], -1).contiguous().cuda()
    out = torch.full((N, H // 2, W // 2, C0), float("nan"), device="cuda", dtype=t)
    ph, pl, out_c = torch.full_like(out, float("nan")), torch.full_like(out, float("nan")), torch.full_like(out, float("nan"))
    L.call("ivid_gn_apply_p", dtype, L.ptr(h0), L.ptr(l0), C0, None, None, 0, L.ptr(ab), L.ptr(out), L.ptr(ph), L.ptr(pl) if with_lo else None,
           N, H, W, 2, 1, G.stream())
    L.call("ivid_gn_apply_c", dtype, L.ptr(h0), L.ptr(l0), C0, None, None, 0, L.ptr(ab), L.ptr(out_c), N, H, W, 2, 1, G.stream())
    torch.cuda.synchronize()
    assert torch.equal(out, out_c)
    want = F.avg_pool2d(v0.double(), 2).float()
    if with_lo:
        assert common.rel_l2(joined(ph, pl), want) < PAIR_BAR[dtype]
    assert torch.equal(ph.cpu(), G.to_nhwc(want, dtype).cpu()) or common.rel_l2(G.from_nhwc(ph), want) < (4e-3 if dtype == 1 else 5e-4)
    # refused where it does not exist
    assert L.load().ivid_gn_apply_p(dtype, L.ptr(h0), None, C0, None, None, 0, L.ptr(ab), L.ptr(out), L.ptr(ph), None, N, H, W, 0, 1, G.stream()) != 0


@pytest.mark.parametrize("res_mode", [0, 1])
def test_conv3x3_gn_o16_writes_the_fp16_twin_of_the_bf16x3_result(res_mode):
    """ivid_conv3x3_gn_o16: the split-precision (IVID_BF16X3) fused convolution whose result leaves as fp16 hi + lo planes.  The fp32
    output must be bit-identical to ivid_conv3x3_gn's, the planes must be exactly fp16(v) and fp16(v - hi) of it, with or without
    the fp32 output, and the GroupNorm partials must not change."""
    L = G.lib()
    N, H, W, C0, Cout = 2, 16, 32, 64, 256
    x = common.seeded_randn(41, N, C0, H, W) * 2
    a = 0.5 + 0.5 * torch.rand(N, C0, generator=torch.Generator().manual_seed(9))
    b = 0.3 * common.seeded_randn(42, N, C0)
    w = common.seeded_randn(43, Cout, C0, 3, 3) / np.sqrt(C0 * 9)
    bias = common.seeded_randn(44, Cout) * 0.1
    res = common.seeded_randn(45, N, Cout, H, W) if res_mode else None
    xd, rd = G.to_nhwc(x, 3), (G.to_nhwc(res, 3) if res is not None else None)
    ab = torch.stack([a, b], -1).contiguous().cuda()
    wp, bd = G.pack_w(w.permute(0, 2, 3, 1).reshape(Cout, -1), 3), bias.cuda()
    ref32 = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    st0 = torch.full((N * H * W // 128, Cout, 2), float("nan"), device="cuda")
    L.call("ivid_conv3x3_gn", 3, L.ptr(xd), C0, None, 0, L.ptr(ab), 0, L.ptr(wp), L.ptr(bd), L.ptr(ref32), L.ptr(rd), res_mode, N, H, W, Cout,
           L.ptr(st0), G.stream())
    for keep32 in (True, False):
        out = torch.full_like(ref32, float("nan"))
        hi = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.float16)
        lo = torch.full_like(hi, float("nan"))
        st = torch.full_like(st0, float("nan"))
        L.call("ivid_conv3x3_gn_o16", L.ptr(xd), C0, None, 0, L.ptr(ab), L.ptr(wp), L.ptr(bd), L.ptr(out) if keep32 else None, L.ptr(hi), L.ptr(lo),
               L.ptr(rd), res_mode, N, H, W, Cout, L.ptr(st), G.stream())
        torch.cuda.synchronize()
        if keep32:
            assert torch.equal(out, ref32)
        assert torch.equal(hi, ref32.half()) and torch.equal(lo, (ref32 - ref32.half().float()).half())
        assert torch.equal(st, st0)
    want = F.conv2d(F.silu(x.double() * a[:, :, None, None].double() + b[:, :, None, None].double()), w.double(), bias.double(), padding=1)
    if res is not None:
        want = want + res.double()
    assert common.rel_l2(joined(hi, lo), want.float()) < 4e-5     # the bf16x3 bar: 16 operand bits
    assert L.load().ivid_conv3x3_gn_o16(L.ptr(xd), C0, None, 0, L.ptr(ab), L.ptr(wp), L.ptr(bd), None, L.ptr(hi), None, None, 0, N, H, W, Cout,
                                        None, G.stream()) != 0


@pytest.mark.parametrize("dtype", DT16)
def test_gn_partial_c_sums_hi_plus_lo(dtype):
    L = G.lib()
    N, H, W, C0, C1 = 2, 16, 16, 64, 32
    x0, x1 = common.seeded_randn(51, N, C0, H, W) * 3, common.seeded_randn(52, N, C1, H, W)
    h0, l0, v0 = planes(x0, dtype)
    d1 = G.to_nhwc(x1, dtype)
    nch = L.load().ivid_gn_num_chunks(H * W)
    part = torch.full((N, nch, C0 + C1, 2), float("nan"), device="cuda")
    L.call("ivid_gn_partial_c", dtype, L.ptr(h0), L.ptr(l0), C0, L.ptr(d1), None, C1, N, H * W, L.ptr(part), G.stream())
    torch.cuda.synchronize()
    v = torch.cat([v0, G.rounded(x1, dtype)], 1).double()
    got_s, got_q = part[..., 0].sum(1).cpu().double(), part[..., 1].sum(1).cpu().double()
    assert float((got_s - v.sum((2, 3))).abs().max()) < 2e-3 and float((got_q / (v * v).sum((2, 3)) - 1).abs().max()) < 1e-5
    hi_only = torch.cat([G.rounded(x0, dtype), G.rounded(x1, dtype)], 1).double()
    assert float(((hi_only * hi_only).sum((2, 3))[:, :C0] / got_q[:, :C0] - 1).abs().max()) > 1e-5     # the lo plane is in the sums


@pytest.mark.parametrize("taps,tile", [(1, 0), (9, 1), (1, 4)])
def test_conv2d_o16_writes_the_fp16_twin_of_the_bf16x3_result(taps, tile):
    """ivid_conv2d_o16: the IVID_BF16X3 implicit GEMM whose NHWC result also leaves as fp16 hi + lo planes: the fp32 output and the
    GroupNorm partials are bit-identical to ivid_conv2d's, the planes are exactly fp16(v) and fp16(v - hi)."""
    L = G.lib()
    N, H, W, C0, Cout = 2, 16, 16, 64, 128
    k = 3 if taps == 9 else 1
    x = common.seeded_randn(61, N, C0, H, W)
    w = common.seeded_randn(62, Cout, C0, k, k) / np.sqrt(C0 * taps)
    bias = common.seeded_randn(63, Cout) * 0.1
    xd, bd = G.to_nhwc(x, 3), bias.cuda()
    wp = G.pack_w(w.permute(0, 2, 3, 1).reshape(Cout, -1), 3)
    blk = L.load().ivid_conv2d_stats_block(N, H, W, Cout, tile)
    ref, st0 = torch.full((N, H, W, Cout), float("nan"), device="cuda"), torch.full((N * H * W // blk, Cout, 2), float("nan"), device="cuda")
    L.call("ivid_conv2d", 3, L.ptr(xd), C0, None, 0, L.ptr(wp), L.ptr(bd), L.ptr(ref), None, 0, 0, N, H, W, Cout, taps, tile, L.ptr(st0), G.stream())
    out, st = torch.full_like(ref, float("nan")), torch.full_like(st0, float("nan"))
    hi = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.float16)
    lo = torch.full_like(hi, float("nan"))
    L.call("ivid_conv2d_o16", L.ptr(xd), C0, None, 0, L.ptr(wp), L.ptr(bd), L.ptr(out), L.ptr(hi), L.ptr(lo), None, 0, N, H, W, Cout, taps, tile,
           L.ptr(st), G.stream())
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(st, st0)
    assert torch.equal(hi, ref.half()) and torch.equal(lo, (ref - ref.half().float()).half())
    want = F.conv2d(x.double(), w.double(), bias.double(), padding=k // 2).float()
    assert common.rel_l2(joined(hi, lo), want) < 4e-5
    assert L.load().ivid_conv2d_o16(L.ptr(xd), C0, None, 0, L.ptr(wp), L.ptr(bd), L.ptr(out), L.ptr(hi), None, None, 0, N, H, W, Cout, taps, tile,
                                    None, G.stream()) != 0
