#!/usr/bin/env python
"""Bitwise check of the persistent (cross-tile pipelined) form of conv3x3_fused_kernel against its one-workgroup-per-tile form:
a launch with >= 2 tiles per CU takes the persistent path, the same pixels launched image by image (tiny grids) do not, and a
tile's result must not depend on which form computed it.  Forms: plain / lo planes out / lo planes in, every residual mode,
upsample, two cout tiles, an odd chunk count, fp32 and the bf16x3 (+ fp16 twins) instantiations."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ivid_amd import _lib  # noqa: E402

lib = _lib.load()
stream = torch.cuda.Stream()
sp = C.c_void_p(stream.cuda_stream)
_seed = [500]


def fresh():   # one generator per tensor (a single generator walked through several large draws handed back NaNs)
    _seed[0] += 1
    gg = torch.Generator(device="cuda"); gg.manual_seed(_seed[0])
    return gg


def rn(*s):
    v = torch.randn(*s, device="cuda", generator=fresh())
    assert torch.isfinite(v).all().item()
    return v
bad = 0
CASES = [  # name, dtype, N, H, W, C0, C1, Cout, up, res_mode, lo_in, lo_out, o16
    ("plain_f16", _lib.F16, 40, 64, 64, 64, 0, 256, 0, 0, 0, 0, 0),
    ("lo_res1_f16", _lib.F16, 40, 64, 64, 64, 64, 256, 0, 1, 0, 1, 0),
    ("loin_res1_f16", _lib.F16, 40, 64, 64, 128, 0, 192, 0, 1, 1, 1, 0),
    ("loin_res2_up_f16", _lib.F16, 40, 64, 64, 64, 0, 256, 1, 2, 1, 1, 0),
    ("loin_res3_f16", _lib.F16, 40, 64, 64, 64, 0, 160, 0, 3, 1, 1, 0),
    ("loin_2ntiles_f16", _lib.F16, 24, 64, 64, 64, 0, 512, 0, 1, 1, 1, 0),
    ("odd_chunks_bf16", _lib.BF16, 40, 64, 64, 192, 0, 256, 0, 1, 0, 0, 0),
    ("fp32", _lib.F32, 40, 64, 64, 32, 32, 256, 0, 1, 0, 0, 0),
    ("bf16x3", _lib.BF16X3, 40, 64, 64, 32, 32, 256, 0, 1, 0, 0, 0),
    ("bf16x3_o16", _lib.BF16X3, 40, 64, 64, 64, 0, 256, 0, 1, 0, 0, 1),
    ("wide_w96", _lib.F16, 48, 40, 96, 64, 0, 256, 0, 1, 1, 1, 0),
]
for (name, dt, N, H, W, C0, C1, Cout, up, rm, lo_in, lo_out, o16) in CASES:
    tdt = {_lib.F16: torch.float16, _lib.BF16: torch.bfloat16}.get(dt, torch.float32)
    Hs, Ws = (H // 2, W // 2) if up else (H, W)
    Cc = C0 + C1
    x0 = rn(N, Hs, Ws, C0).to(tdt); x0l = (rn(N, Hs, Ws, C0) * 1e-3).to(tdt)
    x1 = rn(N, Hs, Ws, C1).to(tdt) if C1 else None
    x1l = (rn(N, Hs, Ws, C1) * 1e-3).to(tdt) if C1 else None
    w = (rn(Cout, 9 * Cc) / (9 * Cc) ** 0.5)
    if dt == _lib.BF16X3:   # [hi8 | lo8] per 8 input channels (plan.split_pack)
        hi = w.bfloat16(); lo = (w - hi.float()).bfloat16()
        w = torch.stack([hi.reshape(Cout, -1, 8), lo.reshape(Cout, -1, 8)], 2).reshape(Cout, -1).contiguous()
    else:
        w = w.to(tdt)
    b = rn(Cout)
    ab = torch.stack([0.5 + torch.rand(N, Cc, device="cuda", generator=fresh()), 0.3 * rn(N, Cc)], -1).contiguous()
    rs = {0: None, 1: (N, H, W, Cout), 2: (N, H // 2, W // 2, Cout), 3: (N, 2 * H, 2 * W, Cout)}[rm]
    res = rn(*rs).to(tdt) if rs else None
    resl = (rn(*rs) * 1e-3).to(tdt) if (rs and lo_out) else None

    def run(n, off=0):
        out = torch.full((n, H, W, Cout), float("nan"), device="cuda", dtype=tdt)
        outl = torch.full((n, H, W, Cout), float("nan"), device="cuda", dtype=tdt) if lo_out else None
        o16h = torch.full((n, H, W, Cout), float("nan"), device="cuda", dtype=torch.float16) if o16 else None
        o16l = torch.full((n, H, W, Cout), float("nan"), device="cuda", dtype=torch.float16) if o16 else None
        st = torch.full((n * H * W // 128, Cout, 2), float("nan"), device="cuda")
        P = lambda t_: None if t_ is None else t_[off:off + n].contiguous().data_ptr()
        keep = [t_[off:off + n].contiguous() if t_ is not None else None for t_ in (x0, x0l, x1, x1l, ab, res, resl)]
        kp = [None if k is None else k.data_ptr() for k in keep]
        if o16:
            _lib.check(lib.ivid_conv3x3_gn_o16(kp[0], C0, kp[2], C1, kp[4], w.data_ptr(), b.data_ptr(), out.data_ptr(), o16h.data_ptr(),
                                               o16l.data_ptr(), kp[5], rm, n, H, W, Cout, st.data_ptr(), sp), name)
        else:
            _lib.check(lib.ivid_conv3x3_gn_skip_c(dt, kp[0], kp[1] if lo_in else None, C0, kp[2], kp[3] if (lo_in and C1) else None, C1, kp[4], up,
                                                  w.data_ptr(), b.data_ptr(), out.data_ptr(), outl.data_ptr() if lo_out else None, kp[5],
                                                  kp[6] if lo_out else None, rm, n, H, W, Cout, st.data_ptr(), None, 0, None, 0, None, sp), name)
        torch.cuda.synchronize()
        return [t_ for t_ in (out, outl, o16h, o16l, st) if t_ is not None]
    big = run(N)
    again = run(N)
    ok = all(torch.equal(a_, b_) for a_, b_ in zip(big, again)) and all(torch.isfinite(t_.float()).all().item() for t_ in big)
    nb = (H // 4) * (W // 32)
    for im in (0, 1, N // 2, N - 1):
        one = run(1, im)
        for a_, b_ in zip(big, one):
            sl = a_[im * nb:(im + 1) * nb] if a_.dim() == 3 else a_[im:im + 1]
            ok = ok and torch.equal(sl, b_)
    ntiles = N * (H // 8) * (W // 32) * ((Cout + 255) // 256)
    print(f"{name:22s} tiles {ntiles:5d}  {'OK' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
