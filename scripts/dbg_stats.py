import sys, os, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from ivid_amd import _lib
lib = _lib.load()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(n, h, cin, cout, taps, res_mode, cfg):
    torch.manual_seed(0)
    x = torch.randn(n, h, h, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, taps * cin, device="cuda") / (taps * cin) ** 0.5).bfloat16()
    b = torch.randn(cout, device="cuda")
    hr = {0: h, 1: h, 2: h // 2, 3: h * 2}[res_mode]
    res = torch.randn(n, hr, hr, cout, device="cuda").bfloat16() if res_mode else None
    out = torch.empty(n, h, h, cout, device="cuda", dtype=torch.bfloat16)
    blk = lib.ivid_conv2d_stats_block(n, h, h, cout, cfg)
    st = torch.full((n * h * h // blk, cout, 2), float("nan"), device="cuda")
    _lib.check(lib.ivid_conv2d(1, x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), out.data_ptr(),
                               res.data_ptr() if res is not None else None, res_mode, 0, n, h, h, cout, taps, cfg, st.data_ptr(), sp), "conv")
    torch.cuda.synchronize()
    o = out.float().reshape(-1, blk, cout)
    ref = torch.stack([o.sum(1), (o * o).sum(1)], -1)
    err = float((st - ref).abs().max() / ref.abs().max())
    return out, st.reshape(n, -1, cout, 2).sum(1), blk, err
for (n, h, cin, cout, taps, rm) in [(64, 32, 512, 512, 1, 1), (64, 64, 256, 256, 9, 3), (64, 32, 512, 512, 9, 3), (64, 128, 64, 256, 9, 0), (128, 32, 512, 512, 1, 1)]:
    o1, s1, b1, e1 = run(n, h, cin, cout, taps, rm, 1)
    o2, s2, b2, e2 = run(n, h, cin, cout, taps, rm, 2)
    oa, sa, ba, ea = run(n, h, cin, cout, taps, rm, 0)
    print((n, h, cin, cout, taps, rm), "blk", b1, b2, ba, "stats err", e1, e2, ea, "out equal", bool(torch.equal(o1, o2)), bool(torch.equal(o1, oa)),
          "per-image stats rel", float((s1 - s2).abs().max() / s1.abs().max()))

print("---- finalize2 on cfg1 vs cfg2 statistics")
def fin(st_raw, n, hw, cout, blk):
    g = torch.ones(cout, device="cuda"); b = torch.zeros(cout, device="cuda")
    ab = torch.empty(n, cout, 2, device="cuda")
    _lib.check(lib.ivid_gn_finalize2(st_raw.data_ptr(), cout, hw // blk, None, 0, 0, n, hw, 32, 1e-5, g.data_ptr(), b.data_ptr(), None, 0, 0, ab.data_ptr(), sp), "fin")
    torch.cuda.synchronize()
    return ab
def run2(n, h, cin, cout, taps, cfg):
    torch.manual_seed(0)
    x = torch.randn(n, h, h, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, taps * cin, device="cuda") / (taps * cin) ** 0.5).bfloat16()
    b = torch.randn(cout, device="cuda")
    out = torch.empty(n, h, h, cout, device="cuda", dtype=torch.bfloat16)
    blk = lib.ivid_conv2d_stats_block(n, h, h, cout, cfg)
    st = torch.full((n * h * h // blk, cout, 2), float("nan"), device="cuda")
    _lib.check(lib.ivid_conv2d(1, x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), out.data_ptr(), None, 0, 0, n, h, h, cout, taps, cfg, st.data_ptr(), sp), "conv")
    torch.cuda.synchronize()
    return fin(st, n, h * h, cout, blk), out
for (n, h, cin, cout, taps) in [(8, 128, 64, 256, 9), (64, 32, 512, 512, 1), (16, 64, 256, 256, 9)]:
    a1, o1 = run2(n, h, cin, cout, taps, 1)
    a2, o2 = run2(n, h, cin, cout, taps, 2)
    ref = torch.nn.functional.group_norm(o1.float().permute(0, 3, 1, 2), 32, eps=1e-5)
    y1 = o1.float().permute(0, 3, 1, 2) * a1[:, :, 0, None, None] + a1[:, :, 1, None, None]
    y2 = o1.float().permute(0, 3, 1, 2) * a2[:, :, 0, None, None] + a2[:, :, 1, None, None]
    print((n, h, cin, cout, taps), "ab rel diff cfg1 vs cfg2", float((a1 - a2).abs().max() / a1.abs().max()),
          "gn err cfg1", float((y1 - ref).abs().max()), "gn err cfg2", float((y2 - ref).abs().max()))
