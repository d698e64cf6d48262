#!/usr/bin/env python
"""Development aid: per-workgroup phase timeline of the fused conv kernel (library built with -DIVID_DEV_ABLATE,
IVID_FUSED_ABLATE=256).  Prints prologue / main loop / epilogue durations and the gap between consecutive workgroups
on the same CU."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ivid_amd import _lib

def main():
    n, h, cin, cout = int(os.environ.get("N", "128")), int(os.environ.get("H", "128")), int(os.environ.get("CIN", "256")), int(os.environ.get("COUT", "256"))
    lib = _lib.load()
    x = torch.randn(n, h, h, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, 9 * cin, device="cuda") / (9 * cin) ** 0.5).bfloat16()
    b = torch.randn(cout, device="cuda")
    ab = torch.rand(n, cin, 2, device="cuda") + 0.5
    out = torch.empty(n, h, h, cout, device="cuda", dtype=torch.bfloat16)
    ntiles = n * (h // 8) * (h // 32) * ((cout + 255) // 256)
    dbg = torch.zeros(ntiles * 13, dtype=torch.int64, device="cuda")
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        _lib.check(lib.ivid_conv3x3_gn(_lib.BF16, x.data_ptr(), cin, None, 0, ab.data_ptr(), 0, w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                       None, 0, n, h, h, cout, dbg.data_ptr(), sp), "fused")
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy()
    d = raw[:ntiles * 5].reshape(-1, 5)
    ph = raw[ntiles * 5:].reshape(ntiles, 2, 4).astype(np.float64)
    t = d[:, :4].astype(np.float64) * 0.01  # us
    t0 = t[:, 0].min()
    pro, main_, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"tiles {ntiles}  kernel span {t[:,3].max()-t0:.1f} us")
    for name, v in (("prologue", pro), ("main", main_), ("epilogue", epi), ("total", t[:, 3] - t[:, 0])):
        print(f"{name:9s} mean {v.mean():7.2f}  p10 {np.percentile(v,10):7.2f}  p50 {np.percentile(v,50):7.2f}  p90 {np.percentile(v,90):7.2f} us")
    # sustained shader clock during the main loop: s_memtime cycles / (100 MHz wall clock ticks)
    cyc = d[:, 4].astype(np.float64)
    mhz = cyc / np.maximum(main_, 1e-9)          # cycles per us = MHz
    print(f"shader clock in the main loop: mean {mhz.mean():.0f} MHz  p10 {np.percentile(mhz,10):.0f}  p90 {np.percentile(mhz,90):.0f}")
    steps = 9 * (cin // 64)
    print(f"cycles per K-step (ideal 2048 = 64 MFMAs x 32 on each SIMD): {cyc.mean()/steps:.0f}")
    for g in (0, 1):
        o, m, bx, by = (ph[:, g, k].mean() / steps for k in range(4))
        print(f"group {g}: per K-step cycles  phase1(other) {o:.0f}  wait@Y {by:.0f}  phase2(mma) {m:.0f}  wait@X {bx:.0f}  sum {o+m+bx+by:.0f}")
    # first round start spread and round structure
    st = np.sort(t[:, 0] - t0)
    print("start times (us) of workgroups 0,255,256,511,512:", [round(float(st[i]), 1) for i in (0, 255, 256, 511, 512) if i < len(st)])

if __name__ == "__main__":
    main()
