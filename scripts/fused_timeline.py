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
    dbg = torch.zeros(ntiles * 5, dtype=torch.int64, device="cuda")
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        _lib.check(lib.ivid_conv3x3_gn(_lib.BF16, x.data_ptr(), cin, None, 0, ab.data_ptr(), 0, w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                       None, 0, n, h, h, cout, dbg.data_ptr(), sp), "fused")
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 5)
    t = d[:, :4].astype(np.float64) * 0.01  # us
    hw = d[:, 4]
    t0 = t[:, 0].min()
    pro, main_, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"tiles {ntiles}  kernel span {t[:,3].max()-t0:.1f} us")
    for name, v in (("prologue", pro), ("main", main_), ("epilogue", epi), ("total", t[:, 3] - t[:, 0])):
        print(f"{name:9s} mean {v.mean():7.2f}  p10 {np.percentile(v,10):7.2f}  p50 {np.percentile(v,50):7.2f}  p90 {np.percentile(v,90):7.2f} us")
    # gaps between consecutive workgroups on one CU: key = hw id without wave/simd bits (bits 0-5), keep cu/sh/se/xcc
    key = hw >> 8
    gaps = []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        o = idx[np.argsort(t[idx, 0])]
        gaps.extend((t[o[1:], 0] - t[o[:-1], 3]).tolist())
    gaps = np.array(gaps)
    print(f"distinct CU keys {len(np.unique(key))}; gap end->next start: mean {gaps.mean():.2f} p50 {np.percentile(gaps,50):.2f} p90 {np.percentile(gaps,90):.2f} us")
    # first round start spread and round structure
    st = np.sort(t[:, 0] - t0)
    print("start times (us) of workgroups 0,255,256,511,512:", [round(float(st[i]), 1) for i in (0, 255, 256, 511, 512) if i < len(st)])

if __name__ == "__main__":
    main()
