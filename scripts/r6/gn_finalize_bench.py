#!/usr/bin/env python
"""ivid_gn_finalize2 on the shapes the large model launches it with (batch 128): us per launch + checksum of the coefficients."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ivid_amd import _lib
lib = _lib.load()
st = torch.cuda.Stream(); sp = C.c_void_p(st.cuda_stream)
N = 128
# (C0, nchunks0, C1, nchunks1, HW)
for (c0, n0, c1, n1, hw) in ((256, 128, 0, 0, 16384), (256, 128, 256, 128, 16384), (256, 256, 0, 0, 16384), (512, 64, 0, 0, 4096), (256, 32, 0, 0, 4096),
                            (768, 16, 0, 0, 1024), (512, 16, 512, 16, 1024), (1024, 1, 0, 0, 64), (768, 8, 512, 16, 1024)):
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    p0 = torch.randn(N, n0, c0, 2, device="cuda", generator=g).abs()
    p1 = torch.randn(N, n1, c1, 2, device="cuda", generator=g).abs() if c1 else None
    c = c0 + c1
    gamma = torch.randn(c, device="cuda", generator=g); beta = torch.randn(c, device="cuda", generator=g)
    film = torch.randn(N, 40960, device="cuda", generator=g)
    ab = torch.zeros(N, c, 2, device="cuda")
    def go():
        _lib.check(lib.ivid_gn_finalize2(p0.data_ptr(), c0, n0, p1.data_ptr() if c1 else None, c1, n1, N, hw, 32, 1e-5, gamma.data_ptr(), beta.data_ptr(),
                                         film.data_ptr(), 40960, 1024, ab.data_ptr(), sp), "gn_finalize2")
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(50): go()
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"C {c0}x{n0}+{c1}x{n1} HW {hw}: {us:6.2f} us  checksum {ab.double().sum().item():.12g} {ab.double().abs().sum().item():.12g}", flush=True)
