#!/usr/bin/env python
"""ivid_gn_apply_c on the shapes the large model launches it with (batch 128, fp16 hi + lo planes): us per launch, GB/s."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ivid_amd import _lib
lib = _lib.load()
st = torch.cuda.Stream(); sp = C.c_void_p(st.cuda_stream)
N = 128
for (c0, c1, h) in ((1024, 0, 8), (1024, 1024, 8), (768, 0, 16), (768, 768, 16), (512, 0, 32), (256, 0, 64)):
    x0 = torch.randn(N, h, h, c0, device="cuda").half(); x0l = (torch.randn(N, h, h, c0, device="cuda") * 1e-3).half()
    x1 = torch.randn(N, h, h, c1, device="cuda").half() if c1 else None
    x1l = (torch.randn(N, h, h, c1, device="cuda") * 1e-3).half() if c1 else None
    ab = torch.rand(N, c0 + c1, 2, device="cuda") + 0.5
    out = torch.empty(N, h, h, c0 + c1, device="cuda", dtype=torch.float16)
    def go():
        _lib.check(lib.ivid_gn_apply_c(_lib.F16, x0.data_ptr(), x0l.data_ptr(), c0, x1.data_ptr() if c1 else None, x1l.data_ptr() if c1 else None, c1,
                                       ab.data_ptr(), out.data_ptr(), N, h, h, 0, 1, sp), "gn_apply_c")
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): go()
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    byt = N * h * h * (c0 + c1) * 2 * 3
    ref = torch.nn.functional.silu((torch.cat([x0, x1], -1) if c1 else x0).float() + (torch.cat([x0l, x1l], -1) if c1 else x0l).float())  # placeholder finite check
    print(f"{h:3d}^2 C {c0}+{c1}: {us:7.1f} us  {byt / us / 1e3:7.1f} GB/s  checksum {out.float().sum().item():.6g}", flush=True)
