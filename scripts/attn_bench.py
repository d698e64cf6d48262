#!/usr/bin/env python
"""Microbenchmark of ivid_attention on the three attention shapes of the large UNet at stacked batch 128 (tuning aid)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ivid_amd import _lib

lib = _lib.load()
stream = torch.cuda.Stream()
sp = C.c_void_p(stream.cuda_stream)
n = int(os.environ.get("N", "128"))
for (t, heads) in [(1024, 8), (256, 12), (64, 16)]:
    c = heads * 64
    qkv = torch.randn(n, t, 3 * c, device="cuda").bfloat16()
    out = torch.empty(n, t, c, device="cuda", dtype=torch.bfloat16)
    f = lambda: _lib.check(lib.ivid_attention(_lib.BF16, qkv.data_ptr(), out.data_ptr(), n, t, heads, sp), "attn")
    f(); torch.cuda.synchronize()
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.call("ivid_event_create", C.byref(e0)); _lib.call("ivid_event_create", C.byref(e1))
    _lib.call("ivid_event_record", e0, sp)
    for _ in range(10):
        f()
    _lib.call("ivid_event_record", e1, sp)
    ms = C.c_float(); _lib.call("ivid_event_elapsed_ms", e0, e1, C.byref(ms))
    ms = ms.value / 10
    print(json.dumps(dict(T=t, heads=heads, ms=round(ms, 4), tflops=round(4.0 * heads * t * t * 64 * n / ms / 1e9, 1))), flush=True)
