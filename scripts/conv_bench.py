#!/usr/bin/env python
"""Kernel-level microbenchmark of ivid_conv2d on the layer shapes of the large UNet at stacked batch 128 (HIP events
on the launch stream).  Prints one line per (shape, tile_cfg): ms, TFLOP/s.  Tuning aid, not part of the bench contract."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ivid_amd import _lib  # noqa: E402

SHAPES = [  # (H, Cin, Cout, taps)
    (128, 256, 256, 9), (128, 512, 256, 9), (128, 512, 256, 1), (64, 256, 256, 9), (64, 768, 256, 9), (32, 512, 512, 9),
    (32, 1280, 512, 9), (32, 512, 1536, 1), (16, 768, 768, 9), (16, 1792, 768, 9), (8, 1024, 1024, 9), (8, 2048, 1024, 9),
    (16, 1024, 1024, 9), (16, 768, 2304, 1), (8, 1024, 3072, 1), (32, 512, 512, 1), (16, 768, 768, 1), (8, 1024, 1024, 1),   # 12..17
    (128, 128, 128, 9), (128, 256, 128, 9), (64, 128, 128, 9),   # 18..20: the small / SR models' 128-channel layers
]


def main():
    n = int(os.environ.get("N", "128"))
    dtype = _lib.BF16 if os.environ.get("DTYPE", "bf16") == "bf16" else _lib.F32
    tdt = torch.bfloat16 if dtype == _lib.BF16 else torch.float32
    cfgs = [int(c) for c in os.environ.get("CFGS", "-1,1,2").split(",")]   # -1 = fused conv3x3_gn kernel
    reps = int(os.environ.get("REPS", "5"))
    lib = _lib.load()
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    rows = []
    only = os.environ.get("SHAPES_ONLY")
    shapes = [SHAPES[int(i)] for i in only.split(",")] if only else SHAPES
    for (h, cin, cout, taps) in shapes:
        x = torch.randn(n, h, h, cin, device="cuda").to(tdt)
        if os.environ.get("ZERO") == "1":     # zero operands: no data toggling in the MFMA datapath (power / clock probe)
            x.zero_()
        w = (torch.randn(cout, taps * cin, device="cuda") / (taps * cin) ** 0.5).to(tdt)
        if os.environ.get("ZERO") == "1":
            w.zero_()
        b = torch.randn(cout, device="cuda")
        out = torch.empty(n, h, h, cout, device="cuda", dtype=tdt)
        res = torch.randn(n, h, h, cout, device="cuda").to(tdt) if os.environ.get("RES") == "1" else None   # same-size residual
        rp, rm = (res.data_ptr(), 1) if res is not None else (None, 0)
        flop = 2.0 * n * h * h * cout * taps * cin
        ab = torch.rand(n, cin, 2, device="cuda") + 0.5
        for cfg in cfgs:
            if cfg == -1 and (taps != 9 or h < 32):
                continue

            def launch():
                if cfg == -1:   # the fused GN-apply + SiLU + conv3x3 kernel
                    _lib.check(lib.ivid_conv3x3_gn(dtype, x.data_ptr(), cin, None, 0, ab.data_ptr(), 0, w.data_ptr(), b.data_ptr(),
                                                   out.data_ptr(), rp, rm, n, h, h, cout, None, sp), "conv3x3_gn")
                else:
                    _lib.check(lib.ivid_conv2d(dtype, x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), out.data_ptr(), rp, rm, 0,
                                               n, h, h, cout, taps, cfg, None, sp), "conv")
            launch()
            torch.cuda.synchronize()
            e0, e1 = C.c_void_p(), C.c_void_p()
            _lib.call("ivid_event_create", C.byref(e0)); _lib.call("ivid_event_create", C.byref(e1))
            _lib.call("ivid_event_record", e0, sp)
            for _ in range(reps):
                launch()
            _lib.call("ivid_event_record", e1, sp)
            ms = C.c_float()
            _lib.call("ivid_event_elapsed_ms", e0, e1, C.byref(ms))
            t = ms.value / reps
            rows.append(dict(h=h, cin=cin, cout=cout, taps=taps, cfg=cfg, ms=round(t, 4), tflops=round(flop / t / 1e9, 1)))
            print(json.dumps(rows[-1]), flush=True)
        del x, w, out
    if os.environ.get("OUT"):
        json.dump(rows, open(os.environ["OUT"], "w"), indent=0)


if __name__ == "__main__":
    main()
