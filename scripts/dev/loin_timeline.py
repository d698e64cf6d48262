#!/usr/bin/env python
"""Where does the lo-plane read of the fused kernel's halo transform (LOIN instantiation, precision modes fp16cx / fp16s) cost its
time?  Phase timeline (scripts/dev/build_timeline.sh build) of the 128^2 512->256 and 256->256 layers in three forms: plain fp16,
+ lo planes of output (LO), + lo planes of the inputs (LOIN).
    IVID_HIP_LIB=$PWD/ab/libivid_timeline.so python scripts/dev/loin_timeline.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivid_amd import _lib  # noqa: E402

lib = _lib.load()
lib.ivid_dev_timeline.argtypes = [C.c_void_p]
lib.ivid_dev_timeline.restype = None
n, h = 128, 128
stream = torch.cuda.Stream()
sp = C.c_void_p(stream.cuda_stream)
tdt = torch.float16
for (c0, c1, cout) in ((256, 256, 256), (256, 0, 256)):
    cin = c0 + c1
    x0 = torch.randn(n, h, h, c0, device="cuda").to(tdt); x0l = (torch.randn(n, h, h, c0, device="cuda") * 1e-3).to(tdt)
    x1 = torch.randn(n, h, h, c1, device="cuda").to(tdt) if c1 else None
    x1l = (torch.randn(n, h, h, c1, device="cuda") * 1e-3).to(tdt) if c1 else None
    w = (torch.randn(cout, 9 * cin, device="cuda") / (9 * cin) ** 0.5).to(tdt)
    b = torch.randn(cout, device="cuda")
    ab = torch.rand(n, cin, 2, device="cuda") + 0.5
    out = torch.empty(n, h, h, cout, device="cuda", dtype=tdt); outl = torch.empty_like(out)
    ntiles = n * (h // 8) * (h // 32)
    stats_buf = torch.zeros(n * (h // 4) * (h // 32) * cout * 2, device="cuda")
    for tag, lo_in, lo_out in (("plain", False, False), ("LO", False, True), ("LOIN", True, True)):
        dbg = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
        lib.ivid_dev_timeline(C.c_void_p(dbg.data_ptr()))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(3):
            if it == 2:
                ev0.record(stream)
            _lib.check(lib.ivid_conv3x3_gn_skip_c(_lib.F16, x0.data_ptr(), x0l.data_ptr() if lo_in else None, c0,
                                                  x1.data_ptr() if c1 else None, x1l.data_ptr() if (c1 and lo_in) else None, c1, ab.data_ptr(), 0,
                                                  w.data_ptr(), b.data_ptr(), out.data_ptr(), outl.data_ptr() if lo_out else None, None, None, 0,
                                                  n, h, h, cout, stats_buf.data_ptr(), None, 0, None, 0, None, sp), "launch")
        ev1.record(stream)
        torch.cuda.synchronize()
        lib.ivid_dev_timeline(None)
        t = dbg.cpu().numpy().reshape(ntiles, 8)
        ts = t[:, :7].astype(np.float64) / 100.0
        names = ["skew", "prologue loads", "first halo transform", "main loop", "skip phase", "epilogue"]
        row = {"shape": f"{cin}->{cout}", "form": tag, "kernel_ms_events": round(ev0.elapsed_time(ev1), 3)}
        for k, nm in enumerate(names):
            d = ts[:, k + 1] - ts[:, k]
            row[nm] = round(float(np.median(d)), 2)
        row["workgroup total (median us)"] = round(float(np.median(ts[:, 6] - ts[:, 0])), 2)
        print(json.dumps(row), flush=True)
