#!/usr/bin/env python
"""Runs the timing-only variants of scripts/dev/kstep_ablation_build.py on the 128^2 256->256 and 512->256 layers (plain fp16 form,
batch 128) and prints, per variant, the main loop's time per K-step (phase stamps, median over the workgroups) and the launch time.
    python scripts/dev/kstep_ablation.py            (GPU box; one subprocess per library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["base", "nodma", "nohalo", "bare", "bare1", "free"]

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import ctypes as C
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from ivid_amd import _lib
    lib = _lib.load()
    lib.ivid_dev_timeline.argtypes = [C.c_void_p]
    lib.ivid_dev_timeline.restype = None
    n, h = 128, 128
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    tdt = torch.float16
    for (c0, c1, cout) in ((256, 0, 256), (256, 256, 256)):
        cin = c0 + c1
        x0 = torch.randn(n, h, h, c0, device="cuda").to(tdt)
        x1 = torch.randn(n, h, h, c1, device="cuda").to(tdt) if c1 else None
        w = (torch.randn(cout, 9 * cin, device="cuda") / (9 * cin) ** 0.5).to(tdt)
        b = torch.randn(cout, device="cuda")
        ab = torch.rand(n, cin, 2, device="cuda") + 0.5
        out = torch.empty(n, h, h, cout, device="cuda", dtype=tdt)
        ntiles = n * (h // 8) * (h // 32)
        stats_buf = torch.zeros(n * (h // 4) * (h // 32) * cout * 2, device="cuda")
        dbg = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
        lib.ivid_dev_timeline(C.c_void_p(dbg.data_ptr()))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(4):
            if it == 2:
                ev0.record(stream)
            _lib.check(lib.ivid_conv3x3_gn_skip_c(_lib.F16, x0.data_ptr(), None, c0, x1.data_ptr() if c1 else None, None, c1, ab.data_ptr(), 0,
                                                  w.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, 0,
                                                  n, h, h, cout, stats_buf.data_ptr(), None, 0, None, 0, None, sp), "launch")
        ev1.record(stream)
        torch.cuda.synchronize()
        lib.ivid_dev_timeline(None)
        t = dbg.cpu().numpy().reshape(ntiles, 8)[:, :7].astype(np.float64) / 100.0
        steps = 9 * cin * 2 // 128
        main = t[:, 4] - t[:, 3]
        print(json.dumps({"variant": sys.argv[2], "layer": f"{cin}->{cout}", "launch_ms": round(ev0.elapsed_time(ev1) / 2, 3),
                          "main_loop_us": round(float(np.median(main)), 2), "us_per_kstep": round(float(np.median(main)) / steps, 3),
                          "p10_p90_us_per_kstep": [round(float(np.percentile(main, q)) / steps, 3) for q in (10, 90)],
                          "prologue_us": round(float(np.median(t[:, 3] - t[:, 0])), 2), "epilogue_us": round(float(np.median(t[:, 6] - t[:, 5])), 2),
                          "workgroup_us": round(float(np.median(t[:, 6] - t[:, 0])), 2)}), flush=True)
    sys.exit(0)

for name in NAMES:
    lib = os.path.join(ROOT, "ab", f"libivid_abl_{name}.so")
    if not os.path.exists(lib):
        print(json.dumps({"variant": name, "error": "not built"}))
        continue
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], env=dict(os.environ, IVID_HIP_LIB=lib),
                       capture_output=True, text=True, timeout=300)
    sys.stdout.write(r.stdout)
    if r.returncode != 0:
        print(json.dumps({"variant": name, "error": r.stderr[-400:]}))
