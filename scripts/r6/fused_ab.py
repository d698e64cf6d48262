#!/usr/bin/env python
"""Phase timeline + launch time of conv3x3_fused_kernel for several builds of the library (scripts/r6/build_exp.py ... --tl) on the
128^2 layers that dominate a step, in the forms the headline ladder runs (plain fp16 / + lo planes of the output / + lo planes of
the inputs / + same-size residual).  One subprocess per library; prints one JSON line per (library, layer, form) incl. a checksum of
the output (variants that must be bit-identical to the base are checked that way).
    python scripts/r6/fused_ab.py ab/libivid_x0_tl.so ab/libivid_x1_tl.so ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    import ctypes as C
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from ivid_amd import _lib
    lib = _lib.load()
    has_tl = hasattr(lib, "ivid_dev_timeline")
    try:
        lib.ivid_dev_timeline.argtypes = [C.c_void_p]
        lib.ivid_dev_timeline.restype = None
    except AttributeError:
        has_tl = False
    n, h = int(os.environ.get("N", "128")), 128
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    tdt = torch.float16
    forms = os.environ.get("FORMS", "plain,LOIN,LOINres").split(",")
    layers = [tuple(int(v) for v in s.split(":")) for s in os.environ.get("LAYERS", "256:0:256,256:256:256").split(",")]
    torch.manual_seed(1)
    _seed = [100]

    def fresh():   # one generator per tensor (a single generator walked through several 2^29-element draws handed back NaNs)
        _seed[0] += 1
        gg = torch.Generator(device="cuda"); gg.manual_seed(_seed[0])
        return gg
    for (c0, c1, cout) in layers:
        cin = c0 + c1
        def rn(*s):
            v = torch.randn(*s, device="cuda", generator=fresh())
            assert torch.isfinite(v).all().item()
            return v
        x0 = rn(n, h, h, c0).to(tdt); x0l = (rn(n, h, h, c0) * 1e-3).to(tdt)
        x1 = rn(n, h, h, c1).to(tdt) if c1 else None
        x1l = (rn(n, h, h, c1) * 1e-3).to(tdt) if c1 else None
        w = (rn(cout, 9 * cin) / (9 * cin) ** 0.5).to(tdt)
        b = rn(cout)
        ab = torch.rand(n, cin, 2, device="cuda", generator=fresh()) + 0.5
        res = rn(n, h, h, cout).to(tdt); resl = (rn(n, h, h, cout) * 1e-3).to(tdt)
        out = torch.empty(n, h, h, cout, device="cuda", dtype=tdt); outl = torch.empty_like(out)
        if any(f.endswith("skip") or f.endswith("sks") for f in forms):     # skip sources: cat(256, 256) -> cout (the 128^2 decoder blocks)
            sk0, sk1 = rn(n, h, h, 256).to(tdt), rn(n, h, h, 256).to(tdt)
            sk0l, sk1l = (rn(n, h, h, 256) * 1e-3).to(tdt), (rn(n, h, h, 256) * 1e-3).to(tdt)
            skw = (rn(cout, 512) / 512 ** 0.5).to(tdt); skwl = (rn(cout, 512) * 1e-4).to(tdt)
        ntiles = n * (h // 8) * (h // 32)
        stats_buf = torch.zeros(n * (h // 4) * (h // 32) * cout * 2, device="cuda")
        for tag in forms:
            lo_in = tag.startswith("LOIN"); lo_out = tag.startswith("LO"); r = tag.endswith("res")
            sk = tag.endswith("skip") or tag.endswith("sks"); sks = tag.endswith("sks")
            dbg = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
            if has_tl:
                lib.ivid_dev_timeline(C.c_void_p(dbg.data_ptr()))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 4
            for it in range(2 + reps):
                if it == 2:
                    ev0.record(stream)
                if sk:    # the ResBlock's 1x1 skip_connection on cat(sk0, sk1) folded in: plain (hi planes) or split (SKS: + lo planes, + lo weights)
                    _lib.check(lib.ivid_conv3x3_gn_skip_s(_lib.F16, x0.data_ptr(), x0l.data_ptr() if lo_in else None, c0,
                                                          x1.data_ptr() if c1 else None, x1l.data_ptr() if (c1 and lo_in) else None, c1, ab.data_ptr(), 0,
                                                          w.data_ptr(), b.data_ptr(), out.data_ptr(), outl.data_ptr() if lo_out else None,
                                                          None, None, 0, n, h, h, cout, stats_buf.data_ptr(),
                                                          sk0.data_ptr(), sk0.shape[-1], sk1.data_ptr(), sk1.shape[-1], skw.data_ptr(),
                                                          sk0l.data_ptr() if sks else None, sk1l.data_ptr() if sks else None, skwl.data_ptr() if sks else None, sp), "launch")
                else:
                    _lib.check(lib.ivid_conv3x3_gn_skip_c(_lib.F16, x0.data_ptr(), x0l.data_ptr() if lo_in else None, c0,
                                                      x1.data_ptr() if c1 else None, x1l.data_ptr() if (c1 and lo_in) else None, c1, ab.data_ptr(), 0,
                                                      w.data_ptr(), b.data_ptr(), out.data_ptr(), outl.data_ptr() if lo_out else None,
                                                      res.data_ptr() if r else None, resl.data_ptr() if (r and lo_out) else None, 1 if r else 0,
                                                      n, h, h, cout, stats_buf.data_ptr(), None, 0, None, 0, None, sp), "launch")
            ev1.record(stream)
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / reps
            if os.environ.get("DEBUG_FINITE"):
                for nm, tt in (("x0", x0), ("x0l", x0l), ("x1", x1), ("x1l", x1l), ("w", w), ("b", b), ("ab", ab), ("res", res), ("resl", resl), ("out", out), ("stats", stats_buf)):
                    if tt is not None:
                        nf = (~torch.isfinite(tt.float())).sum().item()
                        if nf:
                            idx = (~torch.isfinite(tt.float())).nonzero()
                            print("NONFINITE", nm, nf, idx[:3].tolist(), idx[-3:].tolist(), flush=True)
            row = {"lib": sys.argv[2], "layer": f"{cin}->{cout}", "form": tag, "ms": round(ms, 4),
                   "tflops": round(2.0 * n * h * h * cout * 9 * cin / ms / 1e9, 1),
                   "sum": float(out.float().sum().item()), "sumsq": float((out.float() ** 2).sum().item()),
                   "stats_sum": float(stats_buf.sum().item())}
            if has_tl:
                lib.ivid_dev_timeline(None)
                t = dbg.cpu().numpy().reshape(ntiles, 8)[:, :7].astype(np.float64) / 100.0
                steps = 9 * cin * 2 // 128
                row.update({"main_us": round(float(np.median(t[:, 4] - t[:, 3])), 2),
                            "us_per_kstep": round(float(np.median(t[:, 4] - t[:, 3])) / steps, 3),
                            "prologue_us": round(float(np.median(t[:, 3] - t[:, 0])), 2),
                            "epilogue_us": round(float(np.median(t[:, 6] - t[:, 5])), 2),
                            "wg_us": round(float(np.median(t[:, 6] - t[:, 0])), 2)})
            print(json.dumps(row), flush=True)
    sys.exit(0)

for lib in sys.argv[1:]:
    p = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
    if not os.path.exists(p):
        print(json.dumps({"lib": lib, "error": "not built"}))
        continue
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", os.path.basename(lib)], env=dict(os.environ, IVID_HIP_LIB=p),
                       capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout)
    if r.returncode != 0:
        print(json.dumps({"lib": lib, "error": r.stderr[-600:]}))
