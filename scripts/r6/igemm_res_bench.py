"""conv_igemm on the model's residual launches (lo planes + statistics as the compensated modes run them): ms per launch and a
checksum of every output (hi, lo, statistics) -- A/B aid for the epilogue forms.  Not part of the bench contract."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ivid_amd import _lib  # noqa: E402

SHAPES = [  # (H, Cin, Cout, taps, residual, lo planes)
    (32, 512, 512, 1, 1, 1), (16, 768, 768, 1, 1, 1), (8, 1024, 1024, 1, 1, 1), (16, 768, 768, 9, 1, 1), (8, 1024, 1024, 9, 1, 1),
    (16, 1792, 768, 9, 0, 1), (32, 512, 1536, 1, 0, 0), (16, 768, 768, 9, 1, 0), (32, 512, 512, 1, 1, 0), (64, 256, 256, 9, 0, 1),
]


def main():
    n = int(os.environ.get("N", "128"))
    reps = int(os.environ.get("REPS", "10"))
    lib = _lib.load()
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    for (h, cin, cout, taps, hasres, lo) in SHAPES:
        def rnd(*shape, seed):
            g = torch.Generator(device="cuda"); g.manual_seed(seed)
            return torch.randn(*shape, device="cuda", generator=g)
        x = rnd(n, h, h, cin, seed=1).half()
        w = (rnd(cout, taps * cin, seed=2) / (taps * cin) ** 0.5).half()
        b = rnd(cout, seed=3)
        rf = rnd(n, h, h, cout, seed=4)
        res = rf.half(); res_lo = (rf - res.float()).half()
        out = torch.empty(n, h, h, cout, device="cuda", dtype=torch.half); out_lo = torch.empty_like(out)
        blk = lib.ivid_conv2d_stats_block(n, h, h, cout, 0)
        stats = torch.zeros(n * h * h // blk, cout, 2, device="cuda")
        torch.cuda.synchronize()

        def launch():
            _lib.check(lib.ivid_conv2d_c(_lib.F16, x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                         out_lo.data_ptr() if lo else None, res.data_ptr() if hasres else None,
                                         res_lo.data_ptr() if (hasres and lo) else None, 1 if hasres else 0, 0, n, h, h, cout, taps, 0,
                                         stats.data_ptr(), sp), "conv2d_c")
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
        torch.cuda.synchronize()
        t = ms.value / reps
        ck = [float(out.double().sum()), float(out.double().abs().sum()), float(out_lo.double().abs().sum()) if lo else 0.0,
              float(stats.double().sum()), float(stats.double().abs().sum())]
        print(json.dumps(dict(h=h, cin=cin, cout=cout, taps=taps, res=hasres, lo=lo, ms=round(t, 4),
                              tflops=round(2.0 * n * h * h * cout * taps * cin / t / 1e9, 1), checksum=["%.10g" % c for c in ck])), flush=True)


if __name__ == "__main__":
    main()
