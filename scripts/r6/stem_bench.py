"""Stem im2col (split form) at the bench shape: ms per launch and a checksum -- A/B aid."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ivid_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    for (bsrc, n, cin, s, kpad, split) in [(64, 128, 4, 128, 128, 1), (32, 64, 10, 128, 320, 1), (64, 128, 4, 128, 64, 0), (16, 32, 8, 256, 256, 1)]:
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        x = torch.randn(bsrc, cin, s, s, device="cuda", generator=g)
        out = torch.zeros(n, s, s, kpad, device="cuda", dtype=torch.half)
        fn = lib.ivid_stem_im2col_split if split else lib.ivid_stem_im2col
        torch.cuda.synchronize()

        def launch():
            _lib.check(fn(_lib.F16, x.data_ptr(), bsrc, n, cin, s, s, kpad, out.data_ptr(), sp), "stem")
        launch()
        torch.cuda.synchronize()
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.call("ivid_event_create", C.byref(e0)); _lib.call("ivid_event_create", C.byref(e1))
        _lib.call("ivid_event_record", e0, sp)
        for _ in range(10):
            launch()
        _lib.call("ivid_event_record", e1, sp)
        ms = C.c_float()
        _lib.call("ivid_event_elapsed_ms", e0, e1, C.byref(ms))
        torch.cuda.synchronize()
        t = ms.value / 10
        print(json.dumps(dict(bsrc=bsrc, n=n, cin=cin, s=s, kpad=kpad, split=split, ms=round(t, 4), gbps=round(out.numel() * 2 / t / 1e6, 1),
                              checksum=["%.10g" % float(out.double().sum()), "%.10g" % float(out.double().abs().sum())])), flush=True)


if __name__ == "__main__":
    main()
