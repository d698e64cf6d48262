#!/usr/bin/env python
"""Phase timeline of conv3x3_fused_kernel (development aid; needs the -DIVID_DEV_TIMELINE build, scripts/dev/build_timeline.sh):

    IVID_HIP_LIB=$PWD/ab/libivid_timeline.so python scripts/dev/fused_timeline.py [OUT.json]

Thread 0 of every workgroup stamps the 100 MHz real-time counter at: 0 entry, 1 after the start skew, 2 prologue loads landed,
3 first halo image transformed and stored, 4 main loop done, 5 skip phase done, 6 epilogue stores retired; slot 7 = HW_ID /
XCC_ID.  Printed per shape: mean / median / p90 of every phase in us, the gap between consecutive workgroups of one CU,
tiles per CU, and how tightly the CUs stay in lockstep (spread of the epilogue start inside a round)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivid_amd import _lib  # noqa: E402

SHAPES = [  # (H, C0, C1, Cout, res_mode, skipC)
    (128, 256, 0, 256, 0, 0), (128, 256, 0, 256, 1, 0), (128, 256, 256, 256, 0, 0), (128, 256, 0, 256, 0, 512),
    (64, 512, 0, 512, 0, 0), (32, 768, 0, 768, 0, 0),
]


def stats(x):
    x = np.asarray(x, dtype=np.float64)
    return dict(mean=round(float(x.mean()), 2), med=round(float(np.median(x)), 2), p90=round(float(np.percentile(x, 90)), 2),
                max=round(float(x.max()), 2))


def main():
    lib = _lib.load()
    lib.ivid_dev_timeline.argtypes = [C.c_void_p]
    lib.ivid_dev_timeline.restype = None
    n = int(os.environ.get("N", "128"))
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    out_rows = []
    for (h, c0, c1, cout, rm, skc) in SHAPES:
        tdt = torch.bfloat16
        x0 = torch.randn(n, h, h, c0, device="cuda").to(tdt)
        x1 = torch.randn(n, h, h, c1, device="cuda").to(tdt) if c1 else None
        cin = c0 + c1
        w = (torch.randn(cout, 9 * cin, device="cuda") / (9 * cin) ** 0.5).to(tdt)
        b = torch.randn(cout, device="cuda")
        ab = torch.rand(n, cin, 2, device="cuda") + 0.5
        out = torch.empty(n, h, h, cout, device="cuda", dtype=tdt)
        res = torch.randn(n, h, h, cout, device="cuda").to(tdt) if rm else None
        sk = torch.randn(n, h, h, skc, device="cuda").to(tdt) if skc else None
        skw = (torch.randn(cout, skc, device="cuda") / max(skc, 1) ** 0.5).to(tdt) if skc else None
        ntiles = n * (h // 8) * (h // 32) * ((cout + 255) // 256)
        dbg = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
        stats_buf = torch.zeros(n * (h // 4) * (h // 32) * cout * 2, device="cuda")
        lib.ivid_dev_timeline(C.c_void_p(dbg.data_ptr()))

        def launch():
            _lib.check(lib.ivid_conv3x3_gn_skip(_lib.BF16, x0.data_ptr(), c0, x1.data_ptr() if c1 else None, c1, ab.data_ptr(), 0,
                                                w.data_ptr(), b.data_ptr(), out.data_ptr(), res.data_ptr() if rm else None, rm, n, h, h,
                                                cout, stats_buf.data_ptr(), sk.data_ptr() if skc else None, skc, None, 0,
                                                skw.data_ptr() if skc else None, sp), "conv3x3_gn_skip")
        for _ in range(2):
            launch()
        torch.cuda.synchronize()
        lib.ivid_dev_timeline(None)
        t = dbg.cpu().numpy().reshape(ntiles, 8)
        ts = t[:, :7].astype(np.float64) / 100.0           # us
        t0 = ts[:, 0].min()
        ts -= t0
        hw = t[:, 7].astype(np.uint64)
        cu = ((hw >> np.uint64(32)) & np.uint64(0xF)) * np.uint64(256) + ((hw >> np.uint64(8)) & np.uint64(0xFF))
        names = ["skew", "prologue loads", "first halo transform", "main loop", "skip phase", "epilogue"]
        row = dict(shape=dict(h=h, c0=c0, c1=c1, cout=cout, res_mode=rm, skip=skc, tiles=ntiles),
                   kernel_us=round(float(ts[:, 6].max()), 1), cus_seen=int(len(np.unique(cu))))
        for k, nm in enumerate(names):
            row[nm] = stats(ts[:, k + 1] - ts[:, k])
        row["workgroup total"] = stats(ts[:, 6] - ts[:, 0])
        gaps, per_cu = [], []
        for c in np.unique(cu):
            idx = np.where(cu == c)[0]
            o = idx[np.argsort(ts[idx, 0])]
            per_cu.append(len(o))
            gaps.extend((ts[o[1:], 0] - ts[o[:-1], 6]).tolist())
        row["gap between workgroups of a CU"] = stats(gaps) if gaps else None
        row["tiles per CU"] = dict(min=int(min(per_cu)), max=int(max(per_cu)))
        # lockstep: order workgroups by start; round r = the r-th workgroup of each CU; spread of its epilogue start
        spreads = []
        for r in range(min(per_cu)):
            st = []
            for c in np.unique(cu):
                idx = np.where(cu == c)[0]
                o = idx[np.argsort(ts[idx, 0])]
                st.append(ts[o[r], 5])
            spreads.append(float(np.percentile(st, 90) - np.percentile(st, 10)))
        row["p10-p90 spread of the epilogue start per round (first, middle, last)"] = [round(spreads[0], 1), round(spreads[len(spreads) // 2], 1),
                                                                                      round(spreads[-1], 1)]
        print(json.dumps(row), flush=True)
        out_rows.append(row)
    if len(sys.argv) > 1:
        json.dump(out_rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
