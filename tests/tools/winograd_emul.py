"""Winograd F(2x2, 3x3) with 16-bit operands: a measured yes / no (dev tool, TEST INFRASTRUCTURE; round 5).

3x3 convolutions are 91 % of the model's FLOPs and the chip is power-limited on MACs, so 2.25x fewer multiplications is the one
lever that could pass half of the nominal MFMA peak.  What it costs in deviation is decided here BEFORE any kernel: the product's
fp16 roundings are emulated inside the fp32 oracle forward (tests/tools/error_layers.py), and every 3x3 convolution OUTSIDE the
split-precision island is evaluated the way an fp16 Winograd kernel would -- transformed weights U = G g G^T and transformed input
tiles V = B^T d B each ROUNDED TO fp16 (they are the MFMA operands of the 16 position GEMMs), products and the channel sum in fp32,
output transform A^T M A in fp32 -- on the rows of the representative forward set.

    python tests/tools/winograd_emul.py --rows 0,1,2,3 --out profiles/r05_winograd_emulation.json
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import common as C  # noqa: E402
import error_layers as EL  # noqa: E402

G_ = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv3x3(a, w, b, rnd):
    """conv2d(a, w, b, padding=1) as F(2x2, 3x3); rnd(x) = the operand rounding applied to U and V (identity: exact up to fp32)."""
    n, c, h, ww = a.shape
    k = w.shape[0]
    assert h % 2 == 0 and ww % 2 == 0
    U = rnd(torch.einsum("ia,kcab,jb->ijkc", G_, w, G_)).reshape(16, k, c)
    out = torch.empty(n, k, h, ww)
    for i0 in range(0, n, 2):                               # two images at a time: V of a 512-channel 128^2 tensor is 270 MB per image
        d = F.pad(a[i0:i0 + 2], (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)          # [m, c, h/2, w/2, 4, 4]
        m = d.shape[0]
        V = rnd(torch.einsum("ia,mcyxab,jb->ijcmyx", BT, d, BT)).reshape(16, c, -1)
        M = torch.bmm(U, V).reshape(4, 4, k, m, h // 2, ww // 2)
        Y = torch.einsum("pi,ijkmyx,qj->mkypxq", AT, M, AT).reshape(m, k, h, ww)
        out[i0:i0 + m] = Y
    return out + b[None, :, None, None]


class WinoSites(EL.Sites):
    """The rounding classes of error_layers.Sites + Winograd evaluation of the selected 3x3 convolutions."""

    def __init__(self, pred, wino, dt=torch.float16):
        super().__init__(pred, dt)
        self.wino = wino          # site -> bool

    def conv(self, site, a, w, b, padding):
        if w.shape[-1] == 3 and self.wino(site) and a.shape[-1] % 2 == 0:
            on = self.pred is not None and self.pred("W", site)         # operands rounded where the base mode rounds them
            rnd = (lambda x: x.to(self.dt).float()) if on else (lambda x: x)
            return winograd_conv3x3(a, w, b, rnd)
        return super().conv(site, a, w, b, padding)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="0,1,2,3,12,13")
    ap.add_argument("--out", default="gpurun_out/winograd_emulation.json")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    args, seed = C.LARGE128, 4
    sd = {k: v.float() for k, v in C.synth_weights(args, seed).items()}
    ins = C.fwd_set_inputs(4, 128)
    xs, ts, cs, names = [], [], [], []
    for key, x, t, cls in ins:
        for br in ("c", "u"):
            xs.append(x); ts.append(t); cs.append(cls if br == "c" else -1); names.append(f"{key}_{br}")
    keep = [int(v) for v in a.rows.split(",")]
    xs, ts, cs, names = [[v[i] for i in keep] for v in (xs, ts, cs, names)]
    x, t, cl = torch.cat(xs), torch.tensor(ts), torch.tensor(cs)
    ref = EL.forward(sd, args, x, t, cl, EL.Sites())
    skip = lambda s: "skip_connection" in s
    island = ("input_blocks.1.0.", "input_blocks.2.0.")
    cs_pred = lambda k, s: k in "WAQ" and not skip(s)                                         # fp16cs = cx + split skips
    s_pred = lambda k, s: cs_pred(k, s) and not any(s.startswith(p) for p in island)          # fp16s  = + the island exact
    outside = lambda s: not any(s.startswith(p) for p in island)
    res = {"rows": names, "what": __doc__.split("\n\n")[1], "modes": {}}
    runs = {"exact arithmetic, Winograd everywhere (sanity: fp32 noise only)": WinoSites(None, lambda s: True),
            "fp16s (emulated, direct convolutions)": EL.Sites(s_pred),
            "fp16s + fp16 Winograd in every 3x3 convolution outside the island": WinoSites(s_pred, outside),
            "fp16cs (emulated, direct convolutions)": EL.Sites(cs_pred),
            "fp16cs + fp16 Winograd in every 3x3 convolution": WinoSites(cs_pred, lambda s: True)}
    for name, q in runs.items():
        r = EL.rows_rel(EL.forward(sd, args, x, t, cl, q), ref)
        res["modes"][name] = r
        print("%-75s max %.3e  rows %s" % (name, max(r), " ".join("%.2e" % v for v in r)), flush=True)
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
