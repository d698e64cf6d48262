"""Where does the 16-bit mode's deviation from the fp32 reference come from?  (dev tool, CPU, TEST INFRASTRUCTURE)

Emulates the product's fp16 (or bf16) storage / operand roundings inside the fp32 oracle forward, one class of rounding
at a time, on the large-128 synthetic checkpoint of tests/golden/large128_fwd.npz:
    W  conv / qkv / proj weights rounded to the 16-bit type            (plan.PackedWeights)
    A  conv INPUT activations rounded (halo transform / gn_apply output -> MFMA operand)
    T  trunk storage: stem / ResBlock / attention-block OUTPUTS rounded (what the next block and the skip stash read)
    H  inner storage: the in_layers conv output h1 rounded
    B  compensated trunk (fp16 hi + lo planes): branch inputs (GroupNorm -> conv, 1x1 skip conv, attention norm) see the
       rounded hi plane, residual adds and the stored trunk are exact
    N  the attention block's GroupNorm reads the hi plane alone (gn_apply without the lo plane)
    Q  attention internals: normalised input, qkv, softmax probabilities, attention output rounded
Prints rel-L2 vs the committed reference output for: all on, each one alone, each one removed.

    python tests/tools/error_budget.py [--dtype fp16|bf16] [--model large|small|mini]
"""
import argparse
import itertools
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import common as C  # noqa: E402
from oracle import adm_oracle as O  # noqa: E402


class Q:
    def __init__(self, on, dt, sides=None):
        self.on, self.dt, self.sides, self.side = set(on), dt, sides, None
        self.only, self.blk, self.invert = None, None, False

    def __call__(self, kind, x):
        if self.sides is not None and self.side not in self.sides:
            return x
        if self.only is not None and (self.blk in self.only) == self.invert:
            return x
        return x.to(self.dt).float() if kind in self.on else x


def resblock(sd, p, x, emb, mode, groups, q):
    q.side = x.shape[-1] * (2 if mode == "up" else 1) // (2 if mode == "down" else 1)
    q.blk = p
    xb = q("B", x)      # what the branch reads when the trunk is carried as fp16 hi + lo and consumers see only hi
    if mode == "same" and q.side >= 32:
        xb = q("F", xb)  # ... only where the fused halo kernel is the consumer (gn_apply / residual epilogues read hi + lo)
    h = F.silu(O.gn32(xb, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], groups))
    if mode == "up":
        h = F.interpolate(q("A", h), scale_factor=2, mode="nearest")      # gn_apply writes the activated source (16-bit)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif mode == "down":
        h = F.avg_pool2d(h, 2)
        x = F.avg_pool2d(x, 2)
    h = F.conv2d(q("A", h), q("W", sd[p + ".in_layers.2.weight"]), sd[p + ".in_layers.2.bias"], padding=1)
    h = q("H", h)
    eo = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    scale, shift = eo[:, :, None, None].chunk(2, dim=1)
    h = O.gn32(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], groups) * (1 + scale) + shift
    h = F.conv2d(q("A", F.silu(h)), q("W", sd[p + ".out_layers.3.weight"]), sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(q("A", xb if mode == "same" else x), q("W", sd[p + ".skip_connection.weight"]), sd[p + ".skip_connection.bias"])
    return q("T", x + h)


def attnblock(sd, p, x, head_channels, groups, q):
    b, c, hh, ww = x.shape
    q.side = hh
    q.blk = p
    xf = x.reshape(b, c, -1)
    xn = q("Q", O.gn32(q("N", q("B", xf)), sd[p + ".norm.weight"], sd[p + ".norm.bias"], groups))
    qkv = q("Q", F.conv1d(xn, q("W", sd[p + ".qkv.weight"]), sd[p + ".qkv.bias"]))
    heads = c // head_channels
    bb, width, t = qkv.shape
    ch = width // (3 * heads)
    qq, k, v = qkv.reshape(bb * heads, 3 * ch, t).split(ch, dim=1)
    w = torch.softmax(torch.einsum("bct,bcs->bts", qq, k) / ch ** 0.5, dim=-1)
    a = q("Q", torch.einsum("bts,bcs->bct", q("Q", w), v).reshape(bb, -1, t))
    a = F.conv1d(a, q("W", sd[p + ".proj_out.weight"]), sd[p + ".proj_out.bias"])
    return q("T", (xf + a).reshape(b, c, hh, ww))


def stage(sd, prefix, h, emb, args, modes, q):
    j = 0
    while True:
        p = f"{prefix}.{j}"
        if p + ".in_layers.0.weight" in sd:
            h = resblock(sd, p, h, emb, modes.get(p, "same"), args["num_groups"], q)
        elif p + ".qkv.weight" in sd:
            h = attnblock(sd, p, h, args["num_head_channels"], args["num_groups"], q)
        else:
            return h
        j += 1


@torch.no_grad()
def forward(sd, args, x, t, classes, q):
    has_null = bool(args.get("has_null_class", False)) and args.get("num_classes") is not None
    emb = O.embedding(sd, t, classes, has_null)
    modes = O._updown_modes(sd, args)
    hs = []
    q.side, q.blk = x.shape[-1], "stem"
    h = q("T", F.conv2d(q("A", x), q("W", sd["input_blocks.0.0.weight"]), sd["input_blocks.0.0.bias"], padding=1))
    hs.append(h)
    i = 1
    while f"input_blocks.{i}.0.in_layers.0.weight" in sd:
        h = stage(sd, f"input_blocks.{i}", h, emb, args, modes, q)
        hs.append(h)
        i += 1
    h = stage(sd, "middle_block", h, emb, args, modes, q)
    i = 0
    while f"output_blocks.{i}.0.in_layers.0.weight" in sd:
        h = torch.cat([h, hs.pop()], dim=1)
        h = stage(sd, f"output_blocks.{i}", h, emb, args, modes, q)
        i += 1
    q.side, q.blk = x.shape[-1], "head"
    h = F.silu(O.gn32(h, sd["out.0.weight"], sd["out.0.bias"], args["num_groups"]))
    return F.conv2d(q("A", h), q("W", sd["out.2.weight"]), sd["out.2.bias"], padding=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--model", default="large")
    ap.add_argument("--sets", default="")
    ap.add_argument("--only", default="", help="apply the roundings only in these blocks (state_dict prefixes, stem, head)")
    ap.add_argument("--except", dest="exc", default="", help="apply the roundings everywhere but in these blocks")
    ap.add_argument("--sides", default="", help="apply the roundings only in layers whose OUTPUT side is in this list")
    a = ap.parse_args()
    dt = dict(fp16=torch.float16, bf16=torch.bfloat16)[a.dtype]
    torch.set_num_threads(os.cpu_count())
    if a.model == "large":
        args, seed, t, cls, gold = C.LARGE128, 4, 999, [7], "large128_fwd"
    elif a.model == "small":
        args, seed, t, cls, gold = C.SMALL128, 3, 500, None, "small128_fwd"
    else:
        args, seed, t, cls, gold = C.MINI, 0, 37, [3, -1], "mini_fwd"
    sd = {k: v.float() for k, v in C.synth_weights(args, seed).items()}
    S = args["image_size"]
    g = C.load_golden(gold)
    b = g["eps"].shape[0]
    x = C.seeded_randn(100 + seed, b, args["in_channels"], S, S)
    tt = torch.full((b,), t, dtype=torch.long)
    cl = torch.tensor(cls) if cls is not None else None
    ref = torch.from_numpy(g["eps"])
    kinds = "WATHQ"
    sets = a.sets.split(",") if a.sets else ([""] + [kinds] + list(kinds) + [kinds.replace(k, "") for k in kinds])
    for s in sets:
        q = Q(s, dt, [int(v) for v in a.sides.split(",")] if a.sides else None)
        if a.only or a.exc:
            q.only, q.invert = set((a.only or a.exc).split(",")), bool(a.exc)
        out = forward(sd, args, x, tt, cl, q)
        print(f"{a.dtype} rounding of [{s:5s}]: rel-L2 vs reference = {C.rel_l2(out, ref):.3e}", flush=True)


if __name__ == "__main__":
    main()
