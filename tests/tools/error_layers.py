"""Per-site error budget of the 16-bit modes on the representative forward set (dev tool, TEST INFRASTRUCTURE).

Emulates the product's fp16 roundings inside the fp32 oracle forward ONE SITE AT A TIME (torch, on the GPU when there is one:
the reference of every comparison is the same torch forward without roundings, so the backend's own fp32 noise cancels):
    W:<conv>   the weights of one convolution rounded to fp16            A:<conv>   its input activations rounded
    H:<block>  the tensor between a ResBlock's two convolutions rounded    F:<block>  the branch reads the trunk's hi plane alone
    Q:<attn>   attention internals rounded
on all inputs of tests/common.fwd_set_inputs (both CFG branches = 24 rows of one batch).  Roundings are independent, so the
squared deviations add: the table says which sites a precision mode has to upgrade, and what each costs in FLOPs.

    python tests/tools/error_layers.py --out gpurun_out/error_layers.json
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import common as C  # noqa: E402
from oracle import adm_oracle as O  # noqa: E402


class Sites:
    """on(kind, site) -> bool decides whether a rounding is applied; records every site it is asked about + conv MACs."""

    def __init__(self, pred=None, dt=torch.float16):
        self.pred, self.dt, self.seen, self.macs = pred, dt, [], {}

    def __call__(self, kind, site, x):
        key = f"{kind}:{site}"
        if key not in self.seen:
            self.seen.append(key)
        if self.pred is not None and self.pred(kind, site):
            return x.to(self.dt).float()
        return x

    def conv(self, site, a, w, b, padding):
        out = F.conv2d(self("A", site, a), self("W", site, w), b, padding=padding)
        self.macs[site] = out.shape[2] * out.shape[3] * w.numel()
        return out


def resblock(sd, p, x, emb, mode, groups, q):
    side = x.shape[-1] * (2 if mode == "up" else 1) // (2 if mode == "down" else 1)
    xb = q("F", p, x) if (mode == "same" and side >= 32) else x
    h = F.silu(O.gn32(xb, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], groups))
    if mode == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif mode == "down":
        h = F.avg_pool2d(h, 2)
        x = F.avg_pool2d(x, 2)
    h = q.conv(p + ".in_layers.2", h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], 1)
    h = q("H", p, h)
    eo = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    scale, shift = eo[:, :, None, None].chunk(2, dim=1)
    h = O.gn32(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], groups) * (1 + scale) + shift
    h = q.conv(p + ".out_layers.3", F.silu(h), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], 1)
    if p + ".skip_connection.weight" in sd:
        x = q.conv(p + ".skip_connection", xb if mode == "same" else x, sd[p + ".skip_connection.weight"],
                   sd[p + ".skip_connection.bias"], 0)
    return x + h


def attnblock(sd, p, x, head_channels, groups, q):
    b, c, hh, ww = x.shape
    xn = q("Q", p, O.gn32(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], groups))
    qkv = q("Q", p, q.conv(p + ".qkv", xn, sd[p + ".qkv.weight"][..., None], sd[p + ".qkv.bias"], 0)).reshape(b, 3 * c, -1)
    heads = c // head_channels
    ch = c // heads
    qq, k, v = qkv.reshape(b * heads, 3 * ch, -1).split(ch, dim=1)
    w = torch.softmax(torch.einsum("bct,bcs->bts", qq, k) / ch ** 0.5, dim=-1)
    a = q("Q", p, torch.einsum("bts,bcs->bct", q("Q", p, w), v).reshape(b, c, hh, ww))
    a = q.conv(p + ".proj_out", a, sd[p + ".proj_out.weight"][..., None], sd[p + ".proj_out.bias"], 0)
    return x + a


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
    """stem and head stay exact (fp16c runs them in split form)."""
    has_null = bool(args.get("has_null_class", False)) and args.get("num_classes") is not None
    emb = O.embedding(sd, t, classes, has_null)
    modes = O._updown_modes(sd, args)
    hs = []
    h = F.conv2d(x, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
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
    h = F.silu(O.gn32(h, sd["out.0.weight"], sd["out.0.bias"], args["num_groups"]))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def rows_rel(a, ref):
    d = (a.double() - ref.double()).flatten(1).norm(dim=1) / ref.double().flatten(1).norm(dim=1)
    return [float(v) for v in d.cpu()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large")
    ap.add_argument("--out", default="gpurun_out/error_layers.json")
    ap.add_argument("--rows", default="", help="comma list of row indices to keep (default all)")
    ap.add_argument("--combos-only", action="store_true", help="skip the per-site sweep")
    ap.add_argument("--extra", default="", help="extra combos on top of cx+skip: 'name=K@prefix,K@prefix;name=...' -- kind K (W / A) is kept "
                    "EXACT at every site whose name starts with prefix (e.g. W@input_blocks.2.0.,A@input_blocks.1.0.in_layers)")
    ap.add_argument("--only-extra", action="store_true", help="run the --extra combos only")
    ap.add_argument("--extra-base", default="WAQ", help="rounding classes of the --extra combos' base mode: WAQ = fp16cx, WAHFQ = fp16c")
    ap.add_argument("--mid", action="store_true", help="the rows of the mid-t sets (tests/common.FWD_SET_T_MID) instead of the main set")
    ap.add_argument("--combos", default="", help="semicolon list: run these named combos only (fp16c, fp16cx, cx+skip = fp16cs, "
                    "cx+skip+X(ib1,2) = fp16s, ...)")
    ap.add_argument("--tag", default="", help="a tests/common.FWD_SETS entry of a 4-channel model (e.g. large128_s11, small128_tr24): its "
                    "architecture, synthetic checkpoint and rows instead of --model / --mid")
    a = ap.parse_args()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    args, seed = (C.LARGE128, 4) if a.model == "large" else (C.SMALL128, 3)
    ins = C.fwd_set_inputs(args["in_channels"], args["image_size"], *((C.FWD_SET_T_MID, 7050) if a.mid else ()))
    if a.tag:
        args, seed, _g, make, _crop = C.FWD_SETS[a.tag]
        assert args["in_channels"] == 4
        ins = make()
    sd = {k: v.float().to(dev) for k, v in C.synth_weights(args, seed).items()}
    has_cls = args.get("num_classes") is not None
    xs, ts, cs, names = [], [], [], []
    for key, x, t, cls in ins:
        for b in (("c", "u") if has_cls else ("u",)):
            xs.append(x); ts.append(t); cs.append(cls if b == "c" else -1); names.append(f"{key}_{b}")
    if a.rows:
        keep = [int(v) for v in a.rows.split(",")]
        xs, ts, cs, names = [[v[i] for i in keep] for v in (xs, ts, cs, names)]
    x = torch.cat(xs).to(dev); t = torch.tensor(ts, device=dev); cl = torch.tensor(cs, device=dev) if has_cls else None
    q0 = Sites()
    ref = forward(sd, args, x, t, cl, q0)
    sites = list(q0.seen)
    res = dict(rows=names, macs=q0.macs, sites={}, combos={})
    combos = {"fp16c": lambda k, s: k in "WAHQF", "fp16cx": lambda k, s: k in "WAQ", "cx+H": lambda k, s: k in "WAQH",
              "cx+F": lambda k, s: k in "WAQF", "W": lambda k, s: k == "W",
              "A": lambda k, s: k == "A", "H": lambda k, s: k == "H", "F": lambda k, s: k == "F", "Q": lambda k, s: k == "Q"}
    # candidate selective modes: fp16cx + every 1x1 skip_connection in split precision (+ a second MFMA pass with the weights'
    # lo part on the 3x3 convolutions of the first ResBlocks)
    skip = lambda s: "skip_connection" in s
    def sel(wfix, base="WAQ"):
        return lambda k, s: k in base and not skip(s) and not (k == "W" and any(s.startswith(p) for p in wfix))
    ib = lambda *i: tuple(f"input_blocks.{j}.0." for j in i)
    combos.update({"cx+skip": sel(()), "cx+skip+W(ib1)": sel(ib(1)), "cx+skip+W(ib1,2)": sel(ib(1, 2)),
                   "cx+skip+W(ib1,3)": sel(ib(1, 3)), "cx+skip+W(ib1,2,3)": sel(ib(1, 2, 3)),
                   "c+skip+W(ib1)": sel(ib(1), "WAHQF"),
                   "cx+skip+W(ib1,ob14)": sel(ib(1) + ("output_blocks.14.0.",))})
    def selx(xfix, base="WAQ"):   # both operands exact (three MFMA passes) in the blocks of xfix
        return lambda k, s: k in base and not skip(s) and not any(s.startswith(p) for p in xfix)
    ob = lambda *i: tuple(f"output_blocks.{j}.0." for j in i)
    combos.update({"cx+skip+X(ib1)": selx(ib(1)), "cx+skip+X(ib1,2)": selx(ib(1, 2)), "cx+skip+X(ib1,2,3)": selx(ib(1, 2, 3)),
              "cx+skip+X(ib1,2,ob12,13,14)": selx(ib(1, 2) + ob(12, 13, 14)),
              "cx+skip+X(ib1,2,3,ob11,12,13,14)": selx(ib(1, 2, 3) + ob(11, 12, 13, 14)),
              "c+skip+X(ib1,2,3)": selx(ib(1, 2, 3), "WAHQF"),
              "cx+X(ib1,2,3)": lambda k, s: k in "WAQ" and not any(s.startswith(p) for p in ib(1, 2, 3)),
              "cx+skip+X(ib1,2,3)+W(ob12,13,14)": (lambda k, s: selx(ib(1, 2, 3))(k, s) and not (k == "W" and any(s.startswith(p) for p in ob(12, 13, 14))))})
    if a.combos:
        combos = {k: combos[k] for k in a.combos.split(";")}
    if a.only_extra:
        combos = {}
    for spec in filter(None, a.extra.split(";")):
        name, items = spec.split("=", 1)
        fixes = [tuple(it.split("@", 1)) for it in items.split(",")]
        combos[name] = (lambda fx: lambda k, s: k in a.extra_base and not skip(s) and not any(k == kk and s.startswith(pp) for kk, pp in fx))(fixes)
    for name, pred in combos.items():
        res["combos"][name] = rows_rel(forward(sd, args, x, t, cl, Sites(pred)), ref)
        print(name, f"max {max(res['combos'][name]):.3e}", flush=True)
    for key in ([] if a.combos_only else sites):
        k0, s0 = key.split(":", 1)
        res["sites"][key] = rows_rel(forward(sd, args, x, t, cl, Sites(lambda k, s: k == k0 and s == s0)), ref)
        print(key, f"max {max(res['sites'][key]):.3e}", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=0)


if __name__ == "__main__":
    main()
