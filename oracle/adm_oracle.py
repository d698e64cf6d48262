"""Functional fp32 CPU restatement of the ADM UNet forward — TEST INFRASTRUCTURE (the oracle).

Restates /root/reference/diffusion/backbones/adm.py as plain functions over a bare state_dict,
walking the checkpoint's own key structure instead of a module tree.  Pinned against the live
reference by tests/golden/make_golden.py (run in the build container, where /root/reference is
importable) and by the committed fixtures in tests/golden/.
"""
import math

import torch
import torch.nn.functional as F


def pos_encoding(t, freqs):
    """adm.py:30-33 — [cos(t f), sin(t f)], cos first."""
    a = t.float()[:, None] * freqs[None, :]
    return torch.cat([a.cos(), a.sin()], dim=-1)


def embedding(sd, t, classes, has_null_class):
    """adm.py:545-555 — time MLP (+ label embedding, zeroed for the null class / classes None)."""
    e = pos_encoding(t, sd["time_embed.0.freqs"])
    e = F.linear(e, sd["time_embed.1.weight"], sd["time_embed.1.bias"])
    e = F.linear(F.silu(e), sd["time_embed.3.weight"], sd["time_embed.3.bias"])
    if "label_emb.weight" in sd and classes is not None:
        keep = classes >= 0
        ce = sd["label_emb.weight"][classes * keep.long()]
        if has_null_class:
            ce = ce * keep[:, None]
        e = e + ce
    return e


def gn32(x, w, b, groups):
    """adm.py:36-41 — GroupNorm in fp32, eps 1e-5."""
    return F.group_norm(x.float(), groups, w, b, 1e-5)


def resblock(sd, p, x, emb, mode, groups):
    """adm.py:192-222 with use_scale_shift_norm=True; mode in {same, up, down} (adm.py:203-208)."""
    h = F.silu(gn32(x, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], groups))
    if mode == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif mode == "down":
        h = F.avg_pool2d(h, 2)
        x = F.avg_pool2d(x, 2)
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    eo = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    scale, shift = eo[:, :, None, None].chunk(2, dim=1)
    h = gn32(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], groups) * (1 + scale) + shift
    h = F.conv2d(F.silu(h), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def qkv_attention(qkv, heads):
    """adm.py:243-253 — legacy layout: [B, heads*(q|k|v)*ch, T]; softmax in fp32."""
    b, width, t = qkv.shape
    ch = width // (3 * heads)
    q, k, v = qkv.reshape(b * heads, 3 * ch, t).split(ch, dim=1)
    s = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.float(), dim=-1)
    return torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, t)


def attnblock(sd, p, x, head_channels, groups):
    """adm.py:280-286."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(gn32(xf, sd[p + ".norm.weight"], sd[p + ".norm.bias"], groups), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    a = qkv_attention(qkv, c // head_channels)
    a = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + a).reshape(b, c, hh, ww)


def _stage(sd, prefix, h, emb, args, res_modes, trace=None):
    """Run the sub-layers prefix.0, prefix.1, ... of one ModSequential (adm.py:55-66)."""
    j = 0
    while True:
        p = f"{prefix}.{j}"
        if p + ".in_layers.0.weight" in sd:
            h = resblock(sd, p, h, emb, res_modes.get(p, "same"), args["num_groups"])
        elif p + ".qkv.weight" in sd:
            h = attnblock(sd, p, h, args["num_head_channels"], args["num_groups"])
        else:
            return h
        if trace is not None:
            trace[p] = h
        j += 1


def _updown_modes(sd, args):
    """Which ResBlocks resample (adm.py:401-420, 470-484): the single-Res stage that closes every
    encoder level but the last is 'down'; the last Res of the last stage of every decoder level
    but the final one is 'up'."""
    modes = {}
    nrb, nlev = args["num_res_blocks"], len(args["channel_mult"])
    idx = 1
    for level in range(nlev):
        idx += nrb
        if level != nlev - 1:
            modes[f"input_blocks.{idx}.0"] = "down"
            idx += 1
    idx = 0
    for level in reversed(range(nlev)):
        for i in range(nrb + 1):
            if level and i == nrb:
                j = 0
                while f"output_blocks.{idx}.{j + 1}.in_layers.0.weight" in sd or f"output_blocks.{idx}.{j + 1}.qkv.weight" in sd:
                    j += 1
                modes[f"output_blocks.{idx}.{j}"] = "up"
            idx += 1
    return modes


@torch.no_grad()
def unet_forward(sd, args, x, t, classes=None, trace=None):
    """adm.py:526-566 in fp32.  sd: bare state_dict (fp32 CPU); args: configs/*.json:backbone.args."""
    sd = {k: v.float() for k, v in sd.items()}
    has_null = bool(args.get("has_null_class", False)) and args.get("num_classes") is not None
    emb = embedding(sd, t, classes, has_null)
    modes = _updown_modes(sd, args)
    hs = []
    h = F.conv2d(x.float(), sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
    hs.append(h)
    if trace is not None:
        trace["stem"] = h
    i = 1
    while f"input_blocks.{i}.0.in_layers.0.weight" in sd:
        h = _stage(sd, f"input_blocks.{i}", h, emb, args, modes, trace)
        hs.append(h)
        i += 1
    h = _stage(sd, "middle_block", h, emb, args, modes, trace)
    i = 0
    while f"output_blocks.{i}.0.in_layers.0.weight" in sd:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _stage(sd, f"output_blocks.{i}", h, emb, args, modes, trace)
        i += 1
    h = F.silu(gn32(h, sd["out.0.weight"], sd["out.0.bias"], args["num_groups"]))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
