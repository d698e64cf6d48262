"""Topology of the ADM UNet as a flat op list + the state_dict schema it implies.

The reference builds the network as nested nn.Modules (/root/reference/diffusion/backbones/adm.py:
318-490).  Here the same architecture is described as data: a list of `Res` / `Attn` op records
with their state_dict prefixes, channel counts and resolutions, derived from the constructor
arguments of `AdmUnet2d` (the keys of configs/*.json:backbone.args).  Both the parameter registry
of the torch shim (so reference checkpoints load with strict=True) and the HIP launch plan are
generated from this one description.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple


@dataclass
class Res:
    prefix: str          # state_dict prefix, e.g. "input_blocks.3.0"
    cin: int             # channels entering (after skip concat for output blocks)
    cout: int
    res_in: int          # spatial side of the input
    mode: str            # "same" | "down" | "up"
    skip_c: int = 0      # channels of the popped skip tensor concatenated behind h (adm.py:563)
    emb_off: int = 0     # column offset of this block's [scale|shift] in the fused emb projection

    @property
    def res_out(self):
        return {"same": self.res_in, "down": self.res_in // 2, "up": self.res_in * 2}[self.mode]

    @property
    def has_skip_conv(self):
        return self.cin != self.cout


@dataclass
class Attn:
    prefix: str
    c: int
    res: int
    heads: int


@dataclass
class Stage:
    """One entry of input_blocks / middle_block / output_blocks."""
    kind: str                      # "in" | "mid" | "out"
    ops: list = field(default_factory=list)
    conv_in: bool = False          # input_blocks.0 is the plain 3x3 stem conv (adm.py:368-370)


@dataclass
class UNetSpec:
    image_size: int
    in_channels: int
    out_channels: int
    model_channels: int
    emb_dim: int
    num_classes: Optional[int]
    has_null_class: bool
    num_groups: int
    stages: List[Stage]
    stem_out: int
    final_c: int
    emb_total: int                 # sum of 2*cout over all Res ops
    schema: List[Tuple[str, Tuple[int, ...], bool]]  # (name, shape, is_buffer)

    def res_ops(self):
        return [op for st in self.stages for op in st.ops if isinstance(op, Res)]


def build_spec(image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
               dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, num_classes=None, has_null_class=False,
               use_fp16=False, num_groups=32, num_heads=1, num_head_channels=-1, use_scale_shift_norm=True,
               resblock_updown=True) -> UNetSpec:
    if not (use_scale_shift_norm and resblock_updown):
        raise NotImplementedError("ivid_amd implements the FiLM + resblock-updown ADM variant used by every ivid config")
    if dropout:
        raise NotImplementedError("dropout is a training-only feature; inference configs use 0")
    mc = model_channels
    emb_dim = 4 * mc
    att = set(int(a) for a in attention_resolutions)

    def heads_for(c):
        if num_head_channels == -1:
            h = num_heads
        else:
            if c % num_head_channels:
                raise ValueError(f"q,k,v channels {c} is not divisible by num_head_channels {num_head_channels}")
            h = c // num_head_channels
        # csrc/attn.hip is written for head dim 64 (every ivid config: num_head_channels = 64); anything else would
        # stride the legacy [head][q|k|v][d] interleave wrongly, so refuse it here instead of returning garbage
        if not h or c % h or c // h != 64:
            raise NotImplementedError(f"attention head dim {c}/{h}: the HIP attention kernel supports head dim 64 only")
        return h

    stages: List[Stage] = []
    emb_off = 0

    def res(prefix, cin, cout, side, mode="same", skip_c=0):
        nonlocal emb_off
        op = Res(prefix, cin, cout, side, mode, skip_c, emb_off)
        emb_off += 2 * cout
        return op

    # ---- encoder: stem conv, then per level {num_res_blocks x (Res [+Attn])} + Res-down (adm.py:367-421)
    ch = int(channel_mult[0] * mc)
    stem_out = ch
    stages.append(Stage("in", [], conv_in=True))
    skip_chs = [ch]
    side = image_size
    idx = 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            cout = int(mult * mc)
            ops = [res(f"input_blocks.{idx}.0", ch, cout, side)]
            ch = cout
            if side in att:
                if (side * side) % 64:
                    raise NotImplementedError(f"attention at {side}x{side}: the HIP kernel needs T % 64 == 0")
                ops.append(Attn(f"input_blocks.{idx}.1", ch, side, heads_for(ch)))
            stages.append(Stage("in", ops))
            skip_chs.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            stages.append(Stage("in", [res(f"input_blocks.{idx}.0", ch, ch, side, "down")]))
            skip_chs.append(ch)
            idx += 1
            side //= 2

    # ---- bottleneck: Res, Attn, Res (adm.py:423-446)
    stages.append(Stage("mid", [
        res("middle_block.0", ch, ch, side),
        Attn("middle_block.1", ch, side, heads_for(ch)),
        res("middle_block.2", ch, ch, side),
    ]))

    # ---- decoder: per level (num_res_blocks+1) x (Res on [h|skip] [+Attn]) + Res-up (adm.py:448-490)
    idx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            sc = skip_chs.pop()
            cout = int(mc * mult)
            ops = [res(f"output_blocks.{idx}.0", ch + sc, cout, side, "same", skip_c=sc)]
            ch = cout
            if side in att:
                ops.append(Attn(f"output_blocks.{idx}.{len(ops)}", ch, side, heads_for(ch)))
            if level and i == num_res_blocks:
                ops.append(res(f"output_blocks.{idx}.{len(ops)}", ch, ch, side, "up"))
                side *= 2
            stages.append(Stage("out", ops))
            idx += 1

    # ---- state_dict schema (SURVEY.md §8a "State_dict schema"; nn.Module naming of the reference)
    schema: List[Tuple[str, Tuple[int, ...], bool]] = []

    def P(name, *shape):
        schema.append((name, tuple(shape), False))

    schema.append(("time_embed.0.freqs", (mc // 2,), True))  # persistent buffer (adm.py:28)
    P("time_embed.1.weight", emb_dim, mc); P("time_embed.1.bias", emb_dim)
    P("time_embed.3.weight", emb_dim, emb_dim); P("time_embed.3.bias", emb_dim)
    if num_classes is not None:
        P("label_emb.weight", num_classes, emb_dim)
    P("input_blocks.0.0.weight", stem_out, in_channels, 3, 3); P("input_blocks.0.0.bias", stem_out)
    for st in stages:
        for op in st.ops:
            p = op.prefix
            if isinstance(op, Res):
                P(f"{p}.in_layers.0.weight", op.cin); P(f"{p}.in_layers.0.bias", op.cin)
                P(f"{p}.in_layers.2.weight", op.cout, op.cin, 3, 3); P(f"{p}.in_layers.2.bias", op.cout)
                P(f"{p}.emb_layers.1.weight", 2 * op.cout, emb_dim); P(f"{p}.emb_layers.1.bias", 2 * op.cout)
                P(f"{p}.out_layers.0.weight", op.cout); P(f"{p}.out_layers.0.bias", op.cout)
                P(f"{p}.out_layers.3.weight", op.cout, op.cout, 3, 3); P(f"{p}.out_layers.3.bias", op.cout)
                if op.has_skip_conv:
                    P(f"{p}.skip_connection.weight", op.cout, op.cin, 1, 1); P(f"{p}.skip_connection.bias", op.cout)
            else:
                P(f"{p}.norm.weight", op.c); P(f"{p}.norm.bias", op.c)
                P(f"{p}.qkv.weight", 3 * op.c, op.c, 1); P(f"{p}.qkv.bias", 3 * op.c)
                P(f"{p}.proj_out.weight", op.c, op.c, 1); P(f"{p}.proj_out.bias", op.c)
    final_c = ch
    P("out.0.weight", final_c); P("out.0.bias", final_c)
    P("out.2.weight", out_channels, stem_out, 3, 3); P("out.2.bias", out_channels)

    return UNetSpec(image_size, in_channels, out_channels, mc, emb_dim, num_classes,
                    bool(has_null_class) if num_classes is not None else False, num_groups, stages, stem_out,
                    final_c, emb_off, schema)
