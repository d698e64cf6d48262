"""Launch plan of one ADM UNet forward on MI355X.

A plan is built once per (batch, stacked-CFG) shape: it owns the activation arena (NHWC tensors of
the compute dtype), the repacked weights are shared, and the forward is a precomputed list of
C-ABI launches (include/ivid_hip.h) that is replayed as ONE hipGraph after the first eager run —
the reference issues ~650-700 separate torch kernels per forward.

Data flow of a ResBlock (adm.py:192-222), all tensors NHWC:
    stats(x|skip) -> (a,b) -> act1 = silu(x*a+b) [+up/down resample, concat materialised]
    h1 = conv3x3(act1)                     (bias fused)
    stats(h1) + FiLM(scale,shift) -> (a,b) -> act2 = silu(h1*a+b)
    out = conv3x3(act2) + bias + skip      (skip = x, resampled x, or conv1x1(x|skip))
"""
import ctypes as C
import os

import torch

from ... import _lib
from .spec import Attn, Res, UNetSpec

_TORCH_DT = {_lib.F32: torch.float32, _lib.BF16: torch.bfloat16, _lib.F16: torch.float16, _lib.BF16X3: torch.float32}


def split_pack(w2d):
    """IVID_BF16X3 weight layout: fp32 [Cout, K] -> bf16 [Cout, 2K]; per 8 input channels 8 x hi then 8 x lo
    (hi = bf16(w) round-to-nearest-even, lo = bf16(w - hi)), so a row keeps the byte size of an fp32 row."""
    cout, k = w2d.shape
    assert k % 8 == 0
    hi = w2d.to(torch.bfloat16)
    lo = (w2d - hi.float()).to(torch.bfloat16)
    return torch.stack([hi.view(cout, k // 8, 8), lo.view(cout, k // 8, 8)], dim=2).reshape(cout, 2 * k).contiguous()


def hi_lo(w, tdt):
    """fp32 -> (hi, lo) in the 16-bit type tdt: hi = tdt(w), lo = tdt(w - hi) (the difference is exact in fp32)."""
    hi = w.to(tdt)
    return hi, (w - hi.float()).to(tdt)


def up4_weights(w):
    """Conv2d 3x3 weights [Cout,Cin,3,3] -> the phase-summed form [4*Cout, 4*Cin] ivid_conv3x3_up takes
    (rows [phase = py*2+px][cout], columns [tap = a*2+b][cin]): behind a nearest x2 upsample (adm.py:70-83, 203-206)
    the 3x3 taps of output pixel (2y+py, 2x+px) that land on the same source pixel are added up, which leaves a 2x2
    convolution of the source per phase -- rows {ky=0 | ky=1,2} for py = 0 and {ky=0,1 | ky=2} for py = 1, columns alike."""
    cout, cin = w.shape[:2]
    sets = {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}    # (phase bit, tap bit) -> kernel rows / columns
    out = w.new_empty(2, 2, cout, 2, 2, cin)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    out[py, px, :, a, b, :] = w[:, :, sets[(py, a)]][:, :, :, sets[(px, b)]].sum(dim=(2, 3))
    return out.reshape(4 * cout, 4 * cin)


def _pad_to(c, m):
    return (c + m - 1) // m * m


def island_stages(spec: UNetSpec):
    """Indices (into spec.stages) of the encoder stages in front of the first down-sampling ResBlock: with the stem they form
    the split-precision island of the fp16s mode.  Their outputs are the youngest states of the residual trunk (the stem
    output plus one or two branches) and they reach the decoder's last blocks directly through the skip stash
    (adm.py:557-563), so the operand roundings of their convolutions are the ones the output sees least damped."""
    out = []
    for i, st in enumerate(spec.stages):
        if st.conv_in:
            continue
        if st.kind != "in" or any(isinstance(o, Res) and o.mode == "down" for o in st.ops):
            break
        out.append(i)
    return out


class PackedWeights:
    """Weights repacked once into the kernels' layouts: conv [Cout][tap][Cin] in the compute dtype,
    Linear / GroupNorm / bias / embedding tables in fp32, all emb_layers fused into one matrix."""

    def __init__(self, spec: UNetSpec, sd, device, dtype, comp=False, island=None):
        self.dtype = dtype
        # 1: precision mode fp16c (compensated trunk storage, split stem and head); 2: fp16cx (+ input lo planes);
        # 3: fp16s (+ split-precision 1x1 skip convolutions; stem + first encoder level as a bf16x3 island)
        self.comp = int(comp)
        assert not comp or _lib.esz(dtype) == 2
        tdt = _TORCH_DT[dtype]
        self.kstep = 128 // _lib.esz(dtype)
        g = lambda k: sd[k].detach().to(device=device, dtype=torch.float32)
        t = {}
        # matrix operand of the MFMA kernels: compute dtype, or the hi/lo split form of the bf16x3 mode
        mat_split = lambda w: split_pack(w.contiguous())
        mat = mat_split if dtype == _lib.BF16X3 else (lambda w: w.to(tdt).contiguous())
        # island = None: what the level implies (3 -> yes).  False with level 3 = precision mode fp16cs: compensated storage, lo-plane
        # inputs and split-precision skip convolutions WITHOUT the bf16x3 island (inside the tolerance once the input carries
        # diffusion noise, t >= 250: the high-t half of the adaptive mode fp16sa, adm.py)
        self.island_on = (self.comp >= 3) if island is None else bool(island)
        assert not self.island_on or self.comp >= 3
        self.island = set(island_stages(spec)) if self.island_on else set()
        isl = {o.prefix for i in self.island for o in spec.stages[i].ops}    # ops whose weights take the bf16x3 layout

        def conv3(name, pad_cin=None):
            w = g(name + ".weight").permute(0, 2, 3, 1)  # [Cout,3,3,Cin]
            if pad_cin is not None and pad_cin != w.shape[-1]:
                w = torch.nn.functional.pad(w, (0, pad_cin - w.shape[-1]))
            m = mat_split if name.rsplit(".", 2)[0] in isl else mat
            t[name + ".weight"] = m(w.reshape(w.shape[0], -1))
            t[name + ".bias"] = g(name + ".bias").contiguous()

        def conv1(name):
            w = g(name + ".weight")
            w2 = w.reshape(w.shape[0], -1)
            if name.rsplit(".", 1)[0] in isl:
                t[name + ".weight"] = mat_split(w2)
            else:
                t[name + ".weight"] = mat(w2)
                if self.comp >= 3 and name.endswith(".skip_connection"):   # lo part for the split-precision skip phase
                    t[name + ".weight_lo"] = hi_lo(w2, tdt)[1].contiguous()
            t[name + ".bias"] = g(name + ".bias").contiguous()

        def vec(name):
            t[name + ".weight"] = g(name + ".weight").contiguous()
            t[name + ".bias"] = g(name + ".bias").contiguous()

        # stem: the 3x3 patch of the few input channels is ONE K row (k = tap*Cin + c, ivid_stem_im2col), padded to a K-step
        ws = g("input_blocks.0.0.weight").permute(0, 2, 3, 1).reshape(spec.stem_out, -1)   # [Cout, 9*Cin]
        if self.island_on:   # the stem opens the bf16x3 island: plain im2col row, weights in the hi/lo split layout (K-step 32)
            self.stem_k = _pad_to(9 * spec.in_channels, 32)
            t["input_blocks.0.0.weight"] = mat_split(torch.nn.functional.pad(ws, (0, self.stem_k - ws.shape[1])))
        elif comp:   # split stem (ivid_stem_im2col_split): K row [x_hi | x_lo | x_hi] against [w_hi | w_hi | w_lo]
            self.stem_k = _pad_to(27 * spec.in_channels, self.kstep)
            whi, wlo = hi_lo(ws, tdt)
            w3 = torch.cat([whi, whi, wlo], dim=1)
            t["input_blocks.0.0.weight"] = torch.nn.functional.pad(w3, (0, self.stem_k - w3.shape[1])).contiguous()
        else:
            self.stem_k = _pad_to(9 * spec.in_channels, self.kstep)
            t["input_blocks.0.0.weight"] = mat(torch.nn.functional.pad(ws, (0, self.stem_k - ws.shape[1])))
        t["input_blocks.0.0.bias"] = g("input_blocks.0.0.bias").contiguous()
        emb_w, emb_b = [], []
        for op in [o for st in spec.stages for o in st.ops]:
            p = op.prefix
            if isinstance(op, Res):
                vec(p + ".in_layers.0"); conv3(p + ".in_layers.2")
                if op.mode == "up":   # phase-summed weights of upsample + conv (ivid_conv3x3_up), summed in fp32
                    t[p + ".in_layers.2.weight_up4"] = mat(up4_weights(g(p + ".in_layers.2.weight")).contiguous())
                vec(p + ".out_layers.0"); conv3(p + ".out_layers.3")
                if op.has_skip_conv:
                    conv1(p + ".skip_connection")
                emb_w.append(g(p + ".emb_layers.1.weight")); emb_b.append(g(p + ".emb_layers.1.bias"))
            else:
                vec(p + ".norm"); conv1(p + ".qkv"); conv1(p + ".proj_out")
        vec("out.0"); conv3("out.2")
        if comp:                   # split output head (ivid_conv3x3_gn_out_c): hi and lo parts of the weights
            wo = g("out.2.weight").permute(0, 2, 3, 1).reshape(spec.out_channels, -1)
            t["out.2.weight"], t["out.2.weight_lo"] = [v.contiguous() for v in hi_lo(wo, tdt)]
        if dtype == _lib.BF16X3:   # the output head runs its fp32 kernel in this mode (0.05 % of the FLOPs): plain fp32 weights
            t["out.2.weight"] = g("out.2.weight").permute(0, 2, 3, 1).reshape(spec.out_channels, -1).contiguous()
        # embedding path stays fp32 in both modes (the reference never casts nn.Linear, backbones/utils.py:6-13)
        t["emb_all.weight"] = torch.cat(emb_w, 0).contiguous()
        t["emb_all.bias"] = torch.cat(emb_b, 0).contiguous()
        for n in ("time_embed.1", "time_embed.3"):
            vec(n)
        t["freqs"] = g("time_embed.0.freqs").contiguous()
        if spec.num_classes is not None:
            t["label_emb"] = g("label_emb.weight").contiguous()
        self.t = t

    def __getitem__(self, k):
        return self.t[k]

    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in self.t.values())


class _Arena:
    """Stream-ordered buffer pool: a buffer may be handed out again as soon as its last consumer
    launch has been *enqueued* (single stream, in-order)."""

    def __init__(self, device):
        self.device = device
        self.free = []
        self.all = []

    def get(self, nbytes):
        nbytes = max(256, (nbytes + 255) // 256 * 256)
        best = None
        for i, b in enumerate(self.free):
            if b.numel() >= nbytes and (best is None or b.numel() < self.free[best].numel()):
                best = i
        if best is not None and self.free[best].numel() <= 2 * nbytes:
            return self.free.pop(best)
        b = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.all.append(b)
        return b

    def put(self, b):
        self.free.append(b)

    def total_bytes(self):
        return sum(b.numel() for b in self.all)


class _Act:
    """An NHWC activation living in an arena buffer (+ optionally the GroupNorm partial statistics its producing
    convolution wrote: fp32 [n*side*side/32][c][2]).  `lo`: byte offset of the tensor's lo plane inside the same buffer
    (compensated 16-bit storage, include/ivid_hip.h ivid_conv2d_c) or None."""
    __slots__ = ("buf", "n", "side", "c", "stats", "stats_blk", "lo", "twin")

    def __init__(self, buf, n, side, c, stats=None, lo=None):
        self.buf, self.n, self.side, self.c, self.stats, self.lo = buf, n, side, c, stats, lo
        self.stats_blk = 32   # pixels per statistics block, set by the producing launch
        self.twin = None      # fp32 island tensor: its fp16 hi + lo form, written by the producer itself (ivid_conv3x3_gn_o16)

    @property
    def ptr(self):
        return self.buf.data_ptr()

    @property
    def lo_ptr(self):
        return self.buf.data_ptr() + self.lo if self.lo is not None else None


class UNetPlan:
    def __init__(self, spec: UNetSpec, weights: PackedWeights, device, bsrc, stacked, tile_cfg=0, debug=False):
        self.spec, self.w, self.device = spec, weights, device
        self.debug = debug      # eager-only: snapshot every op's output (NCHW fp32) into self.taps
        self.fuse_stats = os.environ.get("IVID_NO_FUSED_STATS", "0") != "1"   # GN partials from conv epilogues
        self.fuse_conv = os.environ.get("IVID_NO_FUSED_CONV", "0") != "1"     # GN-apply+SiLU inside the 3x3 conv (W >= 32)
        self.fuse_skip = os.environ.get("IVID_NO_FUSED_SKIP", "0") != "1"     # 1x1 skip_connection inside that kernel too
        self.fuse_narrow = os.environ.get("IVID_NO_FUSED128", "0") != "1"     # Cout <= 128 through the 16x32x128 variant
        # in_layers conv of an `up` ResBlock as four 2x2 phase convolutions of the low-resolution tensor (4/9 of the
        # multiplications, csrc/conv_igemm.hip ivid_conv3x3_up) for sources up to this side; 0 disables it
        self.up4_max_side = int(os.environ.get("IVID_UP4_MAX_SIDE", "1024"))
        # stacked CFG forward: what both halves of the batch have in common (everything in front of the first FiLM) is
        # computed once and duplicated
        self.share_cfg = os.environ.get("IVID_NO_CFG_SHARE", "0") != "1"
        self.island_o16 = os.environ.get("IVID_NO_ISLAND_O16", "0") != "1"   # island blocks write their fp16 twin themselves
        self._island_last = None
        self.pool_res = os.environ.get("IVID_NO_POOL_RES", "0") != "1"       # `down` blocks: pooled residual from the gn_apply pass
        self._first_res_done = False
        self._sum_bias = {}
        self._tile_1x1 = int(os.environ.get("IVID_TILE_1X1", "0"))
        self._tile_up4 = int(os.environ.get("IVID_TILE_UP4", "0"))   # tuning hook: tile of the phase-form up-convolutions
        self.taps = {}
        self.dtype = weights.dtype
        self.comp = weights.comp
        self.comp_in = weights.comp >= 2     # fp16cx: lo planes feed the fused halo transform; h1 is compensated too
        self.split_skip = weights.comp >= 3  # fp16s: 1x1 skip convolutions in split precision (x_hi.w_hi + x_lo.w_hi + x_hi.w_lo)
        self._main_mode = (self.dtype, self.comp, self.comp_in, self.split_skip)
        self.esz = _lib.esz(self.dtype)
        self.bsrc = bsrc
        self.n = 2 * bsrc if stacked else bsrc
        self.null_from = bsrc if stacked else self.n
        self.tile_cfg = tile_cfg
        self.lib = _lib.load()
        self.arena = _Arena(device)
        self.launches = []      # (cfunc, args)
        self.graph = None
        self.warm = False
        S = spec.image_size
        f32 = dict(dtype=torch.float32, device=device)
        self.x_in = torch.zeros(bsrc, spec.in_channels, S, S, **f32)
        self.t_in = torch.zeros(bsrc, dtype=torch.int64, device=device)
        self.c_in = torch.full((bsrc,), -1, dtype=torch.int64, device=device)
        self.out = torch.zeros(self.n, spec.out_channels, S, S, **f32)
        self._keep = []         # small fp32 tensors referenced by raw pointer
        # hipGraph capture is illegal on the legacy default stream, which is torch's current stream
        # unless the caller changed it: the plan runs on its own stream, fenced against the caller's.
        self.stream = torch.cuda.Stream(device=device)
        self._build()
        self.program = None
        if not debug and os.environ.get("IVID_PY_LAUNCH", "0") != "1":
            self._make_program()

    def _make_program(self):
        """Hand the planned launch list to the C-side program (csrc/program.hip): from here on a forward is ONE C call
        (ivid_unet_forward: input copies + hipGraph replay + nothing else), no Python per launch."""
        h = C.c_void_p()
        _lib.call("ivid_program_create", C.byref(h))
        for fn, name, args in self.launches:
            arr = _lib.pack_args(name, args)
            _lib.call("ivid_program_add", h, _lib.OP_CODES[name], C.cast(arr, C.c_void_p), len(args))
        has_cls = self.spec.num_classes is not None
        _lib.call("ivid_unet_bind", h, self.x_in.data_ptr(), self.x_in.numel() * 4, self.t_in.data_ptr(),
                  self.c_in.data_ptr() if has_cls else None, self.bsrc, self.out.data_ptr(), self.out.numel() * 4)
        self.program = h

    # ---- fp16s: the split-precision island (stem + first encoder level run as the bf16x3 mode would: fp32 storage) ----
    def _set_island(self, on):
        if on:
            self.dtype, self.comp, self.comp_in, self.split_skip = _lib.BF16X3, 0, False, False
        else:
            self.dtype, self.comp, self.comp_in, self.split_skip = self._main_mode
        self.esz = _lib.esz(self.dtype)

    def _to16(self, a):
        """fp32 island tensor -> the compensated 16-bit storage form (hi + lo planes) of the main mode; the GroupNorm partials
        its producer wrote move over unchanged (they describe the same values up to 2^-22)."""
        assert self.dtype != _lib.BF16X3
        y = a.twin if a.twin is not None else self._new(a.n, a.side, a.c, trunk=True)
        y.stats, y.stats_blk, a.stats = a.stats, a.stats_blk, None
        if a.twin is None:
            self._rec("ivid_f32_to_hilo", self.dtype, a.ptr, y.ptr, y.lo_ptr, a.n * a.side * a.side * a.c)
        self._free(a)
        return y

    def _new16(self, n, side, c):
        """An activation in the MAIN mode's compensated 16-bit storage, whatever the current (island) mode is."""
        nb = (n * side * side * c * 2 + 255) // 256 * 256
        return _Act(self.arena.get(2 * nb), n, side, c, None, lo=nb)

    # ---- launch recording ----
    def _rec(self, name, *args):
        self.launches.append((getattr(self.lib, name), name, args))

    def _tap(self, name, act):
        if not self.debug:
            return
        tdt = _TORCH_DT[self.dtype]
        nbytes = act.n * act.side * act.side * act.c * self.esz

        def snap(buf=act.buf, shape=(act.n, act.side, act.side, act.c), lo=act.lo):
            v = buf[:nbytes].view(tdt).view(shape).permute(0, 3, 1, 2).float().clone()
            if lo is not None:
                v += buf[lo:lo + nbytes].view(tdt).view(shape).permute(0, 3, 1, 2).float()
            self.taps[name] = v
        self.launches.append((None, name, snap))

    def _f32(self, *shape):
        t = torch.empty(*shape, dtype=torch.float32, device=self.device)
        self._keep.append(t)
        return t

    def _new(self, n, side, c, stats=False, trunk=False):
        """stats=True: the tensor will be GroupNorm'ed later -> its producer also emits the partial statistics.
        trunk=True: a tensor of the residual stream (adm.py:222,286) -- carries a lo plane in the compensated mode."""
        st = self.arena.get(n * side * side // 32 * c * 2 * 4) if (stats and self.fuse_stats) else None
        nb = n * side * side * c * self.esz
        if trunk and self.comp:
            nb = (nb + 255) // 256 * 256
            return _Act(self.arena.get(2 * nb), n, side, c, st, lo=nb)
        return _Act(self.arena.get(nb), n, side, c, st)

    def _free(self, act):
        self.arena.put(act.buf)
        if act.stats is not None:
            self.arena.put(act.stats)

    def _conv(self, dtype, src0, c0, src1, c1, wname, out_ptr, res_ptr, res_mode, out_mode, n, h, w, cout, taps,
              stats=None, out_act=None, out_lo=None, res_lo=None, wkey=".weight", no_bias=False):
        if out_act is not None and out_act.stats is not None:
            stats = out_act.stats
            out_act.stats_blk = self.lib.ivid_conv2d_stats_block(n, h, w, cout, self.tile_cfg)
        tile = self.tile_cfg
        if tile == 0 and taps == 1 and self._tile_1x1:
            tile = self._tile_1x1          # tuning hook (IVID_TILE_1X1): tile of the pointwise convolutions
        bias = None if no_bias else self.w[wname + ".bias"].data_ptr()
        if out_lo is not None or res_lo is not None:
            self._rec("ivid_conv2d_c", dtype, src0, c0, src1, c1, self.w[wname + wkey].data_ptr(),
                      bias, out_ptr, out_lo, res_ptr, res_lo, res_mode, out_mode, n, h, w, cout,
                      taps, tile, stats.data_ptr() if stats is not None else None)
            return
        self._rec("ivid_conv2d", dtype, src0, c0, src1, c1, self.w[wname + wkey].data_ptr(),
                  bias, out_ptr, res_ptr, res_mode, out_mode, n, h, w, cout, taps,
                  tile, stats.data_ptr() if stats is not None else None)

    def _linear(self, x, k, wname, out, cout, res=None):
        self._conv(_lib.F32, x.data_ptr(), k, None, 0, wname, out.data_ptr(), res.data_ptr() if res is not None else None,
                   1 if res is not None else 0, 0, self.n, 1, 1, cout, 1)

    def _gn_coeffs(self, x0: _Act, x1, gname, film_off, film_row0=0):
        """GroupNorm statistics of cat(x0,x1) folded with gamma/beta (+FiLM) -> per-(image, channel) (a, b) buffer.
        film_row0: image i of x0 takes the FiLM row film_row0 + i (a tensor shared by both halves of a stacked CFG batch is
        normalised once per half, each with its own class embedding)."""
        n, side = x0.n, x0.side
        c0, c1 = x0.c, (x1.c if x1 is not None else 0)
        c = c0 + c1
        hw = side * side
        ab = self.arena.get(n * c * 2 * 4)
        film = self.embproj.data_ptr() + film_row0 * self.spec.emb_total * 4 if film_off is not None else None
        fo = film_off if film_off is not None else 0
        gw, gb = self.w[gname + ".weight"].data_ptr(), self.w[gname + ".bias"].data_ptr()
        if x0.stats is not None and (x1 is None or x1.stats is not None):
            # statistics came for free from the producers' epilogues (one buffer per concat source)
            self._rec("ivid_gn_finalize2", x0.stats.data_ptr(), c0, hw // x0.stats_blk,
                      x1.stats.data_ptr() if x1 is not None else None, c1, hw // x1.stats_blk if x1 is not None else 0,
                      n, hw, self.spec.num_groups, 1e-5, gw, gb, film, self.spec.emb_total, fo, ab.data_ptr())
        else:
            nch = self.lib.ivid_gn_num_chunks(hw)
            partial = self.arena.get(n * nch * c * 2 * 4)
            lo0, lo1 = x0.lo_ptr, (x1.lo_ptr if x1 is not None else None)
            if lo0 is not None or lo1 is not None:   # the consumers of a tensor with a lo plane normalise hi + lo
                self._rec("ivid_gn_partial_c", self.dtype, x0.ptr, lo0, c0, x1.ptr if x1 is not None else None, lo1, c1, n, hw,
                          partial.data_ptr())
            else:
                self._rec("ivid_gn_partial", self.dtype, x0.ptr, c0, x1.ptr if x1 is not None else None, c1, n, hw,
                          partial.data_ptr())
            self._rec("ivid_gn_finalize", partial.data_ptr(), nch, n, c, hw, self.spec.num_groups, 1e-5, gw, gb, film,
                      self.spec.emb_total, fo, ab.data_ptr())
            self.arena.put(partial)
        return ab

    def _gn(self, x0: _Act, x1, gname, film_off, resample, act, use_lo=True, pool=None):
        """GroupNorm(+FiLM)(+SiLU)(+resample) of cat(x0,x1) materialised as a new activation (unfused path).  use_lo=False: read
        the hi planes alone even where lo planes exist.  pool: an activation that receives the 2x2 average of the RAW input
        (resample 2: the residual of a `down` ResBlock, adm.py:205-208)."""
        n, side = x0.n, x0.side
        c0, c1 = x0.c, (x1.c if x1 is not None else 0)
        ab = self._gn_coeffs(x0, x1, gname, film_off)
        so = {0: side, 1: side * 2, 2: side // 2}[resample]
        y = self._new(n, so, c0 + c1)
        lo0, lo1 = (x0.lo_ptr, (x1.lo_ptr if x1 is not None else None)) if use_lo else (None, None)
        if pool is not None:
            assert resample == 2 and self.esz == 2
            self._rec("ivid_gn_apply_p", self.dtype, x0.ptr, lo0, c0, x1.ptr if x1 is not None else None, lo1, c1,
                      ab.data_ptr(), y.ptr, pool.ptr, pool.lo_ptr, n, side, side, resample, act)
        elif lo0 is not None or lo1 is not None:
            self._rec("ivid_gn_apply_c", self.dtype, x0.ptr, lo0, c0, x1.ptr if x1 is not None else None, lo1, c1,
                      ab.data_ptr(), y.ptr, n, side, side, resample, act)
        else:
            self._rec("ivid_gn_apply", self.dtype, x0.ptr, c0, x1.ptr if x1 is not None else None, c1, ab.data_ptr(), y.ptr, n,
                      side, side, resample, act)
        self.arena.put(ab)
        return y

    def _conv3_gn(self, x0: _Act, x1, ab, up, wname, out: _Act, res_ptr, res_mode, skip=None, res_lo=None):
        """Fused GroupNorm-apply + SiLU (+ x2 upsample) + conv3x3 (csrc/conv3x3_fused.hip).  skip = (s0, s1, wname): the
        ResBlock's 1x1 skip_connection on its raw input cat(s0, s1), accumulated in the same kernel.  The halo transform and
        the skip phase's raw inputs are MFMA operands (hi planes); the halo transform starts from hi + lo where a source has a
        lo plane; out / res may carry lo planes."""
        out.stats_blk = 128
        st = out.stats.data_ptr() if out.stats is not None else None
        lo0, lo1 = (x0.lo_ptr, (x1.lo_ptr if x1 is not None else None)) if self.comp_in else (None, None)
        if out.lo is not None or res_lo is not None or lo0 is not None or lo1 is not None:
            bias = self.w[wname + ".bias"]
            s0 = s1 = sname = None
            if skip is not None:
                s0, s1, sname = skip
                key = wname + "+" + sname
                if key not in self._sum_bias:
                    self._sum_bias[key] = (self.w[wname + ".bias"] + self.w[sname + ".bias"]).contiguous()
                bias = self._sum_bias[key]
            args = (self.dtype, x0.ptr, lo0, x0.c, x1.ptr if x1 is not None else None, lo1,
                    x1.c if x1 is not None else 0, ab.data_ptr(), 1 if up else 0, self.w[wname + ".weight"].data_ptr(),
                    bias.data_ptr(), out.ptr, out.lo_ptr, res_ptr, res_lo, res_mode, out.n, out.side, out.side, out.c, st,
                    s0.ptr if s0 is not None else None, s0.c if s0 is not None else 0,
                    s1.ptr if s1 is not None else None, s1.c if s1 is not None else 0,
                    self.w[sname + ".weight"].data_ptr() if sname is not None else None)
            if skip is not None and self.split_skip:   # x_hi.w_hi + x_lo.w_hi + x_hi.w_lo inside the kernel's skip phase
                assert s0.lo is not None and (s1 is None or s1.lo is not None)
                self._rec("ivid_conv3x3_gn_skip_s", *args, s0.lo_ptr, s1.lo_ptr if s1 is not None else None,
                          self.w[sname + ".weight_lo"].data_ptr())
            else:
                self._rec("ivid_conv3x3_gn_skip_c", *args)
            return
        if skip is None:
            self._rec("ivid_conv3x3_gn", self.dtype, x0.ptr, x0.c, x1.ptr if x1 is not None else None,
                      x1.c if x1 is not None else 0, ab.data_ptr(), 1 if up else 0, self.w[wname + ".weight"].data_ptr(),
                      self.w[wname + ".bias"].data_ptr(), out.ptr, res_ptr, res_mode, out.n, out.side, out.side, out.c, st)
            return
        s0, s1, sname = skip
        key = wname + "+" + sname
        if key not in self._sum_bias:   # conv bias + skip bias, added once in the epilogue
            self._sum_bias[key] = (self.w[wname + ".bias"] + self.w[sname + ".bias"]).contiguous()
        self._rec("ivid_conv3x3_gn_skip", self.dtype, x0.ptr, x0.c, x1.ptr if x1 is not None else None,
                  x1.c if x1 is not None else 0, ab.data_ptr(), 1 if up else 0, self.w[wname + ".weight"].data_ptr(),
                  self._sum_bias[key].data_ptr(), out.ptr, res_ptr, res_mode, out.n, out.side, out.side, out.c, st,
                  s0.ptr, s0.c, s1.ptr if s1 is not None else None, s1.c if s1 is not None else 0,
                  self.w[sname + ".weight"].data_ptr())

    # ---- ops ----
    def _res(self, op: Res, x: _Act, skip):
        n = x.n
        resample = {"same": 0, "up": 1, "down": 2}[op.mode]
        so = op.res_out
        # Cout > 128: the 8x32x256 shape of the fused kernel.  Cout <= 128 (the small / SR models' first levels): its 16x32x128
        # shape with 64-byte chunks (csrc/conv3x3_fused_body.h, FusedShape<false>)
        narrow_ok = self.fuse_narrow and so % 32 == 0
        fused2 = self.fuse_conv and so % 32 == 0 and (op.cout > 128 or narrow_ok)   # out_layers conv: input at the output size
        fused = fused2 and op.mode != "down"                             # in_layers conv: not behind the 2x2 average pool
        up4 = (op.mode == "up" and skip is None and x.side <= self.up4_max_side and (x.side * x.side) % 64 == 0 and op.cout > 32)
        # fp16cx: h1 (between the block's two convolutions) carries a lo plane too: out_layers' GroupNorm then
        # sees the unrounded in_layers result (not behind the phase-form up-convolution, whose epilogue scatters single planes)
        # First ResBlock of a stacked CFG forward (rows >= bsrc repeat x and t with the null class, classifier_free_guidance.py:39-42):
        # the stem output and this block's in_layers (GroupNorm without FiLM, SiLU, conv) do not depend on the class, so rows
        # bsrc.. would recompute rows 0..bsrc-1 bit for bit: h1 exists for ONE half only
        share = (fused and self.share_cfg and not self._first_res_done and self.n == 2 * self.bsrc and skip is None and not self.debug
                 and not op.has_skip_conv)
        h1 = self._new(self.bsrc if share else n, so, op.cout, stats=True, trunk=self.comp_in and not up4)
        xpool = None
        if (op.mode == "up" and skip is None and x.side <= self.up4_max_side and (x.side * x.side) % 64 == 0
                and op.cout > 32):
            # activated tensor at the SOURCE size (a quarter of the bytes of the upsampled one), then the phase convolution
            act1 = self._gn(x, None, op.prefix + ".in_layers.0", None, 0, 1)
            h1.stats_blk = 64
            self._rec("ivid_conv3x3_up", self.dtype, act1.ptr, op.cin, None, 0,
                      self.w[op.prefix + ".in_layers.2.weight_up4"].data_ptr(), self.w[op.prefix + ".in_layers.2.bias"].data_ptr(),
                      h1.ptr, n, x.side, x.side, op.cout, self.tile_cfg or self._tile_up4,
                      h1.stats.data_ptr() if h1.stats is not None else None)
            self._free(act1)
        elif share:
            # the convolution runs on the first half of x; out_layers below reads its result for BOTH halves (round 5: two
            # half-batch launches with the same h1 pointer instead of duplicating h1 and its statistics with ivid_copy)
            xh = _Act(x.buf, self.bsrc, x.side, x.c, x.stats, lo=x.lo)
            xh.stats_blk = x.stats_blk
            ab1 = self._gn_coeffs(xh, None, op.prefix + ".in_layers.0", None)
            self._conv3_gn(xh, None, ab1, False, op.prefix + ".in_layers.2", h1, None, 0)
            self.arena.put(ab1)
        elif fused:
            ab1 = self._gn_coeffs(x, skip, op.prefix + ".in_layers.0", None)
            self._conv3_gn(x, skip, ab1, op.mode == "up", op.prefix + ".in_layers.2", h1, None, 0)
            self.arena.put(ab1)
        else:
            # `down` block in a 16-bit mode: the same pass also emits x_upd(x) = the 2x2 average of the raw input, the residual of
            # the block's second convolution (a same-size residual instead of four full-size pixels per output in its epilogue)
            if op.mode == "down" and skip is None and not op.has_skip_conv and self.esz == 2 and self.pool_res:
                xpool = self._new(n, so, op.cin, trunk=True)
            act1 = self._gn(x, skip, op.prefix + ".in_layers.0", None, resample, 1, pool=xpool)
            self._conv(self.dtype, act1.ptr, op.cin, None, 0, op.prefix + ".in_layers.2", h1.ptr, None, 0, 0, n, so, so,
                       op.cout, 9, out_act=h1, out_lo=h1.lo_ptr)
            self._free(act1)
        if share:
            return self._res_out_shared(op, x, h1, so)
        if fused2:
            ab2 = self._gn_coeffs(h1, None, op.prefix + ".out_layers.0", op.emb_off)
        else:
            act2 = self._gn(h1, None, op.prefix + ".out_layers.0", op.emb_off, 0, 1)
            self._free(h1)
        self._first_res_done = True
        out = self._new(n, so, op.cout, stats=True, trunk=True)
        kstep = 128 // self.esz
        if op.cout <= 128:
            kstep //= 2                                       # the 128-wide variant works on 64-byte chunks
        if (fused2 and op.has_skip_conv and self.fuse_skip and x.c % kstep == 0
                and (skip is None or skip.c % kstep == 0)):
            # 1x1 skip_connection folded into the out_layers conv kernel as extra K-steps (no separate launch, no
            # residual round trip)
            assert op.mode == "same"
            self._conv3_gn(h1, None, ab2, False, op.prefix + ".out_layers.3", out, None, 0,
                           skip=(x, skip, op.prefix + ".skip_connection"))
            self.arena.put(ab2)
            self._free(h1)
            return out
        if op.has_skip_conv:
            assert op.mode == "same"
            r = self._new(n, so, op.cout, trunk=True)   # skip_connection(x) is a term of the residual stream
            sc = skip.c if skip is not None else 0
            sname = op.prefix + ".skip_connection"
            self._conv(self.dtype, x.ptr, x.c, skip.ptr if skip is not None else None, sc,
                       sname, r.ptr, None, 0, 0, n, so, so, op.cout, 1, out_lo=r.lo_ptr)
            if self.split_skip:
                # split precision as three chained launches (each result is carried as hi + lo, 2^-22): + x_lo.w_hi, + x_hi.w_lo
                assert x.lo is not None and (skip is None or skip.lo is not None)
                r2 = self._new(n, so, op.cout, trunk=True)
                self._conv(self.dtype, x.lo_ptr, x.c, skip.lo_ptr if skip is not None else None, sc, sname, r2.ptr, r.ptr, 1, 0,
                           n, so, so, op.cout, 1, out_lo=r2.lo_ptr, res_lo=r.lo_ptr, no_bias=True)
                self._conv(self.dtype, x.ptr, x.c, skip.ptr if skip is not None else None, sc, sname, r.ptr, r2.ptr, 1, 0,
                           n, so, so, op.cout, 1, out_lo=r.lo_ptr, res_lo=r2.lo_ptr, wkey=".weight_lo", no_bias=True)
                self._free(r2)
            res_ptr, res_lo, res_mode = r.ptr, r.lo_ptr, 1
        else:
            assert skip is None
            r = xpool            # freed with the skip-conv temporary's code path below
            if xpool is not None:
                res_ptr, res_lo, res_mode = xpool.ptr, xpool.lo_ptr, 1
            else:
                res_ptr, res_lo, res_mode = x.ptr, x.lo_ptr, {"same": 1, "up": 2, "down": 3}[op.mode]
        if fused2 and self.dtype == _lib.BF16X3 and self._main_mode[0] != _lib.BF16X3 and self.island_o16:
            # island of the fp16s mode: the block output also leaves as fp16 hi + lo planes (what everything outside the island
            # reads); the LAST island block's fp32 form has no reader at all
            out.twin = self._new16(n, so, op.cout)
            out.stats_blk = 128
            keep32 = op.prefix != self._island_last
            self._rec("ivid_conv3x3_gn_o16", h1.ptr, h1.c, None, 0, ab2.data_ptr(), self.w[op.prefix + ".out_layers.3.weight"].data_ptr(),
                      self.w[op.prefix + ".out_layers.3.bias"].data_ptr(), out.ptr if keep32 else None, out.twin.ptr, out.twin.lo_ptr,
                      res_ptr, res_mode, n, so, so, op.cout, out.stats.data_ptr() if out.stats is not None else None)
            self.arena.put(ab2)
            self._free(h1)
        elif fused2:
            self._conv3_gn(h1, None, ab2, False, op.prefix + ".out_layers.3", out, res_ptr, res_mode, res_lo=res_lo)
            self.arena.put(ab2)
            self._free(h1)
        else:
            self._conv(self.dtype, act2.ptr, op.cout, None, 0, op.prefix + ".out_layers.3", out.ptr, res_ptr, res_mode, 0,
                       n, so, so, op.cout, 9, out_act=out, out_lo=out.lo_ptr, res_lo=res_lo)
            self._free(act2)
        if r is not None:
            self._free(r)
        return out

    def _res_out_shared(self, op: Res, x: _Act, h1: _Act, so):
        """out_layers of the first ResBlock of a stacked CFG forward (same-size block without skip convolution, fused kernels):
        h1 holds the class-independent in_layers result of ONE half; each half of the batch is its own launch -- GroupNorm + FiLM
        coefficients from that half's class embedding rows, residual and output at that half's offset -- reading the same h1.
        Bit-identical to one full-batch launch on a duplicated h1 (a tile's arithmetic does not know its image index)."""
        assert op.mode == "same" and not op.has_skip_conv and h1.n == self.bsrc and x.c == op.cout and x.side == so
        half, n = self.bsrc, self.n
        self._first_res_done = True
        out = self._new(n, so, op.cout, stats=True, trunk=True)
        out.stats_blk = 128
        island = self.dtype == _lib.BF16X3 and self._main_mode[0] != _lib.BF16X3 and self.island_o16
        keep32 = True
        if island:
            out.twin = self._new16(n, so, op.cout)
            keep32 = op.prefix != self._island_last
        px = so * so * op.cout                                   # elements per image

        def view(a, hf, esz):
            """half hf of activation a as its own _Act (buffer views: the arena only ever sees the parents)"""
            off = hf * half * px * esz
            v = _Act(a.buf[off:], half, a.side, a.c, None, lo=a.lo)
            v.stats_blk = a.stats_blk
            if a.stats is not None:
                v.stats = a.stats[hf * half * (so * so // a.stats_blk) * a.c * 2 * 4:]
            return v
        for hf in (0, 1):
            ab2 = self._gn_coeffs(h1, None, op.prefix + ".out_layers.0", op.emb_off, film_row0=hf * half)
            xv, ov = view(x, hf, self.esz), view(out, hf, self.esz)
            if island:
                tv = view(out.twin, hf, 2)
                self._rec("ivid_conv3x3_gn_o16", h1.ptr, h1.c, None, 0, ab2.data_ptr(), self.w[op.prefix + ".out_layers.3.weight"].data_ptr(),
                          self.w[op.prefix + ".out_layers.3.bias"].data_ptr(), ov.ptr if keep32 else None, tv.ptr, tv.lo_ptr,
                          xv.ptr, 1, half, so, so, op.cout, ov.stats.data_ptr() if ov.stats is not None else None)
            else:
                self._conv3_gn(h1, None, ab2, False, op.prefix + ".out_layers.3", ov, xv.ptr, 1, res_lo=xv.lo_ptr)
            self.arena.put(ab2)
        self._free(h1)
        return out

    def _attn(self, op: Attn, x: _Act):
        n, side, c = x.n, x.side, x.c
        # (reading the hi plane alone here saves 0.4 ms per large forward and costs nothing on the large model, but it takes
        # small-128 from 9.2e-4 to 9.7e-4 of the reference: the lo plane stays)
        xn = self._gn(x, None, op.prefix + ".norm", None, 0, 0)
        qkv = self._new(n, side, 3 * c)
        self._conv(self.dtype, xn.ptr, c, None, 0, op.prefix + ".qkv", qkv.ptr, None, 0, 0, n, side, side, 3 * c, 1)
        self._free(xn)
        a = self._new(n, side, c)
        self._rec("ivid_attention", self.dtype, qkv.ptr, a.ptr, n, side * side, op.heads)
        self._free(qkv)
        out = self._new(n, side, c, stats=True, trunk=True)
        self._conv(self.dtype, a.ptr, c, None, 0, op.prefix + ".proj_out", out.ptr, x.ptr, 1, 0, n, side, side, c, 1,
                   out_act=out, out_lo=out.lo_ptr, res_lo=x.lo_ptr)
        self._free(a)
        return out

    def _build(self):
        sp, w, n = self.spec, self.w, self.n
        mc, ed = sp.model_channels, sp.emb_dim
        # ---- embeddings (adm.py:545-555 + every ResBlock's emb_layers, adm.py:176), fp32 ----
        pos = self._f32(n, mc)
        cls = self._f32(n, ed) if sp.num_classes is not None else None
        self._rec("ivid_embed_inputs", self.t_in.data_ptr(), self.c_in.data_ptr() if cls is not None else None,
                  self.bsrc, n, self.null_from, w["freqs"].data_ptr(), mc // 2,
                  w["label_emb"].data_ptr() if cls is not None else None, ed, pos.data_ptr(),
                  cls.data_ptr() if cls is not None else None)
        e1, s1, emb, semb = self._f32(n, ed), self._f32(n, ed), self._f32(n, ed), self._f32(n, ed)
        self.embproj = self._f32(n, sp.emb_total)
        self._linear(pos, mc, "time_embed.1", e1, ed)
        self._rec("ivid_silu_f32", e1.data_ptr(), s1.data_ptr(), n * ed)
        self._linear(s1, ed, "time_embed.3", emb, ed, res=cls)
        self._rec("ivid_silu_f32", emb.data_ptr(), semb.data_ptr(), n * ed)
        self._linear(semb, ed, "emb_all", self.embproj, sp.emb_total)
        # ---- stem ----
        S = sp.image_size
        island = sorted(w.island)
        if island:
            self._set_island(True)
            last = sp.stages[island[-1]].ops[-1]
            self._island_last = last.prefix if isinstance(last, Res) else None
        xin = self._new(n, S, w.stem_k)
        self._rec("ivid_stem_im2col_split" if self.comp else "ivid_stem_im2col", self.dtype, self.x_in.data_ptr(), self.bsrc, n,
                  sp.in_channels, S, S, w.stem_k, xin.ptr)
        h = self._new(n, S, sp.stem_out, stats=True, trunk=True)
        if island and self.island_o16:
            # the island's stem: fp32 for the island + its fp16 twin for the decoder's last level (through the skip stash)
            h.twin = self._new16(n, S, sp.stem_out)
            if h.stats is not None:
                h.stats_blk = self.lib.ivid_conv2d_stats_block(n, S, S, sp.stem_out, self.tile_cfg)
            self._rec("ivid_conv2d_o16", xin.ptr, w.stem_k, None, 0, self.w["input_blocks.0.0.weight"].data_ptr(),
                      self.w["input_blocks.0.0.bias"].data_ptr(), h.ptr, h.twin.ptr, h.twin.lo_ptr, None, 0, n, S, S, sp.stem_out, 1,
                      self.tile_cfg, h.stats.data_ptr() if h.stats is not None else None)
        else:
            self._conv(self.dtype, xin.ptr, w.stem_k, None, 0, "input_blocks.0.0", h.ptr, None, 0, 0, n, S, S, sp.stem_out, 1,
                       out_act=h, out_lo=h.lo_ptr)
        self._free(xin)
        self._tap("stem", h)
        stash = [h]
        # ---- encoder / bottleneck / decoder ----
        for si, st in enumerate(sp.stages):
            if st.conv_in:
                continue
            if island and si == island[-1] + 1:
                # leaving the island: its tensors (all of them are in the skip stash; the last one is also the next block's
                # input) change to the compensated 16-bit storage of the main mode
                self._set_island(False)
                stash = [self._to16(a) for a in stash]
                h = stash[-1]
            skip = stash.pop() if st.kind == "out" else None
            first = True
            for op in st.ops:
                prev = h
                if isinstance(op, Res):
                    h = self._res(op, prev, skip if first else None)
                else:
                    h = self._attn(op, prev)
                self._tap(op.prefix, h)
                # the stage input may still be referenced by the skip stash; intermediates are not
                if not any(prev is s for s in stash):
                    self._free(prev)
                if first and skip is not None:
                    self._free(skip)
                first = False
            if st.kind == "in":
                stash.append(h)
        assert not stash
        # ---- head: GN + SiLU + conv3x3 -> fp32 NCHW (adm.py:565-566) ----
        # (the split form of the compensated modes takes 1..8 output channels, the plain head 1..16)
        if (self.fuse_conv and S % 32 == 0 and sp.out_channels <= (8 if self.comp else 16) and sp.final_c % (128 // self.esz) == 0
                and os.environ.get("IVID_NO_FUSED_HEAD", "0") != "1"):
            ab = self._gn_coeffs(h, None, "out.0", None)      # one kernel: the input is read once
            if self.comp:
                self._rec("ivid_conv3x3_gn_out_c", self.dtype, h.ptr, h.lo_ptr, sp.final_c, ab.data_ptr(),
                          self.w["out.2.weight"].data_ptr(), self.w["out.2.weight_lo"].data_ptr(),
                          self.w["out.2.bias"].data_ptr(), self.out.data_ptr(), n, S, S, sp.out_channels)
            else:
                self._rec("ivid_conv3x3_gn_out", self.dtype, h.ptr, sp.final_c, ab.data_ptr(), self.w["out.2.weight"].data_ptr(),
                          self.w["out.2.bias"].data_ptr(), self.out.data_ptr(), n, S, S, sp.out_channels)
            self.arena.put(ab)
            self._free(h)
        else:
            act = self._gn(h, None, "out.0", None, 0, 1)
            self._free(h)
            self._conv(_lib.F32 if self.dtype == _lib.BF16X3 else self.dtype, act.ptr, sp.final_c, None, 0, "out.2",
                       self.out.data_ptr(), None, 0, 1, n, S, S, sp.out_channels, 9)
            self._free(act)

    # ---- execution ----
    def _enqueue(self, stream):
        sp = C.c_void_p(stream)
        trace = os.environ.get("IVID_TRACE") == "1"
        for fn, name, args in self.launches:
            if fn is None:
                args()
                continue
            if trace:  # debugging aid: last line printed before a fault names the launch
                print("[ivid] launch", name, [a for a in args if not isinstance(a, int) or a < (1 << 32)], flush=True)
                st = fn(*args, sp)
                torch.cuda.synchronize(self.device)
            else:
                st = fn(*args, sp)
            if st != 0:
                _lib.check(st, name)

    def run(self, x, times, classes, use_graph=True):
        """x [bsrc,Cin,S,S] fp32, times [bsrc] int64, classes [bsrc] int64 or None -> self.out (static buffer)."""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        if self.program is not None:
            # the C call copies device -> device from raw pointers: everything must live on the plan's device
            if x.device != self.x_in.device or x.dtype != torch.float32 or x.shape != self.x_in.shape or not x.is_contiguous():
                raise ValueError(f"x must be a contiguous fp32 {tuple(self.x_in.shape)} tensor on {self.x_in.device}, got "
                                 f"{x.dtype} {tuple(x.shape)} on {x.device}")
            if times.shape != (self.bsrc,) or (classes is not None and classes.shape != (self.bsrc,)):
                raise ValueError(f"times / classes must have shape ({self.bsrc},)")
            t64 = times.to(device=self.x_in.device, dtype=torch.int64).contiguous()
            c64 = (classes.to(device=self.x_in.device, dtype=torch.int64).contiguous()
                   if (classes is not None and self.spec.num_classes is not None) else None)
            with torch.cuda.stream(self.stream):
                _lib.call("ivid_unet_forward", self.program, x.data_ptr(), t64.data_ptr(), c64.data_ptr() if c64 is not None else None,
                          None, 1 if use_graph else 0, C.c_void_p(self.stream.cuda_stream))
                for tns in (x, t64, c64):            # their memory must outlive the copies enqueued on the plan's stream
                    if tns is not None:
                        tns.record_stream(self.stream)
            cur.wait_stream(self.stream)
            return self.out
        with torch.cuda.stream(self.stream):
            self.x_in.copy_(x)
            self.t_in.copy_(times)
            if self.spec.num_classes is not None:
                if classes is None:
                    self.c_in.fill_(-1)
                else:
                    self.c_in.copy_(classes)
            self.launch(use_graph)
        cur.wait_stream(self.stream)
        return self.out

    def launch(self, use_graph=True):
        """Enqueue one forward on self.stream from the static input buffers (no host sync)."""
        stream = self.stream.cuda_stream
        if self.program is not None:
            _lib.call("ivid_program_launch", self.program, 1 if use_graph else 0, C.c_void_p(stream))
            return
        if self.debug or not use_graph or not self.warm:
            self._enqueue(stream)   # first run is eager: sets kernel attributes, creates the zero page
            self.warm = True
            return
        if self.graph is None:
            _lib.call("ivid_graph_begin", C.c_void_p(stream))
            try:
                self._enqueue(stream)
            finally:
                gh = C.c_void_p()
                st = self.lib.ivid_graph_end(C.c_void_p(stream), C.byref(gh))
            _lib.check(st, "ivid_graph_end")
            self.graph = gh
        _lib.call("ivid_graph_launch", self.graph, C.c_void_p(stream))

    def export_engine(self):
        """This plan as an engine file (bytes) for `ivid_unet_load`: a non-Python host runs the forward through the C ABI alone
        (diffusion/backbones/engine.py)."""
        from .engine import export_engine
        return export_engine(self)

    def profile_eager(self):
        """One eager forward with a HIP-event pair around EVERY launch (on the plan's stream, where the
        kernels run).  Returns [(c_abi_name, args, milliseconds)] — bench.py derives the per-kernel-family
        roofline numbers from it; rocprofv3 --kernel-trace --stats must agree (profiles/)."""
        stream = C.c_void_p(self.stream.cuda_stream)
        evs = []
        torch.cuda.synchronize(self.device)
        with torch.cuda.stream(self.stream):
            for fn, name, args in self.launches:
                if fn is None:
                    continue
                e0, e1 = C.c_void_p(), C.c_void_p()
                _lib.call("ivid_event_create", C.byref(e0))
                _lib.call("ivid_event_create", C.byref(e1))
                _lib.call("ivid_event_record", e0, stream)
                _lib.check(fn(*args, stream), name)
                _lib.call("ivid_event_record", e1, stream)
                evs.append((name, args, e0, e1))
        out = []
        for name, args, e0, e1 in evs:
            ms = C.c_float()
            _lib.call("ivid_event_elapsed_ms", e0, e1, C.byref(ms))
            out.append((name, args, ms.value))
            _lib.call("ivid_event_destroy", e0)
            _lib.call("ivid_event_destroy", e1)
        return out

    @property
    def has_graph(self):
        if self.program is not None:
            return bool(self.lib.ivid_program_has_graph(self.program))
        return self.graph is not None

    def __del__(self):
        try:
            if self.graph is not None:
                self.lib.ivid_graph_destroy(self.graph)
            if getattr(self, "program", None) is not None:
                self.lib.ivid_program_destroy(self.program)
        except Exception:
            pass
