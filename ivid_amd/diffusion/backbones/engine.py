"""Engine file of a planned UNet forward: what a NON-PYTHON host needs to run `AdmUnet2d.forward` through the C ABI.

`AdmUnet2d.forward` is planned in Python (plan.py: which entry points, in which order, on which arena buffers, with which repacked
weights).  `export_engine(plan)` freezes one plan -- one (batch, stacked-CFG) shape of one model in one precision mode -- into a
relocatable blob: every buffer the launches touch (arena scratch, repacked weights with their bytes, the model-boundary inputs
and output), and the launch list with its pointer arguments written as (buffer, byte offset).  `ivid_unet_load` (csrc/program.hip)
allocates the buffers, uploads the weights, rebuilds the launch program and hands back the handle `ivid_unet_forward` takes:
planning happens once, offline; the serving process links libivid_hip.so and nothing else.

Layout (little endian):
    8 s   magic "IVIDENG2"
    u32   IVID_ENGINE_ABI of the library the launch list was recorded against (include/ivid_hip.h; any other value is refused)
    u32   batch (rows of x), u32 has_classes, u32 out_rows (batch, or 2 x batch for a stacked-CFG plan), u32 in_channels,
          u32 out_channels, u32 image_size, u64 x_bytes, u64 out_bytes, u32 buffer of x_in / t_in / c_in / out (c_in = 0xffffffff: none)
    u32   n_buffers;  per buffer: u8 kind (0 scratch, 1 constant), u64 nbytes, u64 data offset from the start of the blob (constants)
    u32   n_ops;      per op: u32 op code (IVID_OP_*), u32 nargs; per argument: u8 tag + 8 bytes
                      tag 0: int64 (8 bytes), 1: double (8), 2: pointer = u32 buffer, u32 pad, u64 byte offset (16), 3: NULL pointer (8 zero bytes)
    constants' bytes (16-byte aligned)
"""
import ctypes as C
import struct

import torch

from ... import _lib

MAGIC = b"IVIDENG2"
HEAD = "<IIIIIIIQQIIII"
NONE = 0xFFFFFFFF


def _align(n, a=16):
    return (n + a - 1) // a * a


def export_engine(plan):
    """-> bytes.  `plan`: a UNetPlan (any device: the constants are copied to the host)."""
    bufs = []          # (tensor, kind)

    def add(t, kind):
        if t is None:
            return
        for i, (u, _) in enumerate(bufs):
            if u.data_ptr() == t.data_ptr() and u.numel() * u.element_size() >= t.numel() * t.element_size():
                return
        bufs.append((t, kind))
    for b in plan.arena.all:
        add(b, 0)
    for t in plan._keep:
        add(t, 0)
    add(plan.x_in, 0); add(plan.t_in, 0); add(plan.c_in, 0); add(plan.out, 0)
    for t in plan.w.t.values():
        add(t, 1)
    for t in plan._sum_bias.values():
        add(t, 1)
    spans = sorted(((t.data_ptr(), t.numel() * t.element_size(), i) for i, (t, _) in enumerate(bufs)))

    def locate(ptr):
        for base, nb, i in spans:
            if base <= ptr < base + max(nb, 1):
                return i, ptr - base
        raise ValueError("export_engine: launch argument 0x%x points into no buffer of the plan" % ptr)

    def index_of(t):
        return locate(t.data_ptr())[0]
    ops = bytearray()
    nops = 0
    for fn, name, args in plan.launches:
        if fn is None:
            continue
        sig = _lib.SIGNATURES[name][1][:-1]
        assert len(sig) == len(args), name
        ops += struct.pack("<II", _lib.OP_CODES[name], len(args))
        for ty, v in zip(sig, args):
            if ty is C.c_float:
                ops += struct.pack("<Bd", 1, float(v))
            elif ty is C.c_void_p:
                if v is None or int(v) == 0:
                    ops += struct.pack("<Bq", 3, 0)
                else:
                    i, off = locate(int(v))
                    ops += struct.pack("<BIIQ", 2, i, 0, off)
            else:
                ops += struct.pack("<Bq", 0, int(v))
        nops += 1
    has_cls = plan.spec.num_classes is not None
    head = bytearray(MAGIC)
    sp = plan.spec
    head += struct.pack(HEAD, _lib.ENGINE_ABI, plan.bsrc, 1 if has_cls else 0, plan.n, sp.in_channels, sp.out_channels, sp.image_size,
                        plan.x_in.numel() * 4, plan.out.numel() * 4,
                        index_of(plan.x_in), index_of(plan.t_in), index_of(plan.c_in) if has_cls else NONE, index_of(plan.out))
    table_len = 4 + len(bufs) * 17
    data_start = _align(len(head) + table_len + 4 + len(ops))
    table = bytearray(struct.pack("<I", len(bufs)))
    blobs, off = [], data_start
    for t, kind in bufs:
        nb = t.numel() * t.element_size()
        if kind == 1:
            table += struct.pack("<BQQ", 1, nb, off)
            raw = t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes() if nb else b""
            blobs.append((off, raw))
            off = _align(off + nb)
        else:
            table += struct.pack("<BQQ", 0, nb, 0)
    out = bytearray(head + table + struct.pack("<I", nops) + ops)
    out += b"\0" * (data_start - len(out))
    for o, raw in blobs:
        assert len(out) == o
        out += raw
        out += b"\0" * (_align(len(out)) - len(out))
    return bytes(out)


def parse_engine(blob):
    """The inverse (test / inspection aid): header dict, buffer table, ops with decoded arguments."""
    assert blob[:8] == MAGIC, "not an ivid engine file"
    p = 8
    abi, batch, has_cls, rows, cin, cout, size, xb, ob, ix, it, ic, io = struct.unpack_from(HEAD, blob, p)
    p += struct.calcsize(HEAD)
    (nb,) = struct.unpack_from("<I", blob, p)
    p += 4
    bufs = []
    for _ in range(nb):
        kind, nbytes, off = struct.unpack_from("<BQQ", blob, p)
        p += 17
        bufs.append(dict(kind=kind, nbytes=nbytes, offset=off))
    (nops,) = struct.unpack_from("<I", blob, p)
    p += 4
    ops = []
    for _ in range(nops):
        code, nargs = struct.unpack_from("<II", blob, p)
        p += 8
        args = []
        for _ in range(nargs):
            tag = blob[p]
            p += 1
            if tag == 0:
                args.append(("i", struct.unpack_from("<q", blob, p)[0])); p += 8
            elif tag == 1:
                args.append(("f", struct.unpack_from("<d", blob, p)[0])); p += 8
            elif tag == 2:
                b, _pad, off = struct.unpack_from("<IIQ", blob, p); p += 16
                args.append(("p", b, off))
            else:
                args.append(("null",)); p += 8
        ops.append((code, args))
    return dict(abi=abi, batch=batch, has_classes=bool(has_cls), out_rows=rows, in_channels=cin, out_channels=cout, image_size=size, x_bytes=xb, out_bytes=ob, x_in=ix, t_in=it, c_in=None if ic == NONE else ic, out=io), bufs, ops


class Engine:
    """A loaded engine file, driven from Python the way a C host drives it (ivid_unet_load / ivid_unet_forward): no model object,
    no weights in torch.  `forward` mirrors `AdmUnet2d.forward` for the frozen (batch, stacked) shape; for a stacked plan it
    returns the 2 x batch rows (conditional rows first)."""

    def __init__(self, blob):
        import numpy as np
        self._blob = np.frombuffer(bytes(blob), dtype=np.uint8)
        self._h = C.c_void_p()
        _lib.call("ivid_unet_load", self._blob.ctypes.data, self._blob.size, C.byref(self._h))
        b, hc, xb, ob, dims = C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong(), (C.c_int * 4)()
        _lib.call("ivid_unet_info", self._h, C.byref(b), C.byref(hc), C.byref(xb), C.byref(ob), dims)
        self.batch, self.has_classes, self.x_bytes, self.out_bytes = b.value, bool(hc.value), xb.value, ob.value
        self.out_rows, self.in_channels, self.out_channels, self.image_size = list(dims)
        self._blob = None
        self._stream = torch.cuda.Stream()

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            return cls(f.read())

    @torch.no_grad()
    def forward(self, x, times, classes=None, use_graph=True):
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (self.batch, self.in_channels, self.image_size, self.image_size), \
            "engine: x must be the fp32 [batch, C, S, S] tensor the plan was frozen for"
        x, times = x.contiguous(), times.to(torch.int64).contiguous()
        cls = None if classes is None or not self.has_classes else classes.to(torch.int64).contiguous()
        out = torch.empty(self.out_bytes // 4, dtype=torch.float32, device=x.device)
        self._stream.wait_stream(torch.cuda.current_stream())
        _lib.call("ivid_unet_forward", self._h, x.data_ptr(), times.data_ptr(), None if cls is None else cls.data_ptr(),
                  out.data_ptr(), 1 if use_graph else 0, self._stream.cuda_stream)
        torch.cuda.current_stream().wait_stream(self._stream)
        for t in (x, times, cls, out):
            if t is not None:
                t.record_stream(self._stream)
        return out.view(self.out_rows, self.out_channels, self.image_size, self.image_size)

    def close(self):
        if self._h:
            torch.cuda.synchronize()
            _lib.call("ivid_program_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
