"""Engine files (ivid_amd/diffusion/backbones/engine.py, include/ivid_hip.h ivid_unet_load): the exporter is host logic and the
loader's validation runs before it touches a device, so both are covered without a GPU.  The plan is built on torch's CPU device
(real host pointers): every pointer argument must relocate to the byte it pointed at."""
import ctypes
import struct

import pytest
import torch

import common as C
from ivid_amd import _lib
from ivid_amd.diffusion.backbones import engine as E


@pytest.fixture()
def cpu_plan(monkeypatch):
    from ivid_amd.diffusion.backbones import plan as P
    from ivid_amd.diffusion.backbones.spec import build_spec

    class FakeStream:
        def __init__(self, device=None):
            self.cuda_stream = 0
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setenv("IVID_PY_LAUNCH", "1")

    def make(args, precision, batch=2, stacked=True):
        spec = build_spec(**args)
        g = torch.Generator().manual_seed(1)
        sd = {k: torch.randn(s, generator=g) for k, s in C.schema_for(args)}
        w = P.PackedWeights(spec, sd, "cpu", _lib.PRECISIONS[precision], comp=_lib.COMPENSATED.get(precision, 0))
        return P.UNetPlan(spec, w, "cpu", batch, stacked)
    return make


@pytest.mark.parametrize("precision", ["bf16", "fp16s", "bf16x3"])
def test_export_relocates_every_pointer_to_the_byte_it_pointed_at(cpu_plan, precision):
    pl = cpu_plan(C.MINI, precision)
    blob = pl.export_engine()
    head, bufs, ops = E.parse_engine(blob)
    assert head["batch"] == 2 and head["out_rows"] == 4 and head["has_classes"]
    assert (head["in_channels"], head["out_channels"], head["image_size"]) == (4, 4, 32)
    assert head["x_bytes"] == pl.x_in.numel() * 4 and head["out_bytes"] == pl.out.numel() * 4
    assert len(ops) == len(pl.launches)
    where = {}          # original address -> (buffer, offset): one-to-one
    n_const = 0
    for (fn, name, args), (code, dec) in zip(pl.launches, ops):
        assert code == _lib.OP_CODES[name] and len(dec) == len(args)
        sig = _lib.SIGNATURES[name][1][:-1]
        for ty, v, d in zip(sig, args, dec):
            if ty is ctypes.c_float:
                assert d == ("f", float(v))
            elif ty is ctypes.c_void_p:
                if v is None or int(v) == 0:
                    assert d == ("null",)
                    continue
                assert d[0] == "p"
                b = bufs[d[1]]
                assert d[2] < max(b["nbytes"], 1)
                assert where.setdefault(int(v), d[1:]) == d[1:]
                if b["kind"] == 1:      # a constant: the file holds the bytes the launch would have read
                    n = min(64, b["nbytes"] - d[2])
                    assert blob[b["offset"] + d[2]: b["offset"] + d[2] + n] == ctypes.string_at(int(v), n)
                    n_const += 1
            else:
                assert d == ("i", int(v))
    assert len(set(where.values())) == len(where) and n_const > 100
    # boundary buffers are scratch and distinct
    assert len({head["x_in"], head["t_in"], head["c_in"], head["out"]}) == 4
    assert all(bufs[head[k]]["kind"] == 0 for k in ("x_in", "t_in", "c_in", "out"))
    # constants: 16-byte aligned, inside the file, not overlapping
    spans = sorted((b["offset"], b["nbytes"]) for b in bufs if b["kind"] == 1)
    assert all(o % 16 == 0 and o + n <= len(blob) for o, n in spans)
    assert all(a[0] + a[1] <= b[0] for a, b in zip(spans, spans[1:]))


def test_export_of_a_model_without_classes(cpu_plan):
    pl = cpu_plan(C.MINI_UNCLASS, "bf16", batch=3, stacked=False)
    head, bufs, ops = E.parse_engine(pl.export_engine())
    assert head["c_in"] is None and not head["has_classes"] and head["batch"] == head["out_rows"] == 3


def _load(blob):
    buf = (ctypes.c_ubyte * len(blob)).from_buffer_copy(blob)
    h = ctypes.c_void_p()
    rc = _lib.load().ivid_unet_load(ctypes.cast(buf, ctypes.c_void_p), len(blob), ctypes.byref(h))
    msg = _lib.load().ivid_last_error()
    return rc, (msg or b"").decode(), h


def test_loader_rejects_malformed_files_before_touching_a_device(cpu_plan):
    blob = bytearray(cpu_plan(C.MINI, "fp16s").export_engine())
    head, bufs, ops = E.parse_engine(bytes(blob))

    def bad(b, what):
        rc, msg, h = _load(bytes(b))
        assert rc != 0 and what in msg, (rc, msg)
    bad(b"NOTANENG" + bytes(blob[8:]), "magic")
    bad(blob[:40], "bad arguments")
    bad(blob[:200], "malformed")
    hs = struct.calcsize(E.HEAD)
    ops_at = 8 + hs + 4 + 17 * len(bufs) + 4
    bad(blob[:ops_at + 100], "outside the file")                # file ends inside the op list: the constants are checked first
    b = bytearray(blob); struct.pack_into("<I", b, ops_at - 4, len(ops) + 5); bad(b, "")   # more ops than the list holds: runs into the constants' bytes
    b = bytearray(blob); struct.pack_into("<I", b, 8 + hs - 4, len(bufs)); bad(b, "boundary buffer index")
    b = bytearray(blob); struct.pack_into("<Q", b, 8 + 28, head["x_bytes"] * 2); bad(b, "disagree")
    # a file written against other entry-point signatures (another IVID_ENGINE_ABI), or in an older container format
    assert head["abi"] == _lib.ENGINE_ABI
    b = bytearray(blob); struct.pack_into("<I", b, 8, _lib.ENGINE_ABI + 1); bad(b, "IVID_ENGINE_ABI")
    b = bytearray(blob); b[7:8] = b"1"; bad(b, "format version")
    # an op whose argument count is not its entry point's (run_op reads fixed slots): first op of the list
    b = bytearray(blob); struct.pack_into("<I", b, ops_at + 4, len(ops[0][1]) - 1); bad(b, "argument count")
    b = bytearray(blob); struct.pack_into("<Q", b, 8 + hs + 4 + 1, 16); bad(b, "")   # first buffer shrunk: some pointer / boundary falls outside
    # a constant whose data would lie past the end of the file
    k = next(i for i, x in enumerate(bufs) if x["kind"] == 1)
    b = bytearray(blob); struct.pack_into("<Q", b, 8 + hs + 4 + 17 * k + 9, len(blob) - 8); bad(b, "outside the file")
    # a pointer argument beyond its buffer
    p = ops_at
    done = False
    for code, dec in ops:
        p += 8
        for d in dec:
            if d[0] == "p" and not done:
                b = bytearray(blob); struct.pack_into("<Q", b, p + 1 + 8, 1 << 50); bad(b, "outside its buffer")
                b = bytearray(blob); struct.pack_into("<I", b, p + 1, len(bufs) + 7); bad(b, "outside its buffer")
                b = bytearray(blob); b[p] = 9; bad(b, "unknown argument tag")
                done = True
            p += 17 if d[0] == "p" else 9
    assert done


def test_loading_a_valid_file_without_a_gpu_fails_loudly(cpu_plan):
    rc, msg, h = _load(cpu_plan(C.MINI, "bf16").export_engine())
    if rc == 0:                                  # a GPU is present: the load is real
        assert _lib.load().ivid_program_destroy(h) == 0
    else:
        assert "hipMalloc" in msg
