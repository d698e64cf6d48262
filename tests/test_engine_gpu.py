"""Engine files on the GPU: a plan frozen by `AdmUnet2d.export_engine`, loaded by `ivid_unet_load` (no model object, no torch
weights) must reproduce the model's own forward BIT FOR BIT -- it replays the same launches on the same repacked weights -- both
through the Python handle and through the C host program examples/unet_engine_host.c (built by ivid_amd/build.py)."""
import os
import subprocess

import numpy as np
import pytest
import torch

import common as C

pytestmark = pytest.mark.gpu


def build(args, precision, seed=3):
    from ivid_amd.diffusion.backbones import AdmUnet2d
    m = AdmUnet2d(**args, precision=precision)
    m.load_state_dict(C.synth_weights(args, seed), strict=True)
    return m.cuda().eval()


def inputs(args, batch, seed=11):
    S = args["image_size"]
    x = C.seeded_randn(seed, batch, args["in_channels"], S, S).cuda()
    t = torch.tensor([(37 + 311 * i) % 1000 for i in range(batch)], dtype=torch.long).cuda()
    cls = None if args.get("num_classes") is None else torch.tensor([(3 + 5 * i) % args["num_classes"] for i in range(batch)]).cuda()
    return x, t, cls


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16s", "bf16x3"])
@pytest.mark.parametrize("cfg,stacked", [("MINI", False), ("MINI", True), ("MINI_COND", True), ("MINI_UNCLASS", False)])
def test_loaded_engine_reproduces_the_model_forward_bit_for_bit(cfg, stacked, precision):
    from ivid_amd.diffusion.backbones.engine import Engine
    args = getattr(C, cfg)
    m = build(args, precision)
    x, t, cls = inputs(args, 3)
    if stacked:
        want = torch.cat(m.forward_cfg(x, t, cls)).clone()
    else:
        want = m(x, t, cls)
    eng = Engine(m.export_engine(3, stacked))
    del m                                            # the engine owns everything it needs
    torch.cuda.synchronize()
    assert (eng.batch, eng.out_rows, eng.has_classes) == (3, 6 if stacked else 3, cls is not None)
    for _ in range(3):                               # eager, capture, hipGraph replay
        got = eng.forward(x, t, cls)
        assert torch.equal(got, want)
    if cls is not None and not stacked:              # classes = NULL is the null class for every row
        m2 = build(args, precision)
        null = m2(x, t, torch.full_like(cls, -1))
        assert torch.equal(eng.forward(x, t, None), null)
    eng.close()


def test_c_host_program_runs_an_engine_file_without_python(tmp_path):
    from ivid_amd import build as B
    if not os.path.exists(B.HOST_BIN):
        raise RuntimeError(f"{B.HOST_BIN} missing: python -m ivid_amd.build")
    args = C.MINI128
    m = build(args, "fp16s")
    x, t, cls = inputs(args, 2)
    want = torch.cat(m.forward_cfg(x, t, cls)).cpu().numpy()
    m.export_engine(2, True, path=str(tmp_path / "unet.eng"))
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(x.cpu().numpy().tobytes() + t.cpu().numpy().astype(np.int64).tobytes() + cls.cpu().numpy().astype(np.int64).tobytes())
    r = subprocess.run([B.HOST_BIN, str(tmp_path / "unet.eng"), str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "6"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "ms per forward" in r.stdout
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(want.shape)
    assert np.array_equal(got, want)
    # a wrong-sized input file is refused by the host, a corrupted engine by the library
    with open(tmp_path / "short.bin", "wb") as f:
        f.write(b"\0" * 100)
    assert subprocess.run([B.HOST_BIN, str(tmp_path / "unet.eng"), str(tmp_path / "short.bin"), str(tmp_path / "o2.bin")],
                          capture_output=True, timeout=300).returncode == 2
    blob = bytearray(open(tmp_path / "unet.eng", "rb").read())
    blob[3] ^= 0xFF
    open(tmp_path / "bad.eng", "wb").write(bytes(blob))
    r = subprocess.run([B.HOST_BIN, str(tmp_path / "bad.eng"), str(tmp_path / "in.bin"), str(tmp_path / "o3.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 4 and "magic" in r.stderr
