"""Helpers for the -m gpu parity tests: NCHW<->NHWC staging, raw C-ABI calls, a JSON error report."""
import ctypes as C
import json
import os

import torch

import common

REPORT = os.path.join(common.ROOT, "gpurun_out", "parity_report.json")
_report = {}


def report(key, **vals):
    def plain(v):
        try:
            return float(v)
        except (TypeError, ValueError):
            return str(v)
    _report[key] = {k: plain(v) for k, v in vals.items()}
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        old = {}
        if os.path.exists(REPORT):
            try:
                old = json.load(open(REPORT))
            except Exception:
                old = {}
        old.update(_report)
        json.dump(old, open(REPORT, "w"), indent=1, sort_keys=True)
    except Exception:
        pass


def lib():
    from ivid_amd import _lib
    return _lib


# include/ivid_hip.h dtype codes: IVID_F32, IVID_BF16, IVID_F16, IVID_BF16X3 (fp32 storage, split-bf16 MFMA operands)
DN = {0: "f32", 1: "bf16", 2: "f16", 3: "bf16x3"}
_TDT = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16, 3: torch.float32}


def tdt(dtype):
    """Storage dtype of activations."""
    return _TDT[dtype]


def pack_w(w2d, dtype):
    """fp32 [Cout, K] (cpu) -> the weight operand the C ABI expects, on the GPU.  Written independently of the product's
    packer (plan.split_pack) so that the IVID_BF16X3 layout of include/ivid_hip.h is pinned by the tests:
    per 8 input channels 8 x bf16 hi followed by 8 x bf16 lo, hi = bf16(w), lo = bf16(w - hi)."""
    w2d = w2d.contiguous().float()
    if dtype != 3:
        return w2d.to("cuda", tdt(dtype))
    cout, k = w2d.shape
    assert k % 8 == 0
    out = torch.empty(cout, k // 8, 16, dtype=torch.bfloat16)
    hi = w2d.bfloat16()
    lo = (w2d - hi.float()).bfloat16()
    out[:, :, :8] = hi.view(cout, k // 8, 8)
    out[:, :, 8:] = lo.view(cout, k // 8, 8)
    return out.reshape(cout, 2 * k).cuda()


def to_nhwc(x, dtype):
    """NCHW fp32 (cpu) -> NHWC dtype on the GPU."""
    return x.permute(0, 2, 3, 1).contiguous().to("cuda", tdt(dtype))


def from_nhwc(y):
    return y.float().permute(0, 3, 1, 2).contiguous().cpu()


def rounded(x, dtype):
    """What the kernel actually sees: fp32 value of x after rounding to the compute dtype."""
    return x.to(tdt(dtype)).float()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def tol(dtype, f32=2e-5, bf16=6e-3, f16=None, x3=None):
    """Per-op bar: fp32 round-off; 16-bit modes vs fp32 math on inputs rounded to the storage type (f16 has 3 more mantissa
    bits than bf16); bf16x3 vs UNROUNDED fp32 inputs (its operands carry 16 mantissa bits: ~1e-5)."""
    return {0: f32, 1: bf16, 2: f16 if f16 is not None else bf16 / 6, 3: x3 if x3 is not None else max(f32, 4e-5)}[dtype]
