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


def tdt(dtype):
    return torch.float32 if dtype == 0 else torch.bfloat16


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


def tol(dtype, f32=2e-5, bf16=6e-3):
    return f32 if dtype == 0 else bf16
