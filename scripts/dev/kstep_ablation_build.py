#!/usr/bin/env python
"""What bounds a K-step of conv3x3_fused_kernel's main loop?  Builds TIMING-ONLY variants of the library (results are wrong by
construction) from patched COPIES of csrc/ -- the product sources are not touched -- each with the phase time stamps of
scripts/dev/build_timeline.sh:
    base   the kernel as it is
    nodma  no weight LDS-DMA in the main loop (the MFMAs read stale slabs): what the weights' issue -> landed latency costs a step
    nohalo no halo pipeline of the next chunk (raw loads, GroupNorm + SiLU transform, LDS stores): what phase 1's own work costs
    bare   both removed: barriers + fragment reads + MFMAs = the floor of the two-group ping-pong schedule
    bare1  bare without barrier Y (one barrier per step)
    free   bare without any barrier in the loop: MFMA issue + LDS fragment reads of eight free-running waves
-> ab/libivid_abl_<name>.so, read by scripts/dev/kstep_ablation.py through IVID_HIP_LIB."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "ivid_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
DMA = "if (grp == 1 && (MORE || tap < 8)) issue_b_g1("
MORE = "constexpr bool MORE = decltype(more_c)::value;"
BAR_X = "        __syncthreads();  // barrier X\n"
BAR_Y = "        __syncthreads();  // barrier Y\n"


def patch(text, name):
    def rep(a, b, count=None):
        nonlocal text
        assert a in text, a
        text = text.replace(a, b) if count is None else text.replace(a, b, count)
    if name in ("nodma", "bare", "bare1", "free"):
        rep(DMA, "if (false) issue_b_g1(")
    if name in ("nohalo", "bare", "bare1", "free"):
        rep(MORE, "constexpr bool MORE = false;")
        if name == "nohalo":   # keep the weight stream of the next chunk's first tap
            rep("if (grp == 1 && (MORE || tap < 8)) issue_b_g1(", "if (grp == 1 && (decltype(more_c)::value || tap < 8)) issue_b_g1(")
    if name in ("bare1", "free"):
        rep(BAR_Y, "")
    if name == "free":
        rep(BAR_X, "")
    return text


def build(name):
    d = os.path.join(ROOT, "ab", "abl_" + name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, "ivid_amd", "csrc"))
    os.makedirs(os.path.join(d, "include"))
    shutil.copy(os.path.join(ROOT, "include", "ivid_hip.h"), os.path.join(d, "include"))
    for f in os.listdir(SRC):
        if f.endswith((".h", ".hip")):
            t = open(os.path.join(SRC, f)).read()
            if f == "conv3x3_fused_body.h":
                t = patch(t, name)
            open(os.path.join(d, "ivid_amd", "csrc", f), "w").write(t)
    objs = []
    for unit in ("conv3x3_fused", "conv3x3_fused128"):
        o = os.path.join(d, unit + ".o")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DIVID_DEV_TIMELINE",
                               "-c", os.path.join(d, "ivid_amd", "csrc", unit + ".hip"), "-o", o])
        objs.append(o)
    rest = [os.path.join(SRC, "build", f) for f in os.listdir(os.path.join(SRC, "build"))
            if f.endswith(".o") and not f.startswith("conv3x3_fused")]
    out = os.path.join(ROOT, "ab", f"libivid_abl_{name}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + rest)
    shutil.rmtree(d)
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or ["base", "nodma", "nohalo", "bare", "bare1", "free"]
    with ThreadPoolExecutor(max_workers=3) as ex:
        for p in ex.map(build, names):
            print(p, flush=True)
