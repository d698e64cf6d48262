#!/usr/bin/env python
"""Round-6 kernel A/B helper: ab/libivid_x<mask>[_tl].so = the current library with the fused 3x3 units (and optionally others)
rebuilt with -DIVID_EXP=<mask> (experimental code paths in csrc/, see the `#if IVID_EXP & n` blocks) and, with `tl`, the phase
time stamps of scripts/dev/build_timeline.sh.  ab/ is git-ignored but travels to the GPU box.
    python scripts/r6/build_exp.py 0 1 3 [--tl] [--units conv3x3_fused,conv3x3_fused128]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "ivid_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tl = "--tl" in sys.argv
units = ["conv3x3_fused", "conv3x3_fused128"]
for a in sys.argv[1:]:
    if a.startswith("--units="):
        units = a.split("=", 1)[1].split(",")


def build(mask):
    tag = f"x{mask}" + ("_tl" if tl else "")
    d = os.path.join(ROOT, "ab", "obj_" + tag)
    os.makedirs(d, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", f"-DIVID_EXP={mask}"] + (["-DIVID_DEV_TIMELINE"] if tl else [])
    srcs = sorted(f[:-4] for f in os.listdir(SRC) if f.endswith(".hip")) if tl else units
    objs = []

    def cc(u):
        o = os.path.join(d, u + ".o")
        subprocess.check_call([HIPCC] + flags + ["-c", os.path.join(SRC, u + ".hip"), "-o", o])
        return o
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, srcs))
    rest = [] if tl else [os.path.join(SRC, "build", f) for f in sorted(os.listdir(os.path.join(SRC, "build")))
                          if f.endswith(".o") and f[:-2] not in units]
    out = os.path.join(ROOT, "ab", f"libivid_{tag}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + rest)
    return out


for m in args:
    print(build(int(m)), flush=True)
