#!/usr/bin/env python
"""Compiler's resource report of every kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; no GPU needed):
registers, scratch (spills), occupancy.  -> a table on stdout (profiles/r04_kernel_resources.txt).
    python scripts/kernel_resources.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ivid_amd", "csrc")
units = [f[:-4] for f in sorted(os.listdir(SRC)) if f.endswith(".hip")]
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for u in units:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(SRC, u + ".hip"), "-o", os.path.join(tmp, u + ".o")],
                           capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
            if not m:
                continue
            k, v = m.groups()
            if k == "Function Name":
                # GNU c++filt predates the _Float16 / __bf16 manglings (DF16_, DF16b): hand it the older spellings of the same types
                name = subprocess.run(["c++filt", v.replace("DF16_", "Dh").replace("DF16b", "u6__bf16")], capture_output=True, text=True).stdout.strip()
                name = name.replace("(anonymous namespace)::", "").replace("(FusedArgs)", "").replace("half", "f16").replace("__bf16", "bf16")
                cur = dict(unit=u, name=re.sub(r"\((anonymous namespace)?.*\)$", "", name))
                rows.append(cur)
            else:
                cur[k] = v
print("# hipcc -Rpass-analysis=kernel-resource-usage, --offload-arch=gfx950 -O3 (scripts/kernel_resources.py).  Template arguments of")
print("# conv3x3_fused_kernel: <T, WIDE, LO, LOIN, SKS, O16>.  scratch = bytes per lane (VGPR spills), occ = waves per SIMD.")
print(f"{'unit':18s} {'VGPR':>4s} {'AGPR':>4s} {'SGPR':>4s} {'scratch':>7s} {'vspill':>6s} {'sspill':>6s} {'occ':>3s}  kernel")
for r in rows:
    print(f"{r['unit']:18s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('TotalSGPRs','?'):>4s} {r.get('ScratchSize [bytes/lane]','?'):>7s} "
          f"{r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s} {r.get('Occupancy [waves/SIMD]','?'):>3s}  {r['name'][:150]}")

# where the fused kernel's spills sit: scratch operations of every instantiation relative to its MFMA stream
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "fused.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-S", "--cuda-device-only",
                    "-c", os.path.join(SRC, "conv3x3_fused.hip"), "-o", asm], capture_output=True, text=True)
    lines = open(asm).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*conv3x3_fused_kernel.*:", l)]
ends = [i for i, l in enumerate(lines) if "s_endpgm" in l]
print()
print("# conv3x3_fused.hip, scratch operations per instantiation: position in the instruction stream = MFMAs issued before it / all MFMAs")
print("# (the main loop is straight-line code of 2 x 288 MFMAs per chunk pair in the f16 / bf16 forms; a reload after the last of them sits behind the loop)")
for a in starts:
    b = min(e for e in ends if e > a)
    seg = lines[a:b]
    mangled = seg[0].split(":")[0]
    name = subprocess.run(["c++filt", mangled.replace("DF16_", "Dh").replace("DF16b", "u6__bf16")], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("((anonymous namespace)::FusedArgs)", "").replace("half", "f16").replace("__bf16", "bf16")
    mf = [i for i, l in enumerate(seg) if "v_mfma" in l]
    ops = [(i, "store" if "scratch_store" in l else "load") for i, l in enumerate(seg) if re.search(r"\bscratch_(load|store)", l)]
    where = ", ".join(f"{kind} after {sum(1 for m in mf if m < i)}/{len(mf)}" for i, kind in ops) or "none"
    print(f"{name[:95]:95s} {where}")
