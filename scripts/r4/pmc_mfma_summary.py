#!/usr/bin/env python
"""Summarise the SQ pass of scripts/r4/gpu_pmc_mfma.sh: per kernel (all its dispatches in the bench forward + warm-up), the sums
of the SQ counters, the kernel-trace durations, and the derived MFMA figures:
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GPU-active cycles of the kernel's dispatches), GPU-active cycles =
                    GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3, hence / 8): the share of all MFMA-pipe cycles that were busy
  mfma_issue_frac = same numerator over SQ_BUSY_CYCLES scaled to SIMD count (printed raw as well: the denominators differ in
                    what they count; both are given so that the reader can see the spread)
  eff_clock_ghz   = GRBM_GUI_ACTIVE / 8 / duration
Counters come from ONE rocprofv3 --pmc pass (8 SQ slots + GRBM)."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root, prec = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(repo, "ivid_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def short(k):
    for tag in ("conv3x3_fused128", "conv3x3_fused", "conv3x3_out", "conv_igemm", "attn_kernel", "gn_apply", "gn_finalize", "f32_to_hilo",
                "stem_im2col", "copy16"):
        if tag in k:
            ins = ""
            if "conv3x3_fused_kernel" in k or "conv_igemm" in k:
                ins = "<bf16x3>" if "bf16x3" in k else ("<f16>" if "DF16_" in k else ("<bf16>" if "DF16b" in k else "<f32>"))
            return tag + ins
    return None


f = glob.glob(f"{root}/SQ/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = short(r["Kernel_Name"])
    if k is None:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
dur = defaultdict(float)
kt = glob.glob(f"{root}/SQ/**/*kernel_trace.csv", recursive=True)
if kt:
    for r in csv.DictReader(open(kt[0])):
        k = short(r["Kernel_Name"])
        if k is not None:
            dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
out = {"precision": prec, "commit": os.environ.get("IVID_COMMIT"), "csrc_sha": csrc_sha(),
       "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY "
                  "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -- python bench.py "
                  "--precision <mode> --steps 1 --warmup 1 (IVID_NO_GRAPH=1)", "kernels": {}}
for k, c in sorted(acc.items(), key=lambda kv: -dur.get(kv[0], 0)):
    e = {"dispatches": len(disp[k]), "seconds": round(dur.get(k, 0.0), 6)}
    e.update({n: v for n, v in c.items()})
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui > 0:
        e["mfma_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui), 4)
        if dur.get(k):
            e["eff_clock_ghz"] = round(gui / dur[k] / 1e9, 3)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:   # quad-cycles (MI355X_MICROARCH.md): wave-level shares
        e["wave_share_wait_inst_any"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
        e["wave_share_wait_any"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4)
        e["wave_share_active_inst"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
