"""The REFERENCE's own AdmUnet2d (imported from /root/reference) timed on the host cores of the BUILD CONTAINER with bench.py's
cpu_baseline harness (SURVEY.md 8(d) "CPU reference timing"; the GPU box has no /root/reference, so its bench line can only time
the oracle port in-run).  The oracle port is timed right after it on the same cores: their ratio is what lets a reader translate
the GPU box's in-run `port` figure into a `reference` one.   python scripts/r5/cpu_reference_baseline.py  -> profiles/r05_cpu_reference.json"""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as C  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

unit = "UNet fwd/s (bs=64, 128x128 RGBD)"
ref = bench.cpu_baseline(C, dict(C.LARGE128), True, "large", 64, unit)
assert ref["kind"] == "reference", "run this in the build container (needs /root/reference)"
real = os.path.isdir
os.path.isdir = lambda p: False if p.startswith("/root/reference") else real(p)      # second leg: the oracle port on the same cores
try:
    port = bench.cpu_baseline(C, dict(C.LARGE128), True, "large", 64, unit)
finally:
    os.path.isdir = real
assert port["kind"] == "port"
port["kind_note"] = ("the oracle (oracle/adm_oracle.py, pinned to the reference bit-for-bit by tests/golden), forced here for the "
                     "comparison although /root/reference exists in this container")
for leg in (ref, port):
    leg.pop("reference_timing_committed", None)
cpu = ""
try:
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
out = {"what": "BASELINE config 2's model (rgbd_imagenet_adm_128_large_cfg, fp32) on the build container's host cores: the reference's "
               "own code vs the oracle port, same harness (bench.py cpu_baseline), same cores, back to back",
       "host": {"cpu_model": cpu, "logical_cpus": os.cpu_count(), "threads_used": ref["cores"]},
       "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
       "reference": ref, "port": port,
       "port_over_reference": round(port["sample_fwd_per_s"] / ref["sample_fwd_per_s"], 4)}
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_cpu_reference.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
