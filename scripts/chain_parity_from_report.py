#!/usr/bin/env python
"""gpurun_out/parity_report.json (written by the -m gpu tests) -> profiles/r03_chain_parity.json: the sampler-chain deviations of
every precision mode from the live reference's samples, the figures bench.py's headline selection reads."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rep = json.load(open(os.path.join(root, "gpurun_out", "parity_report.json")))
chains = {k: v for k, v in rep.items() if k.startswith("chain/config2_bs2_") or k.startswith("chain/config1_bs2_")}
fwd = {k: v for k, v in rep.items() if k.startswith("unet/large128_fwd/") or k.startswith("unet/small128_fwd/")}
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
out = {"from": "tests/test_unet_gpu.py on an MI355X (gpurun), tree at or after commit %s%s" % (head, " " + sys.argv[1] if len(sys.argv) > 1 else ""),
       "what": {"config2_bs2": "BASELINE config 2 as a chain: large cfg model, ClassifierFreeGuidance 0.5 + DdimSampler 50 steps eta 0, bs 2, "
                               "samples vs tests/golden/large128_ddim50_cfg.npz (live reference, tests/golden/make_golden_c2.py)",
                "config1_bs2": "BASELINE config 1 at bs 2: small-128 model + DdimSampler 10 steps, samples vs tests/golden/small128_ddim10.npz"},
       "chains": chains, "forwards": fwd}
json.dump(out, open(os.path.join(root, "profiles", "r03_chain_parity.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v["samples"] for k, v in chains.items()}, indent=1))
