#!/usr/bin/env python
"""gpurun_out/parity_report.json (written by the -m gpu tests) -> profiles/r04_chain_parity.json: the sampler-chain deviations of
every precision mode from the live reference's samples (bench.py measures the config-2 / config-1 chains of its headline rule
itself, in the run; this file is the record of ALL chain goldens incl. the CFG-3.0 settings of configs 3 / 4 / 5), plus the
forward-set and teacher-forced rows."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rep = json.load(open(os.path.join(root, "gpurun_out", "parity_report.json")))
chains = {k: v for k, v in rep.items() if k.startswith(("chain/config2_bs2_", "chain/config1_bs2_", "chain16/", "chain/sample_all_scene"))}
fwd = {k: v for k, v in rep.items() if k.startswith(("unet/large128_fwd/", "unet/small128_fwd/", "teacher_forced/"))}
fwd.update({k: {a: b for a, b in v.items() if a in ("max", "argmax", "min")} for k, v in rep.items() if k.startswith("fwd_set/")})
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
out = {"from": "tests/test_unet_gpu.py on an MI355X (gpurun), tree at or after commit %s%s" % (head, " " + sys.argv[1] if len(sys.argv) > 1 else ""),
       "what": {"config2_bs2": "BASELINE config 2 as a chain: large cfg model, ClassifierFreeGuidance 0.5 + DdimSampler 50 steps eta 0, bs 2, "
                               "samples vs tests/golden/large128_ddim50_cfg.npz (live reference, tests/golden/make_golden_c2.py)",
                "config1_bs2": "BASELINE config 1 at bs 2: small-128 model + DdimSampler 10 steps, samples vs tests/golden/small128_ddim10.npz",
                "chain16/ddpm250_cfg3_smallcfg": "DdpmSampler 250 ancestral steps + ClassifierFreeGuidance 3.0 (the first view of configs 3 / 4, "
                                                 "inference/sample.py:44-47,79), class-conditional small-128 backbone, bs 1, vs smallcfg_ddpm250_cfg3.npz",
                "chain16/inpaint50_cfg3_mini128cond": "InpaintCFG 3.0 + DdimSampler 50 steps with replace_rgb / replace_depth / constrain_depth "
                                                      "(sample.py:99-122) on the scene fixture's conditioning, vs mini128cond_inpaint50.npz",
                "chain16/superres4_cfg3_minisr": "SuperResCFG 3.0 + DDIM 4 steps (config 5's setting), vs mini_superres.npz; framework_eps_cfg3 = ONE guided "
                                                 "eps at t = 500 (guidance 3.0 multiplies a forward's deviation by up to 1 + 2s = 7)",
                "chain/sample_all_scene": "the reference's own sample_all on the scene fixture (3 views, conditioning in the loop), fp32 and fp16s",
                "fwd_set": "max over the representative forward set (tests/golden/*_fwd_set.npz)",
                "teacher_forced": "guided eps on the config-2 chain's own inputs at steps 1, 10, 25, 49"},
       "chains": chains, "forwards": fwd}
json.dump(out, open(os.path.join(root, "profiles", "r04_chain_parity.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v.get("samples", v.get("view2")) for k, v in chains.items()}, indent=1))
