"""Round-5 debugging aid: why does a batch-3 forward after plan evictions differ from the batch-32 rows?"""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as C  # noqa: E402
from ivid_amd.diffusion.backbones import AdmUnet2d  # noqa: E402

warnings.simplefilter("ignore")
m = AdmUnet2d(**C.LARGE128, precision="fp16sa")
m.load_state_dict(C.synth_weights(C.LARGE128, 4), strict=True)
m = m.cuda().eval()
x = C.seeded_randn(11, 32, 4, 128, 128).cuda()
cls = (torch.arange(32) * 31 % 1000).cuda()
d = lambda a, b: "max|d| %.3e equal %s" % (float((a - b).abs().max()), bool(torch.equal(a, b)))
ref = {}
for t in (20, 999):
    tt = torch.full((32,), t, dtype=torch.long).cuda()
    m.note_timestep(t)
    ref[t] = [v.clone() for v in m.forward_cfg(x, tt, cls)]
for bs in (2, 3, 5):
    for t in (20, 999):
        tt = torch.full((bs,), t, dtype=torch.long).cuda()
        m.note_timestep(t)
        ec, eu = m.forward_cfg(x[:bs], tt, cls[:bs])
        print("no eviction: bs", bs, "t", t, "cond", d(ec, ref[t][0][:bs]), "| uncond", d(eu, ref[t][1][:bs]), "| plans", len(m._plans), flush=True)
big = max(p.arena.total_bytes() for p in m._plans.values())
m.max_plan_bytes = int(1.2 * big) + sum(w.nbytes() for w in m._packed_tiers.values())
for bs in (4, 3, 6):
    for t in (999, 20):
        tt = torch.full((bs,), t, dtype=torch.long).cuda()
        m.note_timestep(t)
        ec, eu = m.forward_cfg(x[:bs], tt, cls[:bs])
        print("tight budget: bs", bs, "t", t, "cond", d(ec, ref[t][0][:bs]), "| vs the OTHER t", d(ec, ref[1019 - t][0][:bs]), "| plans", sorted(m._plans), flush=True)
        m.note_timestep(t)
        ec2, _ = m.forward_cfg(x[:bs], tt, cls[:bs])
        print("   again (graph):", d(ec2, ref[t][0][:bs]), flush=True)
