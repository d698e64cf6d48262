import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, common as C
from ivid_amd.diffusion.backbones import AdmUnet2d
from ivid_amd.diffusion.backbones.plan import UNetPlan
m = AdmUnet2d(**C.MINI, precision=os.environ.get("PREC", "fp32")); m.load_state_dict(C.synth_weights(C.MINI, 0)); m = m.cuda()
x = C.seeded_randn(100, 2, 4, 32, 32).cuda(); t = torch.full((2,), 37).cuda(); cls = torch.tensor([3, -1]).cuda()
mode = os.environ.get("MODE", "eager")
if mode == "debug":
    plan = UNetPlan(m.spec, m._weights(), m.device, 2, False, m.tile_cfg, debug=True)
    plan.run(x, t, cls, use_graph=False)
    torch.cuda.synchronize()
    print("OK debug taps", len(plan.taps))
elif mode == "graph":
    for i in range(3):
        out = m(x, t, cls)
    torch.cuda.synchronize()
    print("OK graph", float(out.abs().mean()))
else:
    m.use_graph = False
    out = m(x, t, cls)
    torch.cuda.synchronize()
    print("OK eager", float(out.abs().mean()))
