import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, common as C
from ivid_amd.diffusion.backbones import AdmUnet2d
args = C.LARGE128 if os.environ.get("MODEL", "large") == "large" else C.SMALL128
m = AdmUnet2d(**args, precision=os.environ.get("PREC", "bf16")); m.load_state_dict(C.synth_weights(args, 4)); m = m.cuda()
B = int(os.environ.get("B", "8"))
x = C.seeded_randn(7, B, 4, 128, 128).cuda(); t = torch.full((B,), 500).cuda()
cls = (torch.arange(B) % 1000).cuda() if args["num_classes"] else None
def f(xx, tt, cc):
    return m(xx, tt, cc).clone()
a1 = f(x[:1], t[:1], cls[:1] if cls is not None else None)
a2 = f(x[:1], t[:1], cls[:1] if cls is not None else None)
print("repeat bs1 max diff", float((a1 - a2).abs().max()))
ab = f(x, t, cls)
ab2 = f(x, t, cls)
print(f"repeat bs{B} max diff", float((ab - ab2).abs().max()))
print(f"bs{B} row0 vs bs1 rel_l2", C.rel_l2(ab[:1].cpu(), a1.cpu()))
a3 = f(x[1:2], t[:1], cls[1:2] if cls is not None else None)
print(f"bs{B} row1 vs bs1 rel_l2", C.rel_l2(ab[1:2].cpu(), a3.cpu()))
