import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ivid_amd import _lib
lib = _lib.load()
n, h = int(os.environ.get("N", "128")), 128
stream = torch.cuda.Stream(); sp = C.c_void_p(stream.cuda_stream)
tdt = torch.float16
g = torch.Generator(device="cuda"); g.manual_seed(1)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
c0, c1, cout = 256, 256, 256
cin = c0 + c1
x0 = rn(n, h, h, c0).to(tdt); x1 = rn(n, h, h, c1).to(tdt)
w = (rn(cout, 9 * cin) / (9 * cin) ** 0.5).to(tdt)
b = rn(cout)
ab = torch.rand(n, cin, 2, device="cuda", generator=g) + 0.5
print("inputs finite", torch.isfinite(x0.float()).all().item(), torch.isfinite(x1.float()).all().item(), torch.isfinite(w.float()).all().item())
out = torch.full((n, h, h, cout), float("nan"), device="cuda", dtype=tdt)
stats_buf = torch.zeros(n * (h // 4) * (h // 32) * cout * 2, device="cuda")
_lib.check(lib.ivid_conv3x3_gn_skip_c(_lib.F16, x0.data_ptr(), None, c0, x1.data_ptr(), None, c1, ab.data_ptr(), 0, w.data_ptr(), b.data_ptr(), out.data_ptr(), None,
                                      None, None, 0, n, h, h, cout, stats_buf.data_ptr(), None, 0, None, 0, None, sp), "launch")
torch.cuda.synchronize()
fin = torch.isfinite(out.float())
print("out finite frac", fin.float().mean().item())
bad = (~fin).nonzero()
print("n bad", bad.shape[0], bad[:5].tolist(), bad[-5:].tolist() if bad.shape[0] else None)
if bad.shape[0]:
    print("bad images", torch.unique(bad[:, 0]).tolist()[:20], "bad chans", torch.unique(bad[:, 3]).shape[0])
    i = bad[0]
    print("value", out[i[0], i[1], i[2], i[3]].item())
# reference for image 0
import torch.nn.functional as F
x = torch.cat([x0[:1], x1[:1]], -1).float()
act = F.silu(x * ab[:1, None, None, :, 0] + ab[:1, None, None, :, 1]).half().float()
ref = F.conv2d(act.permute(0, 3, 1, 2), w.float().reshape(cout, 9, cin).permute(0, 2, 1).reshape(cout, cin, 3, 3), b, padding=1).permute(0, 2, 3, 1)
print("ref finite", torch.isfinite(ref).all().item(), "absmax ref", ref.abs().max().item(), "rel err img0", ((out[:1].float() - ref).norm() / ref.norm()).item())
