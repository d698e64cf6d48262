"""Full-size engine check: BASELINE config 2's stacked forward (large-128, bs 8 + 8, fp16s) frozen into an engine file, run by
the C host program, compared bit for bit with the model's own forward; prints file size, load + forward times.
    python scripts/r4/engine_fullsize.py [batch]   (GPU box)"""
import os, subprocess, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import common as C
from ivid_amd.diffusion.backbones import AdmUnet2d
from ivid_amd import build as B
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
args = C.LARGE128
m = AdmUnet2d(**args, precision="fp16s")
m.load_state_dict(C.synth_weights(args, 0), strict=True)
m = m.cuda().eval()
x = C.seeded_randn(5, bs, 4, 128, 128).cuda()
t = torch.full((bs,), 500, dtype=torch.long).cuda()
cls = (torch.arange(bs) * 37 % 1000).cuda()
want = torch.cat(m.forward_cfg(x, t, cls)).cpu().numpy()
tmp = "/tmp/eng"; os.makedirs(tmp, exist_ok=True)
t0 = time.time(); blob = m.export_engine(bs, True, path=f"{tmp}/large.eng"); t_exp = time.time() - t0
open(f"{tmp}/in.bin", "wb").write(x.cpu().numpy().tobytes() + t.cpu().numpy().tobytes() + cls.cpu().numpy().astype(np.int64).tobytes())
del m; torch.cuda.empty_cache()
t0 = time.time()
r = subprocess.run([B.HOST_BIN, f"{tmp}/large.eng", f"{tmp}/in.bin", f"{tmp}/out.bin", "12"], capture_output=True, text=True, timeout=600)
t_host = time.time() - t0
got = np.fromfile(f"{tmp}/out.bin", dtype=np.float32).reshape(want.shape)
out = dict(model="large128", precision="fp16s", batch=bs, stacked=True, engine_bytes=len(blob), export_seconds=round(t_exp, 2),
           host_returncode=r.returncode, host_stdout=r.stdout.strip(), host_stderr=r.stderr.strip()[-300:],
           host_wall_seconds=round(t_host, 2), bit_identical_to_model_forward=bool(np.array_equal(got, want)))
print(json.dumps(out, indent=1))
