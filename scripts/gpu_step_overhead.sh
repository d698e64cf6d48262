cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import os, sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, common as C
from ivid_amd.diffusion import frameworks, samplers
from ivid_amd.diffusion.backbones import AdmUnet2d
bs = 32
cargs = dict(C.LARGE128, in_channels=10)
mc = AdmUnet2d(**cargs, precision="bf16"); mc.load_state_dict(C.synth_weights(cargs, 2)); mc = mc.cuda()
fc = frameworks.InpaintCFG(mc, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0)
smp = samplers.DdimSampler(fc)
y = torch.randn(bs, 4, 128, 128, device="cuda"); m = (torch.rand(bs, 1, 128, 128, device="cuda") > 0.3).float()
cls = torch.arange(bs, device="cuda") % 1000
args = dict(y=y, mask=m, mask_rgb=m, replace_rgb=(0.1, y[:, :3], m), replace_depth=(0.2, y[:, 3:], m), constrain_depth=(0.5, y[:, 3:]))
def run(nf, steps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    smp.sample(bs, classes=cls, steps=steps, strength=3.0, verbose=False, keep_intermediates=False, noise_fn=nf, **args)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
gens = [torch.Generator(device="cuda").manual_seed(i) for i in range(bs)]
per = lambda shape: torch.cat([torch.randn((1,) + tuple(shape[1:]), device="cuda", generator=g) for g in gens], 0)
g1 = torch.Generator(device="cuda").manual_seed(1)
glob = lambda shape: torch.randn(tuple(shape), device="cuda", generator=g1)
run(glob, 3)
res = {"ms_per_step_global_noise": run(glob), "ms_per_step_per_sample_generators": run(per)}
plan = mc.plan(bs, stacked=True)
x = torch.randn(2 * bs, 10, 128, 128, device="cuda")
t = torch.full((2*bs,), 500, device="cuda", dtype=torch.long)
import ivid_amd
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    fc.eps_branches(torch.randn(bs, 4, 128, 128, device="cuda"), torch.full((bs,), 500, device="cuda", dtype=torch.long), y, m, classes=cls, strength=3.0, mask_rgb=m, noise_fn=glob)
torch.cuda.synchronize(); res["ms_per_eps_branches_call"] = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps(res))
PY
