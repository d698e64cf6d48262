#!/bin/bash
# Round 4: first run of the fp16s mode: mini forwards vs golden, forward set large/small, quick bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tail -20
import sys; sys.path.insert(0,'tests')
import torch, common as C
from ivid_amd.diffusion.backbones import AdmUnet2d
for name,args,seed in (("mini_fwd",C.MINI,0),("mini_cond_fwd",C.MINI_COND,2),("mini_unclass_fwd",C.MINI_UNCLASS,1)):
    g=C.load_golden(name)
    x=C.seeded_randn(100+seed,2,args["in_channels"],32,32).cuda(); t=torch.full((2,),int(g["t"])).cuda()
    cls=torch.from_numpy(g["classes"]).cuda() if "classes" in g else None
    for p in ("fp16cx","fp16s"):
        m=AdmUnet2d(**args,precision=p); m.load_state_dict(C.synth_weights(args,seed)); m=m.cuda().eval()
        out=m(x,t,cls).cpu(); out2=m(x,t,cls).cpu(); out3=m(x,t,cls).cpu()
        print(name,p,"rel_l2 %.3e"%C.rel_l2(out,g["eps"]), "graph==eager", bool(torch.equal(out,out3)))
PY
python scripts/r4/fwd_set_modes.py large fp16s fp16cx 2>&1 | grep -v amdgpu.ids
python scripts/r4/fwd_set_modes.py small fp16s fp16cx 2>&1 | grep -v amdgpu.ids
for p in fp16s fp16cx; do
  IVID_BENCH_LAYERS=gpurun_out/layers_$p.json timeout 600 python bench.py --precision $p --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > gpurun_out/bench_$p.json 2> gpurun_out/bench_$p.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$p.json").read().strip().splitlines()[-1])
    print("$p", d["value"], d["ms_per_step"], d.get("kernel_time_ms_per_forward"))
except Exception as e:
    print("$p failed", e); print(open("gpurun_out/bench_$p.err").read()[-2500:])
PY
done
