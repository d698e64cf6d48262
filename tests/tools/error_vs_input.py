"""How the deviation of the 16-bit modes depends on the INPUT (dev tool, CPU, TEST INFRASTRUCTURE): emulated fp16c and fp16
roundings inside the fp32 oracle forward (tests/tools/error_budget.py) on the large-128 synthetic checkpoint, for pure-noise inputs
at t = 999 and for structured inputs (half noise, half a smooth field) at other timesteps.

    python tests/tools/error_vs_input.py > profiles/r03_error_vs_input.txt
"""
import sys, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..')); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, '..', '..'))
import common as C, error_budget as E
torch.set_num_threads(os.cpu_count())
args=C.LARGE128
sd={k:v.float() for k,v in C.synth_weights(args,4).items()}
def q(on):
    qq=E.Q(on, torch.float16); qq.only={"stem","head"}; qq.invert=True; return qq
for t in (0, 20, 250, 500, 750, 999):
    for seed,cls in ((104,[7]), (7,[416]), (9,None)):
        x=C.seeded_randn(seed,1,4,128,128)
        if t<999:  # x_t for smaller t looks like signal + noise; use a smoother input: mix of noise and a low-frequency field
            x = 0.5*x + 0.5*torch.nn.functional.interpolate(C.seeded_randn(seed+1,1,4,8,8), size=128, mode='bilinear')
        tt=torch.full((1,),t,dtype=torch.long); cl=torch.tensor(cls) if cls is not None else None
        ref=E.forward(sd,args,x,tt,cl,E.Q("",torch.float16))
        a=E.forward(sd,args,x,tt,cl,q("WAHQF")); ax=E.forward(sd,args,x,tt,cl,q("WAQ")); b=E.forward(sd,args,x,tt,cl,E.Q("WATHQ",torch.float16))
        print(f"t={t:4d} seed={seed} cls={cls} input={'noise' if t == 999 else 'noise/2 + smooth/2'}: fp16c {C.rel_l2(a,ref):.3e}   fp16cx {C.rel_l2(ax,ref):.3e}   fp16 {C.rel_l2(b,ref):.3e}", flush=True)
