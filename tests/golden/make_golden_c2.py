"""BASELINE config 2 as a golden chain, generated from the LIVE reference (/root/reference), build container only:
    python tests/golden/make_golden_c2.py
rgbd_imagenet_adm_128_large_cfg backbone (configs/rgbd_imagenet_adm_128_large_cfg.json, use_fp16 false = fp32) with the
deterministic synthetic checkpoint, ClassifierFreeGuidance (classifier_free_guidance.py:23-42) + DdimSampler.sample
(samplers/ddim.py:105-165): 50 steps, strength 0.5, eta 0, batch 2, recorded x_T.  ~200 CPU forwards of the large model.
Writes tests/golden/large128_ddim50_cfg.npz (samples, first / middle / last pred_x_0) and checks the oracle chain
against the reference's samples while it is at it (the oracle pin of this chain).
Round 4: also records, TEACHER-FORCED, what the reference's framework saw and answered at steps 1, 10, 25 and 49 of that chain
(model_inference's input x_t, timestep and guided eps, classifier_free_guidance.py:39-42) -> large128_ddim50_cfg_steps.npz, so
that a product forward can be compared on the chain's own inputs step by step (the samples alone cannot see steps 2-50: the
synthetic-weight chain is dominated by its first x0 estimate).
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
    __setattr__ = __setitem__


_m = types.ModuleType("easydict"); _m.EasyDict = EasyDict; sys.modules["easydict"] = _m
sys.path.insert(0, "/root/reference")
import diffusion.backbones as rb  # noqa: E402
import diffusion.frameworks as rf  # noqa: E402
import diffusion.samplers as rs  # noqa: E402

import common as C  # noqa: E402
from oracle import adm_oracle, sampler_oracle  # noqa: E402

torch.set_num_threads(os.cpu_count())
STEPS, STRENGTH, B = 50, 0.5, 2
REC_STEPS = (1, 10, 25, 49)   # 0-based index of the sample_once call


@torch.no_grad()
def main():
    args = C.LARGE128
    m = rb.AdmUnet2d(**args).eval()
    sd = C.synth_weights(args, 4)
    m.load_state_dict(sd, strict=True)
    fw = rf.ClassifierFreeGuidance(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
    smp = rs.DdimSampler(fw)
    x_T = C.seeded_randn(2024, B, 4, 128, 128)
    cls = torch.tensor([7, 416])
    torch.manual_seed(3)
    t0 = time.time()
    rec, calls, inner = {}, [0], fw.model_inference

    def spy(x_t, t, classes=None, **kw):
        out = inner(x_t, t, classes=classes, **kw)
        if calls[0] in REC_STEPS:
            k = calls[0]
            rec[f"x_step{k}"], rec[f"eps_step{k}"], rec[f"t_step{k}"] = x_t.numpy().copy(), out.numpy().copy(), np.int64(int(t[0]))
        calls[0] += 1
        return out
    fw.model_inference = spy
    ref = smp.sample(B, noise=x_T, classes=cls, steps=STEPS, strength=STRENGTH, verbose=False)
    dt = time.time() - t0
    fw.model_inference = inner
    print(f"reference chain: {dt:.1f} s", flush=True)
    old = os.path.join(HERE, "large128_ddim50_cfg.npz")
    if os.path.exists(old):   # the spy must not have changed the chain
        assert np.array_equal(np.load(old)["samples"], ref.samples.numpy()), "chain differs from the committed golden"
    rec["classes"] = cls.numpy()
    np.savez_compressed(os.path.join(HERE, "large128_ddim50_cfg_steps.npz"), **rec)
    arrays = dict(samples=ref.samples.numpy(), x0_first=ref.pred_x_0[0].numpy(), x0_mid=ref.pred_x_0[STEPS // 2].numpy(),
                  x0_last=ref.pred_x_0[-1].numpy(), classes=cls.numpy(), x_checksum=np.float64(x_T.double().sum()),
                  steps=np.int64(STEPS), strength=np.float64(STRENGTH))
    errs = dict(ref_seconds=round(dt, 1))
    if os.environ.get("C2_SKIP_ORACLE") != "1":
        torch.manual_seed(3)
        um = lambda a, b, c: adm_oracle.unet_forward(sd, args, a, b, c)
        eps = lambda x, t: sampler_oracle.cfg_eps(um, x, t, cls, STRENGTH)
        orc = sampler_oracle.ddim_sample(eps, x_T, STEPS, fw.betas)
        errs.update(samples=C.rel_l2(orc["samples"], ref.samples), x0_first=C.rel_l2(orc["pred_x_0"][0], ref.pred_x_0[0]))
    np.savez_compressed(os.path.join(HERE, "large128_ddim50_cfg.npz"), **arrays)
    mf = os.path.join(HERE, "manifest.json")
    man = json.load(open(mf))
    man["large128_ddim50_cfg"] = dict(
        note="BASELINE config 2 at bs 2: rgbd_imagenet_adm_128_large_cfg (fp32) + ClassifierFreeGuidance strength 0.5 + "
             "DdimSampler 50 steps eta 0, classes [7, 416], x_T = seeded_randn(2024)", oracle_vs_reference=errs)
    json.dump(man, open(mf, "w"), indent=1, sort_keys=True)
    print("large128_ddim50_cfg:", errs)


main()
