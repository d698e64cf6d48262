"""The adaptive precision modes (ivid_amd/diffusion/backbones/adm.py note_timestep; "fp16sa" is what use_fp16 configs select): fp16s,
except that a forward whose caller announced a timestep >= adaptive_t (150 since round 5) runs fp16s WITHOUT its split-precision
island ("fp16cs"); "fp16sa3" also drops the split skip convolutions from t >= 500 (plain fp16cx); "fp16sx" is the strict ladder that
holds both parity metrics.  The island buys its tolerance on nearly clean inputs only; every row of every forward set -- the four
BASELINE backbones, their mid-t sets (t = 50 .. 350), and six further synthetic checkpoints of the two 128^2 backbones (more draws +
a "trained-like" variant) -- is checked here in the mode its timestep selects, against the live reference's outputs."""
import pytest
import torch

import common as C
import gpu_util as G

pytestmark = pytest.mark.gpu
BAR = 9.5e-4


def build(args, seed, precision):
    from ivid_amd.diffusion.backbones import AdmUnet2d
    m = AdmUnet2d(**args, precision=precision)
    m.load_state_dict(C.synth_weights(args, seed), strict=True)
    return m.cuda().eval()


def test_the_announced_timestep_selects_the_plan_and_is_consumed_by_one_call():
    args = C.MINI
    ma, ms, mc = build(args, 5, "fp16sa"), build(args, 5, "fp16s"), build(args, 5, "fp16cs")
    T = ma.adaptive_t
    x = C.seeded_randn(1, 3, 4, 32, 32).cuda()
    cls = torch.tensor([1, 4, 7]).cuda()
    for t in (T - 1, T, 999):
        tt = torch.full((3,), t, dtype=torch.long).cuda()
        want_s, want_c = ms(x, tt, cls), mc(x, tt, cls)
        assert not torch.equal(want_s, want_c)
        assert torch.equal(ma(x, tt, cls), want_s)                       # nobody announced anything: the base mode
        ma.note_timestep(t)
        assert torch.equal(ma(x, tt, cls), want_c if t >= T else want_s)
        assert torch.equal(ma(x, tt, cls), want_s)                       # the announcement was for ONE call
        ma.note_timestep(t)
        ec, eu = ma.forward_cfg(x, tt, cls)
        rc, ru = (mc if t >= T else ms).forward_cfg(x, tt, cls)
        assert torch.equal(ec, rc) and torch.equal(eu, ru)
    # a call that fails still consumes its announcement
    ma.note_timestep(999)
    with pytest.raises(IndexError):
        ma(x, torch.full((3,), 999).cuda(), torch.tensor([1, 4, 10 ** 6]).cuda())
    tt = torch.full((3,), 999, dtype=torch.long).cuda()
    assert torch.equal(ma(x, tt, cls), ms(x, tt, cls))
    # the other modes ignore announcements
    ms.note_timestep(999)
    assert torch.equal(ms(x, tt, cls), ma(x, tt, cls))


def test_a_sampler_chain_in_the_adaptive_mode_is_the_two_modes_spliced_at_the_threshold():
    from ivid_amd.diffusion import frameworks, samplers
    args = C.MINI
    ms_ = {p: build(args, 5, p) for p in ("fp16sa", "fp16s", "fp16cs")}
    fw = {p: frameworks.ClassifierFreeGuidance(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1) for p, m in ms_.items()}
    sm = {p: samplers.DdimSampler(f) for p, f in fw.items()}
    T = ms_["fp16sa"].adaptive_t
    cls = torch.tensor([2, 9]).cuda()
    x_a = x_b = C.seeded_randn(3, 2, 4, 32, 32).cuda() * 0.05           # small: the synthetic-weight chain must stay finite
    pairs = [(t, t - 100) for t in range(1000, 0, -100)]       # t - 1 = 999 .. 99: the last step runs below the threshold
    used = set()
    for t, tp in pairs:
        x_a = sm["fp16sa"].sample_once(x_a, t, tp, cls, strength=0.5).pred_x_prev
        p = "fp16cs" if t - 1 >= T else "fp16s"
        used.add(p)
        x_b = sm[p].sample_once(x_b, t, tp, cls, strength=0.5).pred_x_prev
        assert torch.equal(x_a, x_b), t
    assert used == {"fp16cs", "fp16s"} and bool(torch.isfinite(x_a).all())
    # DDPM announces its timestep too
    fwp = {p: frameworks.GaussianDiffusion(ms_[p], timesteps=1000, beta_schedule="linear") for p in ms_}
    dp = {p: samplers.DdpmSampler(f) for p, f in fwp.items()}
    noise = C.seeded_randn(4, 2, 4, 32, 32).cuda()
    xa = dp["fp16sa"].sample_once(x_a, 600, noise_fn=lambda s: noise).pred_x_prev
    xc = dp["fp16cs"].sample_once(x_a, 600, noise_fn=lambda s: noise).pred_x_prev
    assert torch.equal(xa, xc)


SEED_TAGS = sorted(C.FWD_SETS_SEEDS)   # round 5: further synthetic checkpoints (more draws + the "trained-like" variant)


@pytest.mark.parametrize("tag", ["large128", "small128", "largecond128", "sr256", "large128_mid", "small128_mid", "largecond128_mid", "sr256_mid",
                                 "largecond128_mid2", "sr256_mid2"] + SEED_TAGS)
def test_every_forward_set_row_in_the_mode_its_timestep_selects(tag):
    args, seed, gname, make, _crop = C.FWD_SETS[tag]
    g = C.load_golden(gname)
    for key, x, _, _ in make():   # the seeded recipe rebuilds the generator's inputs
        assert abs(float(x.double().sum()) - float(g[key + "_xsum"])) < 1e-3 * max(1.0, abs(float(g[key + "_xsum"]))), key
    m = build(args, seed, "fp16sa")
    out = {}
    for prec in ("fp16sa", "fp16s", "fp16cs"):
        m.set_precision(prec)
        rows = C.fwd_set_deviation(m, tag)
        worst = max(rows, key=rows.get)
        out[prec] = rows
        G.report(f"fwd_set/{tag}/{prec}", max=rows[worst], argmax=worst, min=min(rows.values()), **rows)
    T = m.adaptive_t
    tof = lambda key: int(key.split("_t")[1].split("_")[0])
    print(f"forward set {tag}:", {p: "%.3e" % max(r.values()) for p, r in out.items()},
          "| fp16cs at t >= %d: %.3e" % (T, max([v for k, v in out["fp16cs"].items() if tof(k) >= T], default=0.0)))
    for key, v in out["fp16sa"].items():                                   # the adaptive mode IS the splice, row by row
        assert v == out["fp16cs" if tof(key) >= T else "fp16s"][key], key
    assert max(out["fp16sa"].values()) < BAR, (tag, out["fp16sa"])
    assert max(out["fp16s"].values()) < BAR, (tag, out["fp16s"])           # the headline mode on the mid-t rows as well
    # the STRICT ladder "fp16sx" (bf16x3 / fp16s / fp16cs at t < 250 / < 500 / >= 500): BOTH metrics of SURVEY.md 8(c) -- rel-L2 and
    # max-abs error / max-abs reference -- under 1e-3 on every row
    m.set_precision("fp16sx")
    rows, mrel = C.fwd_set_deviation(m, tag, with_max_rel=True)
    G.report(f"fwd_set/{tag}/fp16sx", max=max(rows.values()), max_rel_max=max(mrel.values()), max_rel_argmax=max(mrel, key=mrel.get))
    assert max(rows.values()) < BAR and max(mrel.values()) < 1e-3, (tag, max(rows.values()), max(mrel, key=mrel.get), max(mrel.values()))
    # the three-tier ladder "fp16sa3" (+ plain fp16cx from t >= 500) on the UNCONDITIONAL 128^2 backbones, where it is inside the bar
    if tag.startswith(("large128", "small128")):
        m.set_precision("fp16sa3")
        rows3 = C.fwd_set_deviation(m, tag)
        G.report(f"fwd_set/{tag}/fp16sa3", max=max(rows3.values()), argmax=max(rows3, key=rows3.get))
        assert max(rows3.values()) < BAR, (tag, max(rows3, key=rows3.get), max(rows3.values()))
        assert all(v == out["fp16sa"][k] for k, v in rows3.items() if tof(k) < 500)


def test_a_full_size_ladder_chain_is_reproducible_bit_for_bit():
    """The headline workload in miniature: BASELINE config 2's sampler (DDIM, CFG 0.5) on the large cfg backbone in the three-tier
    ladder, 25 steps from t = 1000 (so every tier's plan is built, captured into its own hipGraph and switched to mid-chain), bs 4.
    Two runs from the same x_T must agree bit for bit -- tier switches between graphs on one stream leave no race or stale buffer --
    and the chain must equal the one a fresh model produces."""
    from ivid_amd.diffusion import frameworks, samplers
    outs = []
    for fresh in range(2):
        m = build(C.LARGE128, 4, "fp16sa3")
        fw = frameworks.ClassifierFreeGuidance(m, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
        smp = samplers.DdimSampler(fw)
        x_T = C.seeded_randn(5, 4, 4, 128, 128).cuda()
        cls = torch.tensor([1, 22, 333, 999]).cuda()
        for rep in range(2 if fresh == 0 else 1):
            outs.append(smp.sample(4, noise=x_T, classes=cls, steps=25, strength=0.5, verbose=False).samples.clone())
        assert sorted(k[2] if len(k) > 2 else 0 for k in m._plans) == [0, 1, 2]      # all three tiers ran
        del m, fw, smp
        torch.cuda.empty_cache()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
