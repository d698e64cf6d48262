#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: UNet forwards per second at 128x128 RGBD, bs = 64.

Workload (BASELINE config 2, SURVEY.md §8d "C2"): rgbd_imagenet_adm_128_large_cfg backbone (421.5 M params,
613.78 GFLOP per sample-forward), ClassifierFreeGuidance strength 0.5, DDIM, batch 64, classes = arange(64) % 1000,
synthetic seeded weights (no checkpoints offline), synthetic x_T.  One bench "step" = one DDIM denoise step =
both guidance branches (2 UNet forwards at bs 64, executed as one stacked batch-128 forward replayed from a
hipGraph) + the fused CFG/DDIM update kernel.  `value` = UNet bs-64 forwards per second summed over all ranks.

Multi-GPU (`torchrun ... bench.py --gpus N`): sample-parallel — every rank runs its own batch of 64 (weak scaling),
weights are synthesised on rank 0 and broadcast once over RCCL; no collective inside the timed region.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (conv_igemm, the dominant kernel family: algorithmic
FLOPs of every launch in one forward / HIP-event time of those launches; plus the whole-forward figure),
`cpu_baseline` (the fp32 CPU oracle timed on the host cores of rank 0 on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md "Chip-level parameters"
GFLOP_PER_SAMPLE_FWD = {"large": 613.78, "small": 156.56}  # BASELINE.md §2 (2 x MACs of conv/linear + attention)


def conv_flops(args):
    """Algorithmic FLOPs of one ivid_conv2d launch from its recorded arguments (2 x MACs)."""
    (dtype, _s0, c0, _s1, c1, _w, _b, _o, _r, _rm, _om, n, h, w, cout, taps, _tc, _st) = args
    return 2.0 * n * h * w * cout * taps * (c0 + c1), dtype


def fused_flops(args):
    """Algorithmic FLOPs of one ivid_conv3x3_gn[_skip] launch (2 x MACs; the GroupNorm/SiLU prologue counts 0)."""
    (_dtype, _s0, c0, _s1, c1, _ab, _up, _w, _b, _o, _r, _rm, n, h, w, cout, _st) = args[:17]
    skc = (args[18] + args[20]) if len(args) > 17 else 0      # 1x1 skip_connection folded into the kernel
    return 2.0 * n * h * w * cout * (9 * (c0 + c1) + skc)


def conv_bytes(args, fused):
    """Algorithmic HBM bytes of one convolution launch: every input element, weight and residual read once, every output
    written once (compulsory traffic; the PMC-measured figure is reported beside it as `traffic`)."""
    skc = 0
    if fused:
        (dtype, _s0, c0, _s1, c1, _ab, up, _w, _b, _o, r, rm, n, h, w, cout, _st) = args[:17]
        skc = (args[18] + args[20]) if len(args) > 17 else 0
        taps, om = 9, 0
    else:
        (dtype, _s0, c0, _s1, c1, _w, _b, _o, r, rm, om, n, h, w, cout, taps, _tc, _st) = args
        up = 0
    esz = 2 if dtype == 1 else 4
    src = n * (h >> up) * (w >> up) * (c0 + c1) * esz
    out = n * h * w * cout * (4 if om else esz)
    res = 0 if not rm else (out if rm == 1 else (out // 4 if rm == 2 else out * 4))
    return float(src + out + res + cout * taps * (c0 + c1) * esz + n * h * w * skc * esz + cout * skc * esz)


def attn_flops(args):
    (_dtype, _q, _o, n, t, heads) = args
    return 2.0 * (2.0 * heads * t * t * 64) * n


def pmc_traffic():
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        return round(float(json.load(open(f))["conv_family_hbm_bytes_per_launch"]), 0)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--model", default="large", choices=["large", "small"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--guidance", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-breakdown", action="store_true")
    a = ap.parse_args()

    import common as C
    from ivid_amd import parallel
    from ivid_amd.diffusion import frameworks, samplers
    from ivid_amd.diffusion.backbones import AdmUnet2d

    rank, world = parallel.init_from_env()
    assert world == max(1, a.gpus) or world == 1, f"--gpus {a.gpus} but WORLD_SIZE {world}"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0)
    torch.cuda.set_device(dev)

    margs = dict(C.LARGE128 if a.model == "large" else C.SMALL128)
    schema = C.schema_for(margs)
    sd = C.synth_weights(margs, 0) if rank == 0 else None
    sd = parallel.broadcast_state_dict(schema, sd, device=dev)       # one RCCL broadcast over xGMI
    model = AdmUnet2d(**margs, precision=a.precision)
    model.load_state_dict(sd, strict=True)
    del sd
    model = model.to(dev).eval()
    has_cls = margs["num_classes"] is not None
    fw = (frameworks.ClassifierFreeGuidance(model, timesteps=1000, beta_schedule="linear", p_uncond=0.1)
          if has_cls else frameworks.GaussianDiffusion(model, timesteps=1000, beta_schedule="linear"))
    smp = samplers.DdimSampler(fw)
    B = a.batch
    x = C.seeded_randn(123 + rank, B, 4, 128, 128).to(dev)
    classes = (torch.arange(B) % 1000).to(dev) if has_cls else None
    kw = dict(strength=a.guidance) if has_cls else {}
    fwd_per_step = 2 if has_cls and a.guidance > 0 else 1
    # DDIM 50-step schedule of config 2: (1000,980) ... (20,0); the bench walks it cyclically
    pairs = [(20 * (i + 1), 20 * i) for i in reversed(range(50))]

    def step(i, x):
        t, tp = pairs[i % 50]
        return smp.sample_once(x, t, tp, classes, False, 0.0, **kw).pred_x_prev

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    xi = x
    for i in range(a.warmup):
        xi = step(i, xi)
    if a.warmup < 2:  # the first call is eager, the second captures the hipGraph: keep both out of the timed region
        for i in range(2 - a.warmup):
            xi = step(i, xi)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        xi = step(a.warmup + i, xi)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(xi).all(), "non-finite samples"

    fwd_s = fwd_per_step * a.steps * world / dt
    gflop = GFLOP_PER_SAMPLE_FWD[a.model]
    peak = PEAK_TFLOPS[a.precision]
    job_tflops = fwd_s * B * gflop / 1e3

    result = {
        "metric": "denoise-steps/sec (UNet fwd/s) at 128x128 RGBD bs=64",
        "value": round(fwd_s, 4),
        "unit": "UNet fwd/s (bs=%d, 128x128 RGBD)" % B,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.precision, "data": "synthetic (seeded random-init weights, N(0,1) x_T)",
        "config": {"workload": "rgbd_imagenet_adm_128_%s_cfg uncond, DDIM 50-step schedule, bs=%d per GPU, CFG=%.2f "
                               "(1 step = %d UNet forwards stacked into one batch-%d hipGraph forward + fused DDIM update)"
                               % (a.model, B, a.guidance, fwd_per_step, B * fwd_per_step),
                   "parallelism": "sample-parallel x%d (no collective in the denoise loop)" % world},
        "denoise_steps_per_s": round(a.steps * world / dt, 4),
        "sample_fwd_per_s": round(fwd_s * B, 2),
        "job_tflops": round(job_tflops, 2),
        "mfma_roofline_frac_whole_step": round(job_tflops / world / peak, 4),
    }

    if rank == 0 and not a.no_kernel_breakdown:
        plan = model.plan(B, stacked=(fwd_per_step == 2))
        prof = plan.profile_eager()
        fam = {}
        for name, args, ms in prof:
            if name == "ivid_conv3x3_gn_out":   # the output head: same family, its own argument list
                (_dt, _s, c_, _ab, _w, _b, _o, n_, h_, w_, co_) = args
                f = fam.setdefault("ivid_conv3x3_gn", dict(ms=0.0, n=0, flop=0.0, byt=0.0))
                f["ms"] += ms; f["n"] += 1
                f["flop"] += 2.0 * n_ * h_ * w_ * co_ * 9 * c_
                f["byt"] += float(n_ * h_ * w_ * c_ * (2 if _dt == 1 else 4) + n_ * h_ * w_ * co_ * 4)
                continue
            fused = name in ("ivid_conv3x3_gn", "ivid_conv3x3_gn_skip")
            f = fam.setdefault("ivid_conv3x3_gn" if fused else name, dict(ms=0.0, n=0, flop=0.0, byt=0.0))
            f["ms"] += ms
            f["n"] += 1
            if name == "ivid_conv2d":
                fl, dt_ = conv_flops(args)
                if dt_ == (1 if a.precision == "bf16" else 0):
                    f["flop"] += fl
                f["byt"] += conv_bytes(args, False)
            elif fused:
                f["flop"] += fused_flops(args)
                f["byt"] += conv_bytes(args, True)
            elif name == "ivid_attention":
                f["flop"] += attn_flops(args)
        total_ms = sum(f["ms"] for f in fam.values())
        conv = dict(fam["ivid_conv2d"])
        if "ivid_conv3x3_gn" in fam:  # the two MFMA convolution kernels together = 97 % of the model's FLOPs
            for k in ("ms", "n", "flop", "byt"):
                conv[k] += fam["ivid_conv3x3_gn"][k]
        ach = conv["flop"] / (conv["ms"] * 1e-3) / 1e12
        result["roofline"] = {
            "kernel": "conv3x3_fused_kernel + conv_igemm_kernel (all %d convolution launches of one batch-%d forward)" % (conv["n"], plan.n),
            "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            # HBM bytes per conv-family launch from the PMC passes of THIS command (rocprofv3 --pmc FETCH_SIZE /
            # WRITE_SIZE, gfx950-corrected; scripts/gpu_pmc_bench.sh -> profiles/pmc_traffic.json); null if not collected
            "traffic": pmc_traffic(),
            "avg_launch_ms": round(conv["ms"] / conv["n"], 4),
            "algorithmic_gflop_per_launch_avg": round(conv["flop"] / conv["n"] / 1e9, 2),
            "algorithmic_bytes_per_launch_avg": round(conv["byt"] / conv["n"], 0),
            "share_of_forward_time": round(conv["ms"] / total_ms, 4),
            # `peak` is the nominal dense figure of MI355X_MICROARCH.md (2.4 GHz).  A pure register-resident MFMA stream on
            # random bf16 operands sustains only 1606 TFLOP/s on this chip (power-limited clock; scripts/micro/mfma_power.hip,
            # profiles/r01_mfma_power.txt) -- the ceiling this kernel family actually works under:
            "power_limited_mfma_peak_random_operands": 1606.0 if a.precision == "bf16" else None,
            "frac_of_power_limited_peak": round(ach / 1606.0, 4) if a.precision == "bf16" else None,
        }
        result["kernel_time_ms_per_forward"] = {k: round(v["ms"], 3) for k, v in sorted(fam.items())}
        if os.environ.get("IVID_BENCH_LAYERS"):  # per-launch table for kernel tuning (not part of the bench line)
            rows = []
            for name, args, ms in prof:
                if name == "ivid_conv2d":
                    fl, _ = conv_flops(args)
                    rows.append(dict(n=args[11], h=args[12], cin=args[2] + args[4], cout=args[14], taps=args[15],
                                     res=args[9], ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1)))
                elif name in ("ivid_conv3x3_gn", "ivid_conv3x3_gn_skip"):
                    fl = fused_flops(args)
                    rows.append(dict(fused=1, n=args[12], h=args[13], cin=args[2] + args[4], cout=args[15], taps=9, up=args[6],
                                     res=args[11], ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1)))
                elif name in ("ivid_gn_apply", "ivid_gn_partial", "ivid_attention"):
                    rows.append(dict(op=name, args=[a for a in args if isinstance(a, int) and a < (1 << 32)], ms=round(ms, 4)))
            with open(os.environ["IVID_BENCH_LAYERS"], "w") as f:
                json.dump(rows, f, indent=0)
        result["forward_ms_eager_events"] = round(total_ms, 3)

    if rank == 0 and world == 1 and not a.no_cpu_baseline:   # reported at N = 1 only (a host-side figure)
        # the oracle (a CPU restatement of the reference forward, pinned to it bit-for-bit by tests/golden) on the
        # host cores: 1 warm-up + 2 timed forwards at bs 2, fp32
        from oracle import adm_oracle
        # threads actually usable by this process (cgroup/affinity), capped: torch's CPU kernels stop scaling (and
        # oversubscribe badly) far below the 256 logical cores of the GPU host
        try:
            avail = len(os.sched_getaffinity(0))
        except Exception:
            avail = os.cpu_count()
        ncores = max(1, min(avail, int(os.environ.get("IVID_CPU_BASELINE_THREADS", "32"))))
        torch.set_num_threads(ncores)
        sd_cpu = C.synth_weights(margs, 0)
        xb = C.seeded_randn(5, 2, 4, 128, 128)
        tb = torch.full((2,), 500, dtype=torch.long)
        cb = torch.tensor([1, 2]) if has_cls else None
        adm_oracle.unet_forward(sd_cpu, margs, xb, tb, cb)
        c0 = time.perf_counter()
        nrep = 0
        while nrep < 2 or (time.perf_counter() - c0 < 10.0 and nrep < 16):   # >= 10 s of CPU work, bounded
            adm_oracle.unet_forward(sd_cpu, margs, xb, tb, cb)
            nrep += 1
        cdt = time.perf_counter() - c0
        s_fwd = nrep * 2 / cdt
        result["cpu_baseline"] = {"value": round(s_fwd / B, 5), "unit": result["unit"], "cores": ncores, "kind": "port",
                                  "sample": "oracle UNet forward (fp32 torch CPU, %s model): %d timed forwards at bs 2 in %.1f s = "
                                            "%.2f sample-fwd/s, scaled to bs-%d forwards" % (a.model, nrep, cdt, s_fwd, B),
                                  "sample_fwd_per_s": round(s_fwd, 3)}
        result["speedup_vs_cpu"] = round(fwd_s / (s_fwd / B), 1)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
