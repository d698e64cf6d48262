"""ctypes binding of libivid_hip.so (include/ivid_hip.h).

The HIP library is the product path: there is no CPU fallback.  Importing this module never
needs a GPU (the symbols are only resolved), but every compute entry point raises if the
library is missing or a launch fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IVID_HIP_LIB") or os.path.join(_HERE, "lib", "libivid_hip.so")   # override: A/B tuning builds

F32, BF16, F16, BF16X3 = 0, 1, 2, 3   # include/ivid_hip.h IVID_*
# "fp16c": the fp16 kernels with COMPENSATED storage -- the residual trunk is kept as two fp16 planes hi + lo (22 mantissa
# bits), stem and output head are evaluated in split form (include/ivid_hip.h ivid_conv2d_c).  "fp16cx" additionally feeds
# the lo planes into the fused kernels' halo transform and keeps a lo plane for the tensor between a ResBlock's two
# convolutions (9 % less deviation for 5 % more time).
# "fp16s" (round 4): fp16cx + every 1x1 skip_connection in split precision (three MFMA passes: the trunk itself is that
# convolution's operand) + the stem and the first encoder level as a split-precision island (fp32 storage, bf16 hi + lo operands,
# three MFMA passes) -- the mode that stays inside 1e-3 of the fp32 reference on clean, smooth inputs at small t too.
PRECISIONS = {"fp32": F32, "bf16": BF16, "fp16": F16, "bf16x3": BF16X3, "fp16c": F16, "fp16cx": F16, "fp16s": F16, "fp16cs": F16,
              "fp16sa": F16, "fp16sa3": F16, "fp16sx": F16}
COMPENSATED = {"fp16c": 1, "fp16cx": 2, "fp16s": 3, "fp16cs": 3, "fp16sa": 3, "fp16sa3": 3, "fp16sx": 3}
# "fp16cs" = fp16s WITHOUT its bf16x3 island (stem + first encoder level): inside the tolerance only when the input carries diffusion
# noise.  "fp16sa" (adaptive, opt-in) = fp16s, except that a forward whose caller announced a timestep >= ADAPTIVE_T
# (AdmUnet2d.note_timestep: the samplers know t on the host) runs the fp16cs plan.
# An adaptive mode is a ladder of TIERS (mode, t_min), t_min ascending: a forward announced with timestep t runs the LAST tier whose
# t_min <= t; an unannounced forward runs tier 0.  "fp16sa3" (round 5) adds a third tier: from t >= 500 the split-precision skip
# convolutions go as well (plain fp16cx) -- measured inside the tolerance there on the two unconditional 128^2 backbones only, so
# it is what bench.py's headline rule may pick for them after checking every row in the run, not what `use_fp16` selects.
NO_ISLAND = {"fp16cs"}
# "fp16sx" (round 5): the STRICT ladder -- SURVEY.md 8(c) names two parity metrics, rel-L2 and max-abs / |ref|_inf, and on these
# outputs the second is by construction ~1.5-1.7 x the first (a Gaussian error field's maximum over 65 k values against a smooth
# reference's peak), so a 16-bit forward at 8.8e-4 rel-L2 sits at 1.4e-3 in the max norm.  This ladder keeps BOTH under 1e-3 on
# every row of every forward set: bf16x3 below t = 250, fp16s up to 500, fp16cs above.
# Thresholds (timesteps of the canonical 1000-step linear schedule), read off the per-timestep deviation table of the mode ladder on
# ten synthetic checkpoints of four backbones (profiles/r05_mode_ladder_per_t.json): without the island (fp16cs) the worst row is
# 8.6e-4 at t = 150 (SR-256; 8.1e-4 on the 128^2 backbones) and 1.04e-3 at t = 50; without the split skips too (fp16cx) 8.8e-4 at t = 500 on the unconditional backbones.
ISLAND_T, SKIPS_T = 150, 500
ADAPTIVE = {"fp16sa": (("fp16s", 0), ("fp16cs", ISLAND_T)),
            "fp16sa3": (("fp16s", 0), ("fp16cs", ISLAND_T), ("fp16cx", SKIPS_T)),
            "fp16sx": (("bf16x3", 0), ("fp16s", 250), ("fp16cs", 500))}
# What `use_fp16` configs / convert_to_fp16() select (round 6: the STRICT ladder).  Its thresholds leave headroom on BOTH parity
# metrics -- worst row of every forward set 7.0e-4 rel-L2 / 7.5e-4 max-norm against the 1e-3 bar -- where the faster ladders sit
# at 8.6 - 8.8e-4 rel-L2 on thresholds read off the same ten synthetic checkpoints they are verified on (and 1.4e-3 in the max
# norm at t <= 20): those stay opt-in (`precision=` / IVID_PRECISION), and bench.py reports the strict ladder beside its headline.
DEFAULT_FP16 = "fp16sx"
# Guidance-aware tier (round 6).  With classifier-free guidance the sampler consumes (1 + s) eps_c - s eps_u; on a pure-noise input
# (the first step of a chain) the two branches nearly coincide and the combination amplifies their rounding: at the CLI's default
# strength 3.0 a 16-bit forward's guided eps deviates 1.3 - 1.8e-3 there and <= 1.3e-4 on every later recorded step
# (tests/golden/*_steps.npz, profiles/r05_parity_report.json).  A forward that a framework announces with strength s, 1 + 2 s >
# GUIDED_AMP, AND a timestep >= GUIDED_T (canonical schedule: the last 1 % = 10 of 1000 DDPM steps, 1 of 50 DDIM steps) runs the
# exact split-precision plan (bf16x3) whatever ladder is selected.
GUIDED_AMP, GUIDED_T, GUIDED_MODE = 3.0, 990, "bf16x3"
# environment overrides of the thresholds, by the rung they introduce (not by tier position)
THRESHOLD_ENV = {"fp16cs": "IVID_ADAPTIVE_T", "fp16cx": "IVID_ADAPTIVE_T2", "fp16s": "IVID_ADAPTIVE_TS"}


def esz(dtype):
    """Bytes per stored activation element (BF16X3 stores fp32)."""
    return 4 if dtype in (F32, BF16X3) else 2

vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_longlong, C.c_void_p


class DdimCoef(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "sqrt_recip_ac", "sqrt_recipm1_ac", "sqrt_ac_prev", "dir_coef", "sigma", "nonzero", "cfg_strength",
        "replace_rgb_w", "replace_depth_w", "constrain_w")] + [("clip_denoised", C.c_int)]


class DdpmCoef(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "sqrt_recip_ac", "sqrt_recipm1_ac", "coef1", "coef2", "std", "cfg_strength")] + [("clip_denoised", C.c_int)]


class SamplePlan(C.Structure):      # ivid_sample_plan
    _fields_ = [("kind", C.c_int), ("n_steps", C.c_int), ("hw", C.c_int), ("t_model", C.POINTER(C.c_longlong)),
                ("coef", C.c_void_p), ("engine_of_step", C.POINTER(C.c_int)), ("generate_noise", C.c_int), ("first_step", C.c_int),
                ("noise_seed", C.c_ulonglong)]


class SampleCond(C.Structure):      # ivid_sample_cond
    _fields_ = ([(n, C.c_void_p) for n in ("y", "mask", "mask_rgb", "hole_noise", "rgb", "rgb_mask", "depth", "depth_mask", "convex",
                                            "sr_y")] + [("sr_channels", C.c_int), ("sr_size", C.c_int)])


SAMPLE_DDIM, SAMPLE_DDPM = 0, 1

# name -> (restype, argtypes); must list every symbol declared in include/ivid_hip.h
SIGNATURES = {
    "ivid_last_error": (C.c_char_p, []),
    "ivid_version": (i32, []),
    "ivid_graph_begin": (i32, [vp]),
    "ivid_graph_end": (i32, [vp, C.POINTER(vp)]),
    "ivid_graph_launch": (i32, [vp, vp]),
    "ivid_graph_destroy": (i32, [vp]),
    "ivid_event_create": (i32, [C.POINTER(vp)]),
    "ivid_event_record": (i32, [vp, vp]),
    "ivid_event_elapsed_ms": (i32, [vp, vp, C.POINTER(C.c_float)]),
    "ivid_event_destroy": (i32, [vp]),
    "ivid_program_create": (i32, [C.POINTER(vp)]),
    "ivid_program_add": (i32, [vp, i32, vp, i32]),
    "ivid_program_op_arity": (i32, [i32]),
    "ivid_program_num_ops": (i32, [vp]),
    "ivid_program_launch": (i32, [vp, i32, vp]),
    "ivid_program_has_graph": (i32, [vp]),
    "ivid_program_destroy": (i32, [vp]),
    "ivid_unet_bind": (i32, [vp, vp, i64, vp, vp, i32, vp, i64]),
    "ivid_unet_forward": (i32, [vp, vp, vp, vp, vp, i32, vp]),
    "ivid_unet_load": (i32, [vp, i64, vp]),
    "ivid_unet_info": (i32, [vp, vp, vp, vp, vp, vp]),
    "ivid_conv2d": (i32, [i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_conv2d_stats_block": (i32, [i32, i32, i32, i32, i32]),
    "ivid_conv2d_c": (i32, [i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_conv3x3_gn_skip_c": (i32, [i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp,
                                     vp, i32, vp, i32, vp, vp]),
    "ivid_conv3x3_gn_skip_s": (i32, [i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp,
                                     vp, i32, vp, i32, vp, vp, vp, vp, vp]),
    "ivid_f32_to_hilo": (i32, [i32, vp, vp, vp, i64, vp]),
    "ivid_conv2d_o16": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_conv3x3_gn_o16": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_conv3x3_gn_out_c": (i32, [i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ivid_gn_apply_c": (i32, [i32, vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ivid_gn_apply_p": (i32, [i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ivid_stem_im2col_split": (i32, [i32, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_conv3x3_up": (i32, [i32, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_conv3x3_gn": (i32, [i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_conv3x3_gn_out": (i32, [i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ivid_conv3x3_gn_skip": (i32, [i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp,
                                   vp, i32, vp, i32, vp, vp]),
    "ivid_gn_num_chunks": (i32, [i32]),
    "ivid_gn_partial": (i32, [i32, vp, i32, vp, i32, i32, i32, vp, vp]),
    "ivid_gn_partial_c": (i32, [i32, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp]),
    "ivid_gn_finalize": (i32, [vp, i32, i32, i32, i32, i32, C.c_float, vp, vp, vp, i32, i32, vp, vp]),
    "ivid_gn_finalize2": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, C.c_float, vp, vp, vp, i32, i32, vp, vp]),
    "ivid_gn_apply": (i32, [i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ivid_attention": (i32, [i32, vp, vp, i32, i32, i32, vp]),
    "ivid_embed_inputs": (i32, [vp, vp, i32, i32, i32, vp, i32, vp, i32, vp, vp, vp]),
    "ivid_silu_f32": (i32, [vp, vp, i64, vp]),
    "ivid_copy": (i32, [vp, vp, i64, vp]),
    "ivid_nchw_to_nhwc": (i32, [i32, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_stem_im2col": (i32, [i32, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ivid_randn": (i32, [C.c_ulonglong, C.c_ulonglong, vp, i64, vp]),
    "ivid_sample_scratch_bytes": (i64, [vp, i32, C.POINTER(SamplePlan), C.POINTER(SampleCond)]),
    "ivid_sample": (i32, [vp, i32, C.POINTER(SamplePlan), vp, C.POINTER(SampleCond), vp, vp, vp, vp, i64, vp]),
    "ivid_ddim_step": (i32, [vp, vp, vp, C.POINTER(DdimCoef), vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "ivid_ddpm_step": (i32, [vp, vp, vp, C.POINTER(DdpmCoef), vp, vp, vp, i32, i32, vp]),
    "ivid_inpaint_cond": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "ivid_sr_cond": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ivid_mesh_build": (i32, [vp, i32, i32, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, C.c_float,
                              i32, vp, vp, vp, vp, vp, vp]),
    "ivid_warp_render": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, i32, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, i32,
                               vp, vp]),
    "ivid_simple_render": (i32, [vp, vp, vp, i32, i32, vp, i32, C.c_float, C.c_float, vp, vp, i32, vp, vp, vp, vp]),
    "ivid_resample8_lanczos": (i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp]),
    "ivid_warp_resolve": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, C.c_float, C.c_float, C.c_float,
                                C.c_float, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
}

# op codes of the launch program (include/ivid_hip.h IVID_OP_*)
OP_CODES = {"ivid_conv2d": 1, "ivid_conv3x3_gn": 2, "ivid_conv3x3_gn_skip": 3, "ivid_conv3x3_gn_out": 4, "ivid_gn_partial": 5,
            "ivid_gn_finalize": 6, "ivid_gn_finalize2": 7, "ivid_gn_apply": 8, "ivid_attention": 9, "ivid_embed_inputs": 10,
            "ivid_silu_f32": 11, "ivid_stem_im2col": 12, "ivid_conv3x3_up": 13, "ivid_copy": 14, "ivid_conv2d_c": 15,
            "ivid_conv3x3_gn_skip_c": 16, "ivid_gn_apply_c": 17, "ivid_conv3x3_gn_out_c": 18, "ivid_stem_im2col_split": 19,
            "ivid_conv3x3_gn_skip_s": 20, "ivid_f32_to_hilo": 21, "ivid_gn_apply_p": 22, "ivid_conv3x3_gn_o16": 23, "ivid_gn_partial_c": 24, "ivid_conv2d_o16": 25}


ENGINE_ABI = 5   # include/ivid_hip.h IVID_ENGINE_ABI


class Slot(C.Union):
    _fields_ = [("i", C.c_longlong), ("f", C.c_double)]


def pack_args(name, args):
    """Arguments of one recorded launch (without the stream) -> the 8-byte slots ivid_program_add expects: floats (per the
    entry point's ctypes signature) as double, everything else (ints, device pointers, None) as int64."""
    sig = SIGNATURES[name][1][:-1]
    assert len(sig) == len(args), (name, len(sig), len(args))
    arr = (Slot * len(args))()
    for k, (ty, v) in enumerate(zip(sig, args)):
        if ty is C.c_float:
            arr[k].f = float(v)
        else:
            arr[k].i = 0 if v is None else int(v)
    return arr


_lib = None


class IvidHipError(RuntimeError):
    pass


def load():
    """Load the HIP library; raise loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IvidHipError(
            f"{LIB_PATH} not found: build it with `python -m ivid_amd.build` (hipcc, gfx950). "
            "ivid_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().ivid_last_error()
        raise IvidHipError(f"{what}: {msg.decode() if msg else status}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return t.data_ptr()


def call(name, *args):
    check(getattr(load(), name)(*args), name)
