"""ctypes binding of the C ABI in include/aewavenet.h.

The structures below mirror the header field-for-field; `load()` checks their sizes against
`aew_sizeof()` so a drifted mirror fails loudly.  There is no CPU fallback: if the shared
library is missing, `load()` raises and nothing in the product path can run.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_SEGS = 32
BF16, F32 = 0, 1

E_ARG, E_UNSUP, E_ALIGN = -1, -2, -3

EPI_STORE, EPI_GATED, EPI_RES_SKIP, EPI_DFG = 0, 1, 2, 3
EF_BIAS, EF_RELU, EF_OUT1_PRE, EF_ADD_AUX0 = 1 << 0, 1 << 1, 1 << 2, 1 << 3
EF_MUL_POS1, EF_OUT1_POS1, EF_ACCUM, EF_OUT2_RELU, EF_COUNT_ZERO = 1 << 4, 1 << 5, 1 << 6, 1 << 7, 1 << 8
EF_RELU_POST = 1 << 9
EF_OUT2_COPY = 1 << 10

(OP_GEMM_NT, OP_GEMM_TN, OP_COPY_TABLE, OP_VQ_NEAREST, OP_VQ_STATS, OP_VQ_EMA, OP_VQ_BWD,
 OP_LC_GATHER, OP_LC_SCATTER, OP_SPK_BIAS, OP_SPK_BWD, OP_BASE_GATHER, OP_SOFTMAX_NLL, OP_COLSUM,
 OP_REDUCE, OP_ADAM, OP_ZERO, OP_VAE, OP_AE_NORM, OP_JITTER, OP_VQ_DIAG, OP_MFCC, OP_MOMENTS, OP_GEMM_TN_GROUP,
 OP_NT_CHAIN) = range(1, 26)

vp, i32, i64, u32, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float


class Seg(C.Structure):
    _fields_ = [("ptr", vp), ("batch_stride", i64), ("row_pitch", i32), ("row_step", i32),
                ("row_off", i32), ("row_lo", i32), ("row_hi", i32), ("k_len", i32)]


class View(C.Structure):
    _fields_ = [("ptr", vp), ("batch_stride", i64), ("row_pitch", i32), ("row_step", i32),
                ("row_off", i32), ("row_lo", i32), ("row_hi", i32), ("dtype", i32)]


class GemmNT(C.Structure):
    _fields_ = [("dtype", i32), ("impl", i32), ("M", i32), ("N", i32), ("N_pad", i32),
                ("batch", i32), ("n_segs", i32), ("K_total", i32), ("seg", Seg * MAX_SEGS),
                ("W", vp), ("epi", i32), ("flags", u32), ("out0", View), ("out1", View),
                ("out2", View), ("aux0", View), ("aux1", View), ("bias", vp), ("bias_bs", i64),
                ("n_split", i32), ("reserved", i32), ("counter", vp),
                ("W2", vp), ("N2", i32), ("N2_pad", i32), ("out3", View),
                ("k_split", i32), ("pad2_", i32), ("ksplit_ws", vp), ("ksplit_tickets", vp)]


class GemmTN(C.Structure):
    _fields_ = [("dtype", i32), ("impl", i32), ("Mc", i32), ("batch", i32), ("N", i32),
                ("N_pad", i32), ("g", Seg), ("n_segs", i32), ("K_total", i32),
                ("seg", Seg * MAX_SEGS), ("out", vp), ("out_batch_stride", i64),
                ("snap_out", vp), ("snap_bs", i64), ("snap_k", i32), ("pad_", i32), ("colsum_out", vp),
                ("grp_splits", i32), ("grp_rows", i32)]


class GemmTNGroup(C.Structure):
    _fields_ = [("descs", vp), ("tile_map", vp), ("n_descs", i32), ("n_blocks", i32), ("tile", i32), ("cursor_stride", i32),
                ("cursors", vp)]


CHAIN_MAXDEP = 3


class ChainDep(C.Structure):
    _fields_ = [("cnt_base", i32), ("n_mt", i32), ("need", i32), ("d_lo", i32), ("d_hi", i32), ("c_lo", i32), ("c_hi", i32),
                ("bm", i32)]


class NtStage(C.Structure):
    """aew_nt_stage_t: one stage of a chained NT launch (filled in by aew_nt_chain_build, uploaded to device memory)."""
    _fields_ = [("g", GemmNT), ("kind", i32), ("first_block", i32), ("n_blocks", i32), ("n_mt", i32), ("n_nt", i32),
                ("cnt_base", i32), ("publish", i32), ("n_deps", i32), ("dep", ChainDep * CHAIN_MAXDEP)]


class NtChain(C.Structure):
    _fields_ = [("stages", vp), ("block_stage", vp), ("counters", vp), ("n_stages", i32), ("n_blocks", i32),
                ("n_counters", i32), ("set", i32), ("n_ops", i32), ("spin_max", i32), ("flags", i32), ("built_window", i32),
                ("sticky", vp)]


class CopyRec(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("dims", i32 * 4), ("ss", i64 * 4), ("ds", i64 * 4),
                ("src_dtype", i32), ("dst_dtype", i32), ("red_n", i32), ("accumulate", i32),
                ("red_stride", i64), ("scale", f32), ("first_block", i32), ("tr_a", i32), ("tr_b", i32)]


class CopyTable(C.Structure):
    _fields_ = [("recs", vp), ("block_rec", vp), ("n_blocks", i32), ("n_recs", i32)]


class VqNearest(C.Structure):
    _fields_ = [("ze", vp), ("emb", vp), ("Q", i32), ("K", i32), ("d", i32), ("d_pitch", i32),
                ("metric", i32), ("ind", vp), ("dist", vp), ("zq", vp), ("scratch", vp), ("n_split", i32)]


class VqStats(C.Structure):
    _fields_ = [("ze", vp), ("ind", vp), ("Q", i32), ("K", i32), ("d", i32), ("d_pitch", i32),
                ("z_sum", vp), ("n_sum", vp), ("hist", vp)]


class VqEma(C.Structure):
    _fields_ = [("numer", vp), ("denom", vp), ("z_sum", vp), ("n_sum", vp), ("emb", vp),
                ("K", i32), ("d", i32), ("update_codebook", i32), ("gamma", f32),
                ("gamma_comp", f32), ("guard", vp)]


class VqBwd(C.Structure):
    _fields_ = [("ze", vp), ("emb", vp), ("ind", vp), ("dzq", vp), ("Q", i32), ("d", i32),
                ("d_pitch", i32), ("metric", i32), ("coef", f32), ("demb_coef", f32),
                ("dze", vp), ("demb", vp), ("gmul", vp)]


class LcGather(C.Structure):
    _fields_ = [("src", vp), ("src_bs", i64), ("src_pitch", i32), ("jitter", vp),
                ("jit_pitch", i32), ("dst", vp), ("dst_bs", i64), ("dst_pitch", i32),
                ("B", i32), ("N", i32), ("C", i32), ("C_pad", i32), ("take_compat", i32)]


class LcScatter(C.Structure):
    _fields_ = [("d", vp), ("d_bs", i64), ("d_pitch", i32), ("jitter", vp), ("jit_pitch", i32),
                ("dsrc", vp), ("dsrc_bs", i64), ("dsrc_pitch", i32), ("B", i32), ("N", i32),
                ("C", i32), ("take_compat", i32)]


class SpkBias(C.Structure):
    _fields_ = [("params", vp), ("voice", vp), ("off_bias_sig", vp), ("off_bias_gate", vp),
                ("off_proj_sig", vp), ("off_proj_gate", vp), ("off_spk_w", i64),
                ("off_spk_b", i64), ("B", i32), ("L", i32), ("D", i32), ("D_pad", i32),
                ("C_lc", i32), ("G", i32), ("n_speakers", i32), ("bias", vp), ("gc", vp)]


class SpkBwd(C.Structure):
    _fields_ = [("params", vp), ("voice", vp), ("off_bias_sig", vp), ("off_bias_gate", vp),
                ("off_proj_sig", vp), ("off_proj_gate", vp), ("off_spk_w", i64),
                ("off_spk_b", i64), ("B", i32), ("L", i32), ("D", i32), ("D_pad", i32),
                ("C_lc", i32), ("G", i32), ("n_speakers", i32), ("colsum", vp), ("gc", vp),
                ("grads", vp), ("colsum_running", i32), ("layer_range", i32), ("det_scratch", vp), ("det_tickets", vp)]


class Tuning(C.Structure):
    """aew_tuning_t: kernel-shape choices as a record (the aew_set_* switches edit the process-wide one; Plan.run(...,
    tuning=...) / aew_run_plan_tuned apply a caller's own to one call)."""
    _fields_ = [("nt_wave_rows", i32), ("nt_pipe", i32), ("nt_rows192", i32), ("nt_small_tiles", i32), ("nt_small_n64", i32), ("nt_small_w8", i32), ("nt_small_deep", i32), ("nt_window", i32), ("nt_mem128", i32), ("nt_deep", i32), ("nf_loaders", i32), ("nf_deep", i32), ("fn_enable", i32), ("fn_ring3", i32), ("tn_safe", i32), ("tn_big", i32), ("tn_big_target", i32), ("tn_fold_rows", i32), ("tn_target_blocks", i32), ("tn_small_tiles", i32), ("tn_small_target", i32), ("lanes", i32), ("tn_cursor_epoch", i32), ("tn_cursor_slack", i32), ("nt_chain", i32), ("deterministic", i32), ("tn_mfma32", i32), ("reserved_", i32 * 5)]


class BaseGather(C.Structure):
    _fields_ = [("wav", vp), ("wav_pitch", i32), ("wav_off", i32), ("W", vp), ("bias", vp),
                ("Wt", vp), ("B", i32), ("T", i32), ("R", i32), ("R_pad", i32), ("Q", i32), ("x", vp),
                ("x_bs", i64), ("x_pitch", i32), ("onehot", vp), ("oh_bs", i64),
                ("oh_pitch", i32), ("Q_pad", i32), ("ones_channel", i32)]


class SoftmaxNll(C.Structure):
    _fields_ = [("logits", vp), ("bs", i64), ("pitch", i32), ("wav", vp), ("wav_pitch", i32),
                ("tgt_off", i32), ("B", i32), ("w", i32), ("Q", i32), ("Q_pad", i32),
                ("nll", vp), ("ptgt", vp), ("dlogits", vp), ("dl_bs", i64), ("dl_pitch", i32),
                ("scale", f32), ("backward", i32), ("gmul", vp), ("peak", vp), ("amax", vp)]


class Colsum(C.Structure):
    _fields_ = [("x", Seg), ("dtype", i32), ("M", i32), ("N", i32), ("batch", i32),
                ("out", vp), ("out_bs", i64), ("accumulate", i32), ("pad_", i32), ("det_scratch", vp), ("det_tickets", vp)]


class Reduce(C.Structure):
    _fields_ = [("x", vp * 4), ("n", i32 * 4), ("scale", f32 * 4), ("clamp", i32 * 4),
                ("clamp_min", f32 * 4), ("post_scale", f32 * 4), ("n_terms", i32), ("out", vp),
                ("post_scale_dev", vp * 4)]


class Adam(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64), ("lr", f32),
                ("beta1", f32), ("beta2", f32), ("eps", f32), ("bc1", f32), ("bc2", f32),
                ("grad_scale", f32), ("pad_", i32), ("guard", vp)]


class Zero(C.Structure):
    _fields_ = [("ptr", vp), ("bytes", i64)]


class Vae(C.Structure):
    _fields_ = [("lin", vp), ("lin_pitch", i32), ("eps", vp), ("Q", i32), ("d", i32),
                ("d_pitch", i32), ("sample", vp), ("kl_terms", vp), ("dsample", vp),
                ("kl_coef", f32), ("kl_value", vp), ("free_nats", f32), ("dlin", vp),
                ("backward", i32), ("kl_coef_dev", vp), ("gmul", vp)]


class AeNorm(C.Structure):
    _fields_ = [("ze", vp), ("Q", i32), ("d", i32), ("d_pitch", i32), ("term", vp),
                ("dze_in", vp), ("coef", f32), ("dze", vp), ("backward", i32), ("gmul", vp)]


class Jitter(C.Structure):
    _fields_ = [("out", vp), ("out_pitch", i32), ("B", i32), ("n", i32), ("p", f32), ("mode", i32),
                ("seed", C.c_uint64), ("step", C.c_uint64)]


class VqDiag(C.Structure):
    _fields_ = [("ze", vp), ("Q", i32), ("d", i32), ("d_pitch", i32), ("emb", vp), ("K", i32), ("hist", vp),
                ("n_sum", vp), ("logits", vp), ("bs", i64), ("pitch", i32), ("B", i32), ("w", i32),
                ("n_quant", i32), ("scratch", vp), ("out", vp), ("peak", vp), ("amax", vp)]


class Mfcc(C.Structure):
    _fields_ = [("wav", vp), ("wav_bs", i64), ("n", i32), ("B", i32), ("win", i32), ("hop", i32), ("n_bins", i32),
                ("n_mels", i32), ("n_mfcc", i32), ("left_pad", i32), ("trim_left", i32), ("trim_right", i32),
                ("n_frames", i32), ("window", vp), ("twiddle", vp), ("melw", vp), ("dct", vp), ("sg", vp),
                ("scratch", vp), ("out", vp), ("out_bs", i64), ("out_pitch", i32)]


class Moments(C.Structure):
    _fields_ = [("x", View), ("rows", i32), ("cols", i32), ("batch", i32), ("out", vp)]


class _OpU(C.Union):
    _fields_ = [("nt", GemmNT), ("tn", GemmTN), ("copy", CopyTable), ("vqn", VqNearest),
                ("vqs", VqStats), ("vqe", VqEma), ("vqb", VqBwd), ("lcg", LcGather),
                ("lcs", LcScatter), ("spk", SpkBias), ("spkb", SpkBwd), ("base", BaseGather),
                ("sm", SoftmaxNll), ("cs", Colsum), ("red", Reduce), ("adam", Adam),
                ("zero", Zero), ("vae", Vae), ("aen", AeNorm), ("jit", Jitter), ("diag", VqDiag), ("mfcc", Mfcc),
                ("mom", Moments), ("tng", GemmTNGroup), ("chain", NtChain)]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("tag", i32), ("lane", i32), ("join", i32), ("u", _OpU)]


OP_FIELD = {OP_GEMM_NT: "nt", OP_GEMM_TN: "tn", OP_COPY_TABLE: "copy", OP_VQ_NEAREST: "vqn",
            OP_VQ_STATS: "vqs", OP_VQ_EMA: "vqe", OP_VQ_BWD: "vqb", OP_LC_GATHER: "lcg",
            OP_LC_SCATTER: "lcs", OP_SPK_BIAS: "spk", OP_SPK_BWD: "spkb",
            OP_BASE_GATHER: "base", OP_SOFTMAX_NLL: "sm", OP_COLSUM: "cs", OP_REDUCE: "red",
            OP_ADAM: "adam", OP_ZERO: "zero", OP_VAE: "vae", OP_AE_NORM: "aen", OP_JITTER: "jit",
            OP_VQ_DIAG: "diag", OP_MFCC: "mfcc", OP_MOMENTS: "mom", OP_GEMM_TN_GROUP: "tng", OP_NT_CHAIN: "chain"}

# ---- autoregressive sampler (aew_actor_t / aew_sampler_t) ----
ACT_NONE, ACT_EARLY, ACT_LATE, ACT_RES, ACT_SKIP, ACT_POST1, ACT_POST2, ACT_SAMPLE = -1, 0, 1, 2, 3, 4, 5, 6


class Sbuf(C.Structure):
    _fields_ = [("ptr", vp), ("bstride", i64), ("entry", i64), ("pitch", i64), ("ring", i32), ("layout", i32)]


class Wait(C.Structure):
    _fields_ = [("flags", vp), ("n", i32), ("lag", i32)]


class Actor(C.Structure):
    _fields_ = [("role", i32), ("layer", i32), ("index", i32), ("nt", i32), ("nk", i32), ("nk2", i32),
                ("dil", i32), ("pad", i32), ("wait", Wait * 2), ("flag", vp), ("w", vp), ("w2", vp),
                ("bias", vp), ("bias_pitch", i64), ("in0", Sbuf), ("in1", Sbuf), ("out", Sbuf), ("out2", Sbuf),
                ("n_quant", i32), ("row_bytes", i32)]


class Sampler(C.Structure):
    _fields_ = [("actors", vp), ("n_slots", i32), ("n_batches", i32), ("n_steps", i32), ("flag_stride", i32),
                ("kr_max", i32), ("spin_max", i32), ("flags", vp), ("status", vp), ("forced", vp),
                ("wav_out", vp), ("seed", C.c_uint64), ("nap_eighths", i32), ("pad", i32), ("prof", vp)]


LIB_PATH = os.environ.get("AEW_LIB_PATH") or \
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libaewavenet_hip.so")   # (AEW_LIB_PATH: tools-only builds)
_lib = None


class AewError(RuntimeError):
    pass


def load():
    """Load libaewavenet_hip.so (built by __graft_entry__.build()).  Raises if it is missing
    or if the struct mirrors above have drifted from the header."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as /opt/rocm's, which this
    # library is linked against).  Whichever is loaded first serves both, and torch does not find its devices
    # through the system copy ("no ROCm-capable device"), so torch has to be imported before the dlopen below.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AewError(f"HIP extension not built: {LIB_PATH} is missing "
                       "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                       "There is no CPU fallback for the product path.")
    if torch.cuda.is_available():
        arch = getattr(torch.cuda.get_device_properties(0), "gcnArchName", "")
        if arch and not arch.startswith("gfx950"):
            raise AewError(f"libaewavenet_hip.so is built for gfx950 (MI355X) only; device 0 is {arch}")
    lib = C.CDLL(LIB_PATH)
    lib.aew_strerror.restype = C.c_char_p
    lib.aew_run_plan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    lib.aew_timing_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.aew_selftest.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.aew_tn_slabs.argtypes = [C.c_void_p]
    lib.aew_tn_fold.argtypes = [C.c_void_p]
    lib.aew_tn_group_check.argtypes = [C.c_void_p]
    lib.aew_nt_kernel.argtypes = [C.c_void_p]
    lib.aew_graph_capture.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.aew_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
    lib.aew_graph_destroy.argtypes = [C.c_void_p]
    lib.aew_sampler_run.argtypes = [C.c_void_p, C.c_void_p]
    lib.aew_run_plan_tuned.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    lib.aew_graph_capture_tuned.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
    for fn in (lib.aew_tuning_default, lib.aew_tuning_get, lib.aew_tuning_set):
        fn.argtypes = [C.c_void_p]
    lib.aew_nt_chain_build.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.c_int]
    lib.aew_nt_chain_build_tuned.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.POINTER(C.c_int), C.c_int, C.c_void_p]
    lib.aew_probe_box.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.aew_nt_chain_dep_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.aew_gemm_nt_small_split.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    lib.aew_colsum_det_size.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    for which, cls in ((0, Op), (1, GemmNT), (2, GemmTN), (3, Seg), (4, View), (5, CopyRec), (6, Actor), (7, Sampler), (8, Tuning),
                       (9, NtStage), (10, NtChain)):
        want = lib.aew_sizeof(which)
        if want != C.sizeof(cls):
            raise AewError(f"ABI mirror drift: sizeof({cls.__name__}) = {C.sizeof(cls)} in Python, "
                           f"{want} in the library")
    if lib.aew_abi_version() != 20:
        raise AewError("ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what="aew call", fail_index=None):
    if rc != 0:
        msg = load().aew_strerror(rc).decode()
        at = f" at op {fail_index}" if fail_index is not None else ""
        raise AewError(f"{what} failed{at}: rc={rc} ({msg})")


def default_tuning(**over) -> "Tuning":
    """The library's default tuning record, with fields overridden by keyword."""
    t = Tuning()
    check(load().aew_tuning_default(C.byref(t)), "aew_tuning_default")
    for k, v in over.items():
        if not hasattr(t, k):
            raise AttributeError(f"aew_tuning_t has no field {k}")
        setattr(t, k, int(v))
    return t


def current_tuning(**over) -> "Tuning":
    """A copy of the process-wide tuning record (what the aew_set_* switches have made of it), fields overridden by keyword."""
    t = Tuning()
    check(load().aew_tuning_get(C.byref(t)), "aew_tuning_get")
    for k, v in over.items():
        if not hasattr(t, k):
            raise AttributeError(f"aew_tuning_t has no field {k}")
        setattr(t, k, int(v))
    return t


def tn_slabs(tn: GemmTN) -> int:
    """Number of fp32 partial slabs a TN op writes (host-side mirror is the library itself so
    the split heuristic has a single definition)."""
    return load().aew_tn_slabs(C.byref(tn))


EXPORTS = ("aew_abi_version", "aew_sizeof", "aew_run_plan", "aew_timing_enable",
           "aew_timing_read", "aew_strerror", "aew_selftest", "aew_tn_slabs", "aew_set_tn_safe",
           "aew_graph_capture", "aew_graph_launch", "aew_graph_destroy", "aew_tn_fold", "aew_set_tn_fold_rows",
           "aew_set_lanes", "aew_set_tn_cursor", "aew_set_nt_wave_rows", "aew_set_nt_pipe",
           "aew_set_tn_target_blocks", "aew_set_tn_small", "aew_set_nt_small_tiles", "aew_set_nt_small_deep", "aew_set_nt_small_waves", "aew_set_nf_deep", "aew_set_nf_loaders", "aew_set_nt_rows192",
           "aew_sampler_run", "aew_set_fn", "aew_nt_kernel", "aew_set_tn_big", "aew_set_nt_window", "aew_set_fn_ring3", "aew_set_nt_small_n64", "aew_tn_group_check", "aew_set_nt_mem128", "aew_set_nt_deep", "aew_tuning_default", "aew_tuning_get",
           "aew_tuning_set", "aew_run_plan_tuned", "aew_graph_capture_tuned", "aew_nt_chain_build", "aew_nt_chain_dep_tiles", "aew_probe_box", "aew_gemm_nt_small_split", "aew_colsum_det_size", "aew_nt_chain_build_tuned")
