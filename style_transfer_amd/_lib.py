"""ctypes binding of libst_kernels.so (include/st_kernels.h).  Fails loudly when the HIP library is
missing: there is no CPU or PyTorch fallback for the product path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ST_KERNELS_LIB") or os.path.join(HERE, "libst_kernels.so")  # override: A/B kernel builds

ST_MAX_LAYERS = 4
ST_MAX_REFS = 16
ST_ARGMAX_SPLITS = 4
ST_NOALLOC_MAX_ROWS = 65536
ST_F_NOT_NLL, ST_F_SIMPLEST, ST_F_JUST_USE_ONE = 1, 2, 4
ST_F_OPT_SHIFT = 8
ST_OPTIMIZERS = {"Adam": 0, "SGD": 1, "AdamW": 2, "Adamax": 3, "RMSprop": 4, "Adagrad": 5,
                 "NAdam": 6, "RAdam": 7, "Adadelta": 8, "ASGD": 9, "Rprop": 10}   # torch.optim names (every first-order rule; LBFGS and SparseAdam are refused)
ST_DEC_EARLY_EXIT = 1
ST_DEC_NO_SCREEN = 2
ST_DEC_NO_L0_TABLE = 4
ST_PROF_OFF, ST_PROF_GENERATOR, ST_PROF_HALF_LAYER = 0, 1, 2
ST_GEMM_F32, ST_GEMM_F16X3 = 0, 1
GEMM_FORMATS = {"f32": ST_GEMM_F32, "f16x3": ST_GEMM_F16X3}
ABI_VERSION = 8

fp = C.c_void_p  # device pointers travel as integers


class StScorerWeights(C.Structure):
    _fields_ = [(n, fp) for n in (
        "s1_w", "s1_wt", "s1_b", "s2_w", "s2_wt", "s2_b", "s3_w", "s3_wt", "s3_b",
        "p1_w", "p1_wt", "p1_b", "p2_w", "p2_wt", "p2_b")] + [("gemm_format", C.c_int32), ("unscale", C.c_float * 5)]


class StDecLayer(C.Structure):
    _fields_ = [(n, fp) for n in (
        "ln0_a", "ln0_b", "wqkv", "bqkv", "wo", "bo", "xv_w", "xv_b", "xo_w", "xo_b",
        "ln2_a", "ln2_b", "w1", "b1", "w2", "b2")]


class StDecoderWeights(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_heads", C.c_int32), ("d_ff", C.c_int32),
                ("n_layers", C.c_int32), ("vocab", C.c_int32), ("pe_rows", C.c_int32),
                ("gemm_format", C.c_int32), ("gen_w_maxnorm", C.c_float),
                ("emb", fp), ("pe", fp), ("lnf_a", fp), ("lnf_b", fp), ("gen_w", fp), ("gen_b", fp), ("gen_w_norm", fp),
                ("layer", StDecLayer * ST_MAX_LAYERS),
                ("l0_table", fp), ("l0_table_len", C.c_int32), ("reserved1", C.c_int32)]


class StEncLayer(C.Structure):
    _fields_ = [(n, fp) for n in (
        "ln0_a", "ln0_b", "wqkv", "bqkv", "wo", "bo", "ln1_a", "ln1_b", "w1", "b1", "w2", "b2")]


class StEncoderWeights(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_heads", C.c_int32), ("d_ff", C.c_int32),
                ("n_layers", C.c_int32), ("vocab", C.c_int32), ("pe_rows", C.c_int32),
                ("gemm_format", C.c_int32), ("l0_table_len", C.c_int32),
                ("emb", fp), ("pe", fp), ("lnf_a", fp), ("lnf_b", fp),
                ("layer", StEncLayer * ST_MAX_LAYERS), ("l0_table", fp)]


class StAeLayer(C.Structure):
    _fields_ = [(n, fp) for n in (
        "ln0_a", "ln0_b", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "xv_w", "xv_b", "xo_w", "xo_b",
        "ln2_a", "ln2_b", "w1", "b1", "w2", "b2")]


class StAeParams(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_heads", C.c_int32), ("d_ff", C.c_int32),
                ("n_layers", C.c_int32), ("vocab", C.c_int32), ("pe_rows", C.c_int32),
                ("emb", fp), ("pe", fp), ("enc_norm_a", fp), ("enc_norm_b", fp), ("dec_norm_a", fp), ("dec_norm_b", fp),
                ("gen_w", fp), ("gen_b", fp),
                ("enc", StAeLayer * ST_MAX_LAYERS), ("dec", StAeLayer * ST_MAX_LAYERS)]


class StClassifierWeights(C.Structure):
    _fields_ = [(n, fp) for n in ("w1", "b1", "w2", "b2", "w3", "b3")]


EXPORTS = {
    # name: (restype, argtypes)
    "st_fgim_perturb": (C.c_int, [C.POINTER(StClassifierWeights), fp, fp, fp, C.c_int, C.c_int, fp, fp, fp, fp]),
    "st_abi_version": (C.c_int, []),
    "st_last_error": (C.c_char_p, []),
    "st_profile_enable": (C.c_int, [C.c_int]),
    "st_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]),
    "st_work_counters_read": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "st_packed_weight_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "st_pack_weight": (C.c_int, [fp, C.c_int, C.c_int, C.c_int, fp, fp]),
    "st_scorer_status": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "st_scorer_embed": (C.c_int, [C.POINTER(StScorerWeights), fp, fp, C.c_int, C.c_int, fp]),
    "st_scorer_edit_step": (C.c_int, [C.POINTER(StScorerWeights), fp, fp, fp, fp, fp, C.c_int, C.c_int, fp, fp, fp,
                                      C.c_int, C.c_float, C.c_int, fp, fp, fp, fp]),
    "st_decode_l0_table_bytes": (C.c_size_t, [C.POINTER(StDecoderWeights), C.c_int]),
    "st_decode_build_l0_table": (C.c_int, [C.POINTER(StDecoderWeights), C.c_int, fp, fp, C.c_size_t, fp]),
    "st_decode_workspace_bytes": (C.c_size_t, [C.POINTER(StDecoderWeights), C.c_int, C.c_int]),
    "st_decode_workspace_bytes_opt": (C.c_size_t, [C.POINTER(StDecoderWeights), C.c_int, C.c_int, C.c_int]),
    "st_decode_greedy": (C.c_int, [C.POINTER(StDecoderWeights), fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, fp, fp, fp, C.c_size_t, C.POINTER(C.c_int), fp]),
    "st_gemm_argmax": (C.c_int, [fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp]),
    "st_gemm_argmax_noalloc": (C.c_int, [fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp]),
    "st_accept_changes": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, fp,
                                    fp, fp, fp, fp, fp]),
    "st_simloss_format": (C.c_int, [C.c_int]),
    "st_simloss_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "st_simloss_fwd": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_float, C.c_int, fp, fp, fp, C.c_size_t, fp]),
    "st_simloss_bwd": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_float, C.c_int, fp, fp, fp, fp, fp, C.c_size_t, fp]),
    "st_encode_l0_table_bytes": (C.c_size_t, [C.POINTER(StEncoderWeights), C.c_int]),
    "st_encode_build_l0_table": (C.c_int, [C.POINTER(StEncoderWeights), C.c_int, fp, fp, C.c_size_t, fp]),
    "st_encode_workspace_bytes": (C.c_size_t, [C.POINTER(StEncoderWeights), C.c_int, C.c_int]),
    "st_scorer_train_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "st_scorer_train_step": (C.c_int, [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, fp, C.c_uint64,
                                       fp, fp, fp, fp, C.c_size_t, fp]),
    "st_adam_step": (C.c_int, [fp, fp, fp, fp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fp]),
    "st_ae_train_workspace_bytes": (C.c_size_t, [C.POINTER(StAeParams), C.c_int, C.c_int, C.c_int, C.c_int]),
    "st_ae_train_step": (C.c_int, [C.POINTER(StAeParams), C.POINTER(StAeParams), fp, C.c_int64, fp, fp, fp, C.c_int, C.c_int,
                                   fp, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint64,
                                   fp, fp, C.c_size_t, fp]),
    "st_format_attack_log": (C.c_longlong, [C.c_char_p, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                            C.c_int, C.c_longlong, fp, fp, fp, fp, C.c_int, fp, C.c_int, fp, fp, fp, fp,
                                            C.c_longlong]),
    "st_encode_latent": (C.c_int, [C.POINTER(StEncoderWeights), fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_size_t, fp]),
}

_lib = None


class StError(RuntimeError):
    pass


def lib():
    """Load the HIP library (once).  Raises if it is missing — build it with
    `python -m style_transfer_amd.build` (or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StError("HIP extension %s is missing; run `python -m style_transfer_amd.build`. "
                          "There is no CPU fallback for the product path." % LIB_PATH)
        # torch ships its own libamdhip64; load it FIRST so this library binds to the same HIP runtime
        # instance (streams, device pointers) instead of pulling in a second one from /opt/rocm.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise StError("%s failed (%d): %s" % (what, rc, lib().st_last_error().decode()))
