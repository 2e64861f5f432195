"""ctypes binding of libxfm_hip.so (include/xfm_hip.h).

The product path has no CPU or PyTorch-eager fallback: if the library is missing or was built for another ABI,
loading fails loudly, and every op refuses non-CUDA(HIP) tensors.
"""
import ctypes
import os

# torch FIRST: the PyTorch-ROCm wheel bundles its own HIP runtime; loading libxfm_hip.so before it would bind the
# system libamdhip64 and leave two runtimes in one process ("no ROCm-capable device is detected").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# XFM_HIP_LIB: A/B a differently built library (kernel experiments); the default is the in-tree build.
LIB_PATH = os.environ.get("XFM_HIP_LIB") or os.path.join(_HERE, "libxfm_hip.so")
ABI_VERSION = 9

c_void_p, c_int, c_long, c_float, c_u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint32


class LnFwdArgs(ctypes.Structure):
    _fields_ = [("x32", c_void_p), ("x16", c_void_p), ("h", c_void_p), ("res", c_void_p), ("ls_gamma", c_void_p),
                ("row_scale", c_void_p), ("w", c_void_p), ("b", c_void_p), ("x_out", c_void_p), ("z_out", c_void_p),
                ("y", c_void_p), ("y32", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
                ("rows", c_int), ("rows_per_sample", c_int), ("eps", c_float),
                ("drop_thresh", c_u32), ("drop_scale", c_float), ("seed_lo", c_u32), ("seed_hi", c_u32), ("gelu", c_int),
                ("res32", c_void_p), ("z32_out", c_void_p)]


class LnBwdArgs(ctypes.Structure):
    _fields_ = [("dy1", c_void_p), ("dy2", c_void_p), ("dy32", c_void_p), ("x32", c_void_p), ("x16", c_void_p),
                ("mean", c_void_p), ("rstd", c_void_p), ("w", c_void_p),
                ("dx32", c_void_p), ("dx16", c_void_p), ("dx_accum", c_int),
                ("dh", c_void_p), ("dres", c_void_p), ("dstream", c_void_p),
                ("h", c_void_p), ("ls_gamma", c_void_p), ("row_scale", c_void_p), ("partial", c_void_p),
                ("rows", c_int), ("rows_per_sample", c_int),
                ("drop_thresh", c_u32), ("drop_scale", c_float), ("seed_lo", c_u32), ("seed_hi", c_u32), ("gelu_b", c_void_p),
                ("defer", c_void_p), ("dres32", c_void_p)]


class ReduceItem(ctypes.Structure):   # xfm_reduce_item
    _fields_ = [("partial", c_void_p), ("out", c_void_p * 4), ("nblocks", c_int), ("D", c_int), ("nset", c_int), ("reserved", c_int)]


class CastItem(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("wb", c_void_p), ("wt", c_void_p), ("ldb", c_long), ("ldt", c_long), ("N", c_int), ("K", c_int),
                ("tiles_x", c_int), ("reserved", c_int), ("tile_start", c_long)]


class AttnArgs(ctypes.Structure):
    _fields_ = [("q", c_void_p), ("q_rs", c_long), ("k", c_void_p), ("k_rs", c_long), ("v", c_void_p), ("v_rs", c_long),
                ("o", c_void_p), ("o_rs", c_long), ("lse", c_void_p), ("bias", c_void_p), ("bias_ld", c_long),
                ("key_keep", c_void_p), ("B", c_int), ("H", c_int), ("Sq", c_int), ("Sk", c_int),
                ("scale", c_float), ("causal", c_int),
                ("drop_thresh", c_u32), ("drop_scale", c_float), ("seed_lo", c_u32), ("seed_hi", c_u32),
                ("dout", c_void_p), ("do_rs", c_long), ("dq", c_void_p), ("dq_rs", c_long), ("dk", c_void_p),
                ("dk_rs", c_long), ("dv", c_void_p), ("dv_rs", c_long), ("delta", c_void_p), ("dbias", c_void_p),
                ("stat_ld", c_long), ("bias_t", c_void_p), ("bias_t_ld", c_long), ("kv_index", c_void_p),
                ("grp_start", c_void_p), ("grp_rows", c_void_p), ("n_groups", c_int),
                ("q_start", c_void_p), ("q_len", c_void_p), ("k_start", c_void_p), ("k_len", c_void_p),
                ("bwd_phase", c_int), ("o_lo", c_void_p), ("bias_tiled", c_void_p), ("bias_t_tiled", c_void_p),
                ("dbias_ws", c_void_p)]


class EmbedArgs(ctypes.Structure):
    _fields_ = [("ids", c_void_p), ("word", c_void_p), ("pos", c_void_p), ("type", c_void_p), ("w", c_void_p),
                ("b", c_void_p), ("y", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("pos_ids", c_void_p),
                ("B", c_int), ("T", c_int), ("pad_id", c_int), ("eps", c_float),
                ("drop_thresh", c_u32), ("drop_scale", c_float), ("seed_lo", c_u32), ("seed_hi", c_u32),
                ("dy", c_void_p), ("dword", c_void_p), ("dpos", c_void_p), ("partial", c_void_p), ("pos_mode", c_int),
                ("row_map", c_void_p), ("y32", c_void_p), ("dy32", c_void_p), ("dz_out", c_void_p)]


_P, _L, _I, _F, _U = c_void_p, c_long, c_int, c_float, c_u32


class RLayerParams(ctypes.Structure):
    _fields_ = ([(n, _P) for n in ("wqkv", "wqkv_t", "wo", "wo_t", "wq2", "wq2_t", "wkv2_t", "wo2", "wo2_t", "wi", "wi_t", "wout", "wout_t")]
                + [(n, _L) for n in ("ld_wqkv_t", "ld_wo_t", "ld_wq2_t", "ld_wkv2_t", "ld_wo2_t", "ld_wi_t", "ld_wout_t")]
                + [(n, _P) for n in ("bqkv", "bo", "bq2", "bo2", "bi", "bout", "ln1_w", "ln1_b", "ln2_w", "ln2_b", "ln3_w", "ln3_b",
                                     "dwqkv", "dbqkv", "dwo", "dbo", "dwq2", "dbq2", "dwkv2", "dbkv2", "dwo2", "dbo2", "dwi", "dbi",
                                     "dwout", "dbout", "dln1_w", "dln1_b", "dln2_w", "dln2_b", "dln3_w", "dln3_b")]
                + [("D", _I), ("H", _I), ("FF", _I), ("has_cross", _I), ("eps", _F)])


class RLayerIO(ctypes.Structure):
    _fields_ = [("R", _I), ("B", _I), ("T", _I), ("R_alloc", _I), ("B_alloc", _I), ("Nenc", _I), ("U", _I),
                ("seq_start", _P), ("seq_len", _P), ("key_keep", _P), ("enc_keep", _P), ("grp_start", _P), ("grp_rows", _P),
                ("xq_start", _P), ("xq_len", _P), ("xq_max", _I),
                ("causal", _I), ("zero_fill", _I), ("scale", _F),
                ("att_thresh", _U), ("att_scale", _F), ("hid_thresh", _U), ("hid_scale", _F), ("seed_hi", _U), ("seed_ctr", _U),
                ("x", _P), ("slab", _P), ("kv", _P), ("kv_ld", _L), ("kv_event", _P), ("f32_stream", _I), ("x32", _P)]


class RLayerBwd(ctypes.Structure):
    _fields_ = [("bslab", _P), ("dy_a", _P), ("dy_b", _P), ("enc", _P), ("dkv", _P), ("dkv_ld", _L), ("denc32", _P),
                ("need_dprev", _I), ("side_stream", _P), ("ws_main", _P), ("ws_main_bytes", _L), ("ws_side", _P), ("ws_side_bytes", _L),
                ("ln_ws", _P), ("ln_ws_stride", _L), ("ln_items", _P), ("ln_count", _P), ("dy_b32", _P), ("defer_wgrad", _I)]


class TnItem(ctypes.Structure):   # xfm_tn_item
    _fields_ = [("dY", _P), ("ldy", _L), ("X", _P), ("ldx", _L), ("dW", _P), ("ldw", _L), ("dbias", _P), ("N", _I), ("K", _I)]


class RLayerLayout(ctypes.Structure):
    _fields_ = [(n, _L) for n in ("qkv", "c1", "lse1", "h", "z1", "m1", "r1", "y1", "q2", "c2", "lse2", "z2", "m2", "r2", "y2", "hact", "u",
                                  "z3", "m3", "r3", "y3", "fwd_bytes",
                                  "dh3", "dres3", "du", "d1a", "dh2", "dres2", "dc2", "dq2", "delta2", "d2a", "dh1", "dres1", "dc1", "dqkv",
                                  "delta1", "dprev", "bwd_bytes", "ws_main_bytes", "ws_side_bytes", "c2lo", "y1_32", "y2_32", "y3_32")]


class AdamWArgs(ctypes.Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("group", c_void_p),
                ("lr", c_float * 4), ("wd", c_float * 4),
                ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("bc1", c_float), ("bc2", c_float),
                ("clip_coef", c_void_p), ("n", c_long), ("zero_grad", c_int)]


MIM_SUMS_FLOATS = 3 + 3 * 512   # XFM_MIM_SUMS_FLOATS


# name -> (restype, argtypes); mirrors include/xfm_hip.h one to one
SIGNATURES = {
    "xfm_last_error": (ctypes.c_char_p, []),
    "xfm_abi_version": (c_int, []),
    "xfm_gemm_nt": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long,
                            c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "xfm_gemm_nt_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "xfm_gemm_nt_ksplit_workspace": (c_long, [c_int, c_int, c_int]),
    "xfm_gemm_nt_ksplit": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                   c_long, c_void_p]),
    "xfm_gemm_tn_workspace": (c_long, [c_int, c_int, c_int]),
    "xfm_gemm_tn": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_int,
                            c_void_p, c_long, c_void_p]),
    "xfm_cast_transpose": (c_int, [c_void_p, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p]),
    "xfm_cast_transpose_batch": (c_int, [c_void_p, c_int, c_long, c_void_p]),
    "xfm_colsum_workspace": (c_long, [c_int, c_int]),
    "xfm_colsum": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_long, c_void_p]),
    "xfm_layernorm_fwd": (c_int, [ctypes.POINTER(LnFwdArgs), c_int, c_int, c_void_p]),
    "xfm_reduce_sets_batch": (c_int, [c_int, c_void_p, c_void_p]),
    "xfm_layernorm_bwd_workspace": (c_long, [c_int, c_int, c_int]),
    "xfm_layernorm_bwd": (c_int, [ctypes.POINTER(LnBwdArgs), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_long, c_void_p]),
    "xfm_attn_fwd": (c_int, [ctypes.POINTER(AttnArgs), c_void_p]),
    "xfm_attn_bwd": (c_int, [ctypes.POINTER(AttnArgs), c_void_p]),
    "xfm_attn_bwd_workspace": (c_long, [c_void_p]),
    "xfm_bias_tile": (c_int, [c_void_p, c_int, c_int, c_long, c_float, c_void_p, c_void_p, c_void_p]),
    "xfm_rows_index_sum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p]),
    "xfm_relpos_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p]),
    "xfm_relpos_scatter": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p]),
    "xfm_relpos_grid_grad": (c_int, [c_void_p, c_int, c_int, c_long, c_void_p, c_void_p]),
    "xfm_relpos_scatter_sorted": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_void_p, c_void_p]),
    "xfm_patchify": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xfm_vit_tokens_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xfm_vit_tokens_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xfm_pool_rows_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "xfm_pool_rows_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xfm_mim_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xfm_mim_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xfm_mim_masks": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_float, ctypes.c_uint64, c_void_p, c_void_p, c_void_p]),
    "xfm_embed_ln_fwd": (c_int, [ctypes.POINTER(EmbedArgs), c_int, c_void_p]),
    "xfm_embed_ln_bwd_workspace": (c_long, [c_int, c_int]),
    "xfm_embed_ln_bwd": (c_int, [ctypes.POINTER(EmbedArgs), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_long,
                                 c_void_p]),
    "xfm_ce_fwd": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xfm_ce_bwd": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_long, c_void_p]),
    "xfm_adamw": (c_int, [ctypes.POINTER(AdamWArgs), c_void_p]),
    "xfm_sumsq": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "xfm_dp_unique_id": (c_int, [c_void_p]),
    "xfm_dp_init": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "xfm_dp_bucket_allreduce": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "xfm_dp_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "xfm_dp_broadcast": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "xfm_dp_finalize": (c_int, [c_void_p]),
    "xfm_rlayer_layout": (c_int, [c_int] * 11 + [ctypes.POINTER(RLayerLayout)]),
    "xfm_rlayer_fwd": (c_int, [ctypes.POINTER(RLayerParams), ctypes.POINTER(RLayerIO), c_void_p]),
    "xfm_rlayer_bwd": (c_int, [ctypes.POINTER(RLayerParams), ctypes.POINTER(RLayerIO), ctypes.POINTER(RLayerBwd), c_void_p]),
    "xfm_gemm_tn_group_workspace": (c_long, [c_int, c_void_p, c_int]),
    "xfm_gemm_tn_group": (c_int, [c_int, c_void_p, c_int, c_void_p, c_long, c_void_p]),
    "xfm_gemm_tn_batch_workspace": (c_long, [c_int, c_int, c_int, c_int]),
    "xfm_gemm_tn_batch": (c_int, [c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_void_p, c_long,
                                  c_void_p]),
    "xfm_rownorm_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "xfm_rownorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "xfm_itc_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xfm_itc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p]),
    "xfm_hard_negatives": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xfm_rows_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "xfm_rows_scatter_add": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "xfm_rows_segment_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_void_p, c_void_p]),
}

_lib = None


class XfmHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library.  Fails loudly when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XfmHipError(f"{LIB_PATH} not found: build it with `python -m xfm_amd.build` "
                          f"(there is no CPU / eager fallback for the XFM hot path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise XfmHipError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.xfm_abi_version() != ABI_VERSION:
        raise XfmHipError(f"ABI mismatch: library {lib.xfm_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().xfm_last_error().decode(errors="replace")
        raise XfmHipError(f"{what} failed (code {rc}): {msg}")
