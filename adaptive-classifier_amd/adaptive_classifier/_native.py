"""ctypes binding of libacamd.so (the C ABI declared in include/acamd.h).

The library is the product: there is NO CPU or eager-PyTorch fallback behind these wrappers.
If the shared object is missing or a call fails, an exception is raised.
PyTorch is used only for device memory (tensor.data_ptr()) and streams.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("AC_LIBACAMD_PATH") or os.path.join(_HERE, "libacamd.so")   # env override: A/B experiments only
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
_lib = None

c_void_p, c_int, c_int64, c_size_t, c_float, c_uint64 = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float, ctypes.c_uint64)


class NativeError(RuntimeError):
    pass


AC_GEMM_F32, AC_GEMM_BF16X3, AC_GEMM_F16X2 = 0, 1, 2     # include/acamd.h: ac_gemm_set_arith
AC_BERT_LAYERED = 1     # include/acamd.h: ac_bert_encode_cls_opts never takes the one-launch path
AC_BERT_PATH_PACKED, AC_BERT_PATH_PADDED, AC_BERT_PATH_PADDED_MASK = 0, 1, 2     # ac_bert_encode_cls_unpad's *path


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libacamd.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", _CSRC])
    return _LIB_PATH


class ac_head_dims(ctypes.Structure):
    _fields_ = [("D", c_int), ("H1", c_int), ("H2", c_int), ("C", c_int)]


class ac_bert_config(ctypes.Structure):
    _fields_ = [("hidden", c_int), ("layers", c_int), ("heads", c_int), ("intermediate", c_int),
                ("vocab", c_int), ("max_pos", c_int), ("type_vocab", c_int), ("ln_eps", c_float),
                # per-call options, 0 = process default, else value + 1 (include/acamd.h)
                ("gemm_arith_opt", c_int), ("ln_fusion_opt", c_int), ("one_launch_opt", c_int)]


class ac_prune_job(ctypes.Structure):
    _fields_ = [("rows", c_void_p), ("ld", c_int64), ("n_old", c_int), ("n_new", c_int), ("cap", c_int),
                ("reserved", c_int), ("sum", c_void_p), ("alive", c_void_p), ("dist", c_void_p), ("dropped", c_void_p)]


class ac_modernbert_config(ctypes.Structure):
    _fields_ = [("hidden", c_int), ("layers", c_int), ("heads", c_int), ("intermediate", c_int), ("vocab", c_int),
                ("max_pos", c_int), ("global_every", c_int), ("local_window", c_int), ("norm_eps", c_float),
                ("gemm_arith_opt", c_int)]       # per-call option, 0 = process default, else AC_GEMM_* + 1


class ac_modernbert_weights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "tok_emb", "emb_norm_g", "emb_norm_b", "final_norm_g", "final_norm_b",
        "rope_cos_global", "rope_sin_global", "rope_cos_local", "rope_sin_local",
        "attn_norm_g", "attn_norm_b", "wqkv", "wqkv_b", "wo", "wo_b", "mlp_norm_g", "mlp_norm_b",
        "wi", "wi_b", "wo2", "wo2_b", "zero_bias", "wqkv3", "wo3", "wi3", "wo23")] + [("wi_interleaved32", c_int)]


class ac_wordpiece_vocab(ctypes.Structure):
    _fields_ = [("keys", c_void_p), ("offs", c_void_p), ("ids", c_void_p), ("lens", c_void_p), ("blob", c_void_p),
                ("slots", c_int), ("max_piece_bytes", c_int), ("unk_id", c_int), ("cls_id", c_int), ("sep_id", c_int),
                ("pad_id", c_int), ("lower_case", c_int)]


class ac_bert_weights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b",
        "qkv_w", "qkv_b", "ao_w", "ao_b", "ln1_g", "ln1_b",
        "ff1_w", "ff1_b", "ff2_w", "ff2_b", "ln2_g", "ln2_b",
        "qkv_w3", "ao_w3", "ff1_w3", "ff2_w3", "qkv_wh", "ao_wh", "ff1_wh", "ff2_wh")]


# name -> (restype, argtypes); must list every symbol include/acamd.h declares
_SIGNATURES = {
    "ac_last_error": (ctypes.c_char_p, []),
    "ac_version": (c_int, []),
    "ac_device_info": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_size_t)]),
    "ac_device_cus": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "ac_persistent_launches": (c_int, [ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "ac_knn_l2_topk_workspace": (c_int, [c_int64, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "ac_knn_l2_topk": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_int64,
                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ac_knn_l2_topk_x": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ac_knn_store_bytes": (c_int, [c_int64, c_int, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    "ac_knn_prepare_store": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "ac_knn_update_store": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "ac_knn_l2_topk_batch_workspace": (c_int, [c_int64, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "ac_knn_l2_topk_batch": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                     c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ac_knn_set_profile_events": (c_int, [c_void_p, c_void_p]),
    "ac_topk_merge": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ac_topk_merge_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ac_rows_to_class": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "ac_proto_scores": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ac_synth_unit_rows": (c_int, [c_void_p, c_int64, c_int64, c_int, c_uint64, c_int64, c_void_p]),
    "ac_linear_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                              c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "ac_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_void_p, c_int64,
                            c_float, c_void_p, c_int64, c_void_p]),
    "ac_gemm_set_arith": (c_int, [c_int]),
    "ac_gemm_get_arith": (c_int, []),
    "ac_gemm_set_variant": (c_int, [c_int]),
    "ac_gemm_debug_stamps": (c_int, [c_void_p, c_int64]),
    "ac_gemm_set_pipe_table": (c_int, [ctypes.c_char_p]),
    "ac_gemm_set_pipe_table_f16": (c_int, [ctypes.c_char_p]),
    "ac_gemm_set_krot": (c_int, [c_int]),
    "ac_gemm_set_ln_fusion": (c_int, [c_int]),
    "ac_gemm_ln_fusion_launches": (c_int64, []),
    "ac_gemm_qkv_attn_launches": (c_int64, []),
    "ac_set_persistent_kernels": (c_int, [c_int]),
    "ac_clock_stamp": (c_int, [c_void_p, c_void_p]),
    "ac_gemm_occupancy": (c_int, [c_int, c_int, ctypes.POINTER(c_int)]),
    "ac_split_bf16x3": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "ac_split_f16x2": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "ac_linear_f16x2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int,
                                c_int, c_int, c_void_p]),
    "ac_linear_bf16x3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                 c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ac_head_param_count": (c_int64, [ctypes.POINTER(ac_head_dims)]),
    "ac_head_workspace": (c_int, [ctypes.POINTER(ac_head_dims), c_int, ctypes.POINTER(c_size_t)]),
    "ac_head_forward": (c_int, [ctypes.POINTER(ac_head_dims), c_void_p, c_void_p, c_int64, c_int, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "ac_head_fwd_bwd_ce": (c_int, [ctypes.POINTER(ac_head_dims), c_void_p, c_void_p, c_int64, c_void_p,
                                   c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "ac_head_fwd_bwd_loss": (c_int, [ctypes.POINTER(ac_head_dims), c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_int64, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "ac_sigmoid": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "ac_head_train_step": (c_int, [ctypes.POINTER(ac_head_dims), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_float, c_uint64,
                                   c_void_p, c_void_p,
                                   c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "ac_head_train_epoch": (c_int, [ctypes.POINTER(ac_head_dims), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_float,
                                    c_uint64, c_void_p, c_void_p,
                                    c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p,
                                    c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_int), c_void_p]),
    "ac_softmax_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ac_l2_normalize_rows": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "ac_fisher_accumulate": (c_int, [c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "ac_ewc_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "ac_ewc_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float,
                                  c_float, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p,
                                  c_void_p, c_void_p]),
    "ac_memory_add_prune": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ac_blend_topk": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "ac_modernbert_workspace": (c_int, [ctypes.POINTER(ac_modernbert_config), c_int, c_int, ctypes.POINTER(c_size_t)]),
    "ac_modernbert_encode_cls": (c_int, [ctypes.POINTER(ac_modernbert_config), ctypes.POINTER(ac_modernbert_weights),
                                         c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_size_t,
                                         c_void_p]),
    "ac_modernbert_encode_cls_packed": (c_int, [ctypes.POINTER(ac_modernbert_config), ctypes.POINTER(ac_modernbert_weights),
                                                c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64,
                                                c_void_p, c_size_t, c_void_p]),
    "ac_wordpiece_hash": (c_uint64, [c_void_p, c_int, c_int]),
    "ac_wordpiece_encode": (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(ac_wordpiece_vocab), c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "ac_bert_workspace": (c_int, [ctypes.POINTER(ac_bert_config), c_int, c_int, ctypes.POINTER(c_size_t)]),
    "ac_bert_pack": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ac_bert_encode_cls_packed": (c_int, [ctypes.POINTER(ac_bert_config), ctypes.POINTER(ac_bert_weights), c_void_p, c_void_p,
                                          c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_size_t,
                                          c_void_p]),
    "ac_bert_unpad_last_wait_ns": (ctypes.c_longlong, []),
    "ac_predict_post": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ac_host_alloc": (c_int, [c_size_t, ctypes.POINTER(c_void_p)]),
    "ac_host_free": (c_int, [c_void_p]),
    "ac_bert_encode_cls_unpad": (c_int, [ctypes.POINTER(ac_bert_config), ctypes.POINTER(ac_bert_weights), c_void_p, c_void_p,
                                         c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_size_t, c_int,
                                         ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p]),
    "ac_bert_encode_cls_opts": (c_int, [ctypes.POINTER(ac_bert_config), ctypes.POINTER(ac_bert_weights), c_void_p,
                                        c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_size_t, c_int,
                                        ctypes.POINTER(c_int), c_void_p]),
    "ac_bert_one_launch_status": (c_int, [ctypes.POINTER(ac_bert_config), c_int, c_int, c_void_p, c_size_t,
                                          ctypes.POINTER(c_int), c_void_p]),
    "ac_bert_ln_fusion_clear": (c_int, [c_void_p, c_size_t, c_void_p]),
    "ac_bert_ln_fusion_status": (c_int, [ctypes.POINTER(ac_bert_config), c_int, c_int, c_void_p, c_size_t,
                                         ctypes.POINTER(c_int), c_void_p]),
    "ac_bert_encode_cls": (c_int, [ctypes.POINTER(ac_bert_config), ctypes.POINTER(ac_bert_weights), c_void_p,
                                   c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_size_t,
                                   c_void_p]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load libacamd.so once.  Raises NativeError if it is absent -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NativeError(
                f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(make -C {_CSRC}). There is no CPU fallback for the MI355X hot path.")
        L = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(L, name)   # AttributeError => header/library mismatch, fail loudly
            except AttributeError:
                if os.environ.get("AC_LIBACAMD_PATH"):      # an OLDER build under A/B (tools/): symbols it predates stay unbound
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().ac_last_error()
        raise NativeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def require_gpu():
    if not torch.cuda.is_available():
        raise NativeError("no MI355X visible to PyTorch (torch.cuda.is_available() is False); "
                          "the hot path has no CPU fallback")


def ptr(t):
    """Device (or host) address of a tensor as a void*; None -> NULL."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def device_info():
    cu, lds, hbm = c_int(0), c_int(0), c_size_t(0)
    check(lib().ac_device_info(ctypes.byref(cu), ctypes.byref(lds), ctypes.byref(hbm)), "ac_device_info")
    return {"cus": cu.value, "lds_per_block": lds.value, "hbm_bytes": hbm.value}
