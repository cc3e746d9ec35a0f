"""ctypes binding of libamt_hip.so (include/amt.h, include/amt_ops.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every entry
point raises.  Build it with ``python __graft_entry__.py`` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libamt_hip.so")
POST_LIB_PATH = os.path.join(_HERE, "libamt_post.so")

AMT_F32, AMT_BF16 = 0, 1
AMT_MAX_STAGES = 8


class AmtConfig(C.Structure):
    _fields_ = [
        ("n_stages", C.c_int32),
        ("dims", C.c_int32 * AMT_MAX_STAGES),
        ("depths", C.c_int32 * AMT_MAX_STAGES),
        ("n_layer_pairs", C.c_int32),
        ("n_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("kv_compress", C.c_int32),
        ("ffn_inter", C.c_int32),
        ("n_events", C.c_int32),
        ("local_window", C.c_int32),
        ("dropout_rate", C.c_float),
        ("act_dtype", C.c_int32),
    ]


class AmtSegment(C.Structure):
    _fields_ = [
        ("name", C.c_char * 96),
        ("offset", C.c_uint64),
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("lr_depth", C.c_int32),
    ]


class AmtProfEntry(C.Structure):
    _fields_ = [
        ("name", C.c_char * 64),
        ("launches", C.c_int32),
        ("total_ms", C.c_double),
        ("flops", C.c_double),
        ("bytes", C.c_double),
    ]


def profile_begin():
    load().amt_profile_begin()


def profile_end(max_entries: int = 128):
    """-> list of dict(name, launches, total_ms, flops, bytes), one per kernel class"""
    buf = (AmtProfEntry * max_entries)()
    n = load().amt_profile_end(C.cast(buf, C.c_void_p), max_entries)
    return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), total_ms=float(buf[i].total_ms),
                 flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(n)]


_lib = None
_post = None

_vp = C.c_void_p
_i32 = C.c_int32
_u32 = C.c_uint32
_u64 = C.c_uint64
_f32 = C.c_float

# name -> (restype, argtypes); every symbol declared in include/amt.h and include/amt_ops.h
SIGNATURES = {
    "amt_default_config": (None, [C.POINTER(AmtConfig)]),
    "amt_param_count": (_u64, [C.POINTER(AmtConfig)]),
    "amt_layout_count": (_i32, [C.POINTER(AmtConfig)]),
    "amt_layout_segment": (_i32, [C.POINTER(AmtConfig), _i32, C.POINTER(AmtSegment)]),
    "amt_create": (_i32, [C.POINTER(AmtConfig), _i32, C.POINTER(_vp)]),
    "amt_destroy": (None, [_vp]),
    "amt_last_error": (C.c_char_p, [_vp]),
    "amt_load_weights": (_i32, [_vp, _vp, _u64, _i32, _vp]),
    "amt_weights_ptr": (_vp, [_vp]),
    "amt_refresh_weights": (_i32, [_vp, _vp]),
    "amt_set_rope": (_i32, [_vp, _vp, _vp, _i32]),
    "amt_set_attn_fp8": (_i32, [_vp, _i32]),
    "amt_num_frames": (_i32, [C.POINTER(AmtConfig), _i32]),
    "amt_forward": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "amt_forward_train": (_i32, [_vp, _vp, _i32, _i32, _u64, _vp, _vp, _vp]),
    "amt_forward_backward": (_i32, [_vp, _vp, _vp, _i32, _i32, _f32, _u64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "amt_forward_backward_phase": (_i32, [_vp, _vp, _vp, _i32, _i32, _f32, _u64, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "amt_forward_backward_accum": (_i32, [_vp, _vp, _vp, _i32, _i32, _f32, _u64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "amt_grad_bucket_split": (_u64, [_vp]),
    "amt_workspace_bytes": (_u64, [_vp]),
    "amt_plan_generation": (_u64, [_vp]),
    "amt_seed_word": (_vp, [_vp]),
    "amt_advance_seed": (_i32, [_vp, _u64, _vp]),
    "amt_pack_tail": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amt_team_status": (_i32, [_vp]),
    "amt_set_label_dtype": (_i32, [_vp, _i32]),
    "amt_tlayer_active": (_i32, [_vp]),
    "amt_tlayer_stamps": (_i32, [_vp, _vp, _i32]),
    "amt_adamw_step": (_i32, [_vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "amt_adamw_step_gated": (_i32, [_vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp, _vp]),
    "amt_optimizer_init": (_i32, [_vp, _f32]),
    "amt_optimizer_count": (_i32, [_vp]),
    "amt_optimizer_moment_ptr": (_vp, [_vp, _i32]),
    "amt_optimizer_set_count": (_i32, [_vp, _i32]),
    "amt_grad_accumulator_reset": (_i32, [_vp, _vp]),
    "amt_optimizer_status": (_i32, [_vp]),
    "amt_first_half_wait": (_i32, [_vp, _i32]),
    "amt_first_half_sync": (_i32, [_vp, _vp]),
    "amt_first_half_status": (_i32, [_vp]),
    "amt_grad_bucket": (_vp, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "amt_adamw_step_reduced": (_i32, [_vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp, _vp, _vp]),
    "amt_profile_begin": (None, []),
    "amt_profile_end": (_i32, [_vp, _i32]),
    # ops
    "amt_op_gemm_nt": (_i32, [_i32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp,
                              _i32, _vp, _i32, _vp, _u64, _u32, _f32, _vp]),
    "amt_op_gemm_tn_wgrad": (_i32, [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp]),
    "amt_op_layernorm_fwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "amt_op_layernorm_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "amt_op_stem_fwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "amt_op_stem_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "amt_op_dwln_fwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "amt_op_dwln_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "amt_op_dwconv_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "amt_op_dwblock_bwd": (_i32, [_i32] + [_vp] * 15 + [_i32, _i32, _i32, _vp]),
    "amt_op_scale_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "amt_op_attn_fwd": (_i32, [_i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _u64,
                               _u32, _f32, _vp]),
    "amt_op_attn_global_fwd_fp8": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _u64, _u32,
                                          _f32, _vp, _vp]),
    "amt_op_attn_bwd": (_i32, [_i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _u64,
                               _u32, _f32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp]),
    "amt_op_bce_loss": (_i32, [_i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "amt_op_dropout": (_i32, [_i32, _vp, _vp, _u64, _u64, _u32, _f32, _vp]),
    "amt_op_stream_count": (_i32, [_vp]),
    "amt_op_stream_checkpoint": (_i32, [_vp, _i32, C.c_char_p, C.POINTER(_vp), C.POINTER(C.c_int64), C.POINTER(_i32)]),
    "amt_rng_u32": (_u32, [_u64, _u32, _u32]),
    "amt_drop_threshold24": (_u32, [_f32]),
}


def load():
    """Load libamt_hip.so (once).  Raises ImportError loudly if it is missing or broken."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("AMT_HIP_LIB", LIB_PATH)      # (another build of the same sources: tile-size variants for A/B runs, timing builds)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built.  Run `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{path} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def make_config(conf: dict, dtype: str) -> AmtConfig:
    """model_config dict (model.py:20-34) -> amt_config."""
    cfg = AmtConfig()
    dims, depths = list(conf["dims"]), list(conf["depths"])
    if len(dims) != len(depths) or len(dims) > AMT_MAX_STAGES:
        raise ValueError("dims/depths length mismatch")
    cfg.n_stages = len(dims)
    for i, (d, k) in enumerate(zip(dims, depths)):
        cfg.dims[i] = int(d)
        cfg.depths[i] = int(k)
    hidden = dims[-1]
    cfg.n_layer_pairs = int(conf["num_transformer_layers"])
    cfg.n_heads = int(conf["num_transformer_heads"])
    cfg.head_dim = int(conf["attention_size"])
    cfg.kv_compress = int(conf["compressed_attention_kv_size"])
    cfg.ffn_inter = int(hidden * conf["transformer_hidden_expansion"])
    cfg.n_events = int(conf.get("n_events", 90))
    cfg.local_window = 16
    cfg.dropout_rate = float(conf["transformer_dropout_rate"])
    if float(conf.get("cnn_hidden_expansion", 2.0)) != 2.0:
        raise ValueError("cnn_hidden_expansion must be 2.0")
    cfg.act_dtype = {"float32": AMT_F32, "bfloat16": AMT_BF16}[dtype]
    return cfg


def load_post():
    """CPU post-processing library (stitch_probs / extract_events)."""
    global _post
    if _post is not None:
        return _post
    path = os.environ.get("AMT_POST_LIB", POST_LIB_PATH)      # (an instrumented build of the same sources, e.g. AddressSanitizer: tests/test_post_cpu.py)
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: run `python __graft_entry__.py`")
    _post = C.CDLL(path)
    return _post
