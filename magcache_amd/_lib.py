"""ctypes binding of libmagcache_hip.so (include/magcache_hip.h).

There is deliberately no fallback: if the HIP library is missing or cannot be loaded the import
of the product path fails loudly -- a silent CPU/PyTorch path would void every parity and
performance claim made for the engine.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmagcache_hip.so")
# kernel A/B runs (tools/build_variants.py) point this at an alternative build of the same library
LIB_PATH = os.environ.get("MAGCACHE_HIP_LIB", LIB_PATH)

MC_OK, MC_EINVAL, MC_ENOMEM, MC_EHIP, MC_ESTATE = 0, 1, 2, 3, 4
MC_F32, MC_BF16 = 0, 1
MC_MODE_FULL, MC_MODE_SKIP, MC_MODE_CALIB = 0, 1, 2
# mc_prof_class, in enum order (include/magcache_hip.h)
PROF_CLASSES = ("attn_self", "attn_cross", "gemm_qkv", "gemm_o", "gemm_cross_q", "gemm_cross_o", "gemm_ffn1", "gemm_ffn2",
                "ln_modulate", "rmsnorm_rope", "embed", "head", "other", "sp_wait")
RULE_VARIANTS = {"wan21": 0, "hunyuan": 1, "flux": 2, "wan22_t2v": 3, "wan22_i2v": 4, "wan22_ti2v": 5, "framepack": 6,
                 "omnigen2": 7, "qwen": 8, "eval_wan": 9, "eval_opensora": 10}


class McConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dim", "ffn_dim", "num_heads", "num_layers", "in_dim", "out_dim", "freq_dim",
                                       "text_dim", "text_len", "latent_f", "latent_h", "latent_w")] + \
               [("eps", C.c_float)] + \
               [(n, C.c_int) for n in ("sp_rank", "sp_size", "n_branches", "calibration", "clip_dim", "vace_layers",
                                         "vace_stride", "vace_in_dim", "fp8_linear", "no_context_cache",
                                         "no_token_timesteps", "sp_phases")]


class MagCacheHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"magcache_hip status {status}: {msg}")
        self.status = status


_vp, _i, _l, _f, _d, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t
# mc_sp_gather_fn: int (*)(void* user, int layer, int phase, mc_stream stream)
SP_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p)

# name -> (restype, argtypes); every symbol include/magcache_hip.h declares
SIGNATURES = {
    "mc_last_error": (C.c_char_p, []),
    "mc_version": (C.c_char_p, []),
    "mc_set_option": (_i, [C.c_char_p, _i]),
    "mc_create": (_i, [C.POINTER(McConfig), C.POINTER(_vp)]),
    "mc_create_sized": (_i, [C.POINTER(McConfig), _sz, C.POINTER(_vp)]),
    "mc_destroy": (None, [_vp]),
    "mc_workspace_bytes": (_sz, [_vp]),
    "mc_set_workspace": (_i, [_vp, _vp, _sz]),
    "mc_buffer_info": (_i, [_vp, C.c_char_p, C.POINTER(_sz), C.POINTER(_sz)]),
    "mc_set_weight": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i, _vp]),
    "mc_weights_missing": (_i, [_vp, C.c_char_p, _sz]),
    "mc_forward": (_i, [_vp, _vp, _vp, _d, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mc_set_clip_fea": (_i, [_vp, _vp, _i, _i, _vp]),
    "mc_set_vace_context": (_i, [_vp, _vp, _f, _vp]),
    "mc_set_token_timesteps": (_i, [_vp, _vp, _vp]),
    "mc_set_context": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "mc_use_context": (_i, [_vp, _i]),
    "mc_profile_enable": (_i, [_vp, _i]),
    "mc_profile_read": (_i, [_vp, C.POINTER(_d), C.POINTER(_i)]),
    "mc_profile_read_classes": (_i, [_vp, C.POINTER(_d), C.POINTER(_i)]),
    "mc_embed": (_i, [_vp, _vp, _vp, _d, _vp, _i, _i, _vp]),
    "mc_block_pre_attn": (_i, [_vp, _i, _vp]),
    "mc_block_pre_kv": (_i, [_vp, _i, _vp]),
    "mc_block_pre_q": (_i, [_vp, _i, _vp]),
    "mc_sp_set_chunks": (_i, [_vp, _i]),
    "mc_sp_round_info": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "mc_sp_geometry": (_i, [_vp] + [C.POINTER(_i)] * 6),
    "mc_workspace_base": (_vp, [_vp]),
    "mc_block_attn_local": (_i, [_vp, _i, _vp]),
    "mc_block_attn_round": (_i, [_vp, _i, _i, _vp]),
    "mc_sp_rccl_available": (_i, []),
    "mc_sp_comm_id": (_i, [_vp]),
    "mc_sp_comm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "mc_sp_comm_destroy": (None, [_vp]),
    "mc_sp_comm_info": (C.c_char_p, [_vp]),
    "mc_blocks_sp_rccl": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mc_forward_sp_rccl": (_i, [_vp, _vp, _vp, _vp, _d, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mc_block_post_attn": (_i, [_vp, _i, _i, _i, _vp]),
    "mc_blocks_sp": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mc_vace_block_pre": (_i, [_vp, _i, _vp]),
    "mc_vace_block_post": (_i, [_vp, _i, _i, _i, _vp]),
    "mc_head": (_i, [_vp, _i, _i, _vp]),
    "mc_unpatchify": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mc_calib_ready": (_i, [_vp, _i, C.POINTER(_i)]),
    "mc_calib_finalize": (_i, [_vp, _i, _vp]),
    "mc_state_reset": (_i, [_vp]),
    "mc_import_residual": (_i, [_vp, _i, _vp, _vp]),
    "mc_rule_create": (_vp, [_i, _i, _d, _i, _d, C.POINTER(_d), _i, _i]),
    "mc_rule_destroy": (None, [_vp]),
    "mc_rule_step": (_i, [_vp, C.POINTER(_i)]),
    "mc_rule_cnt": (_i, [_vp]),
    "mc_rule_state": (None, [_vp, C.POINTER(_d), C.POINTER(_i), C.POINTER(_d)]),
    "mc_nearest_interp": (None, [C.POINTER(_d), _i, C.POINTER(_d), _i]),
    "mc_op_gemm_bf16": (_i, [_vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _l,
                             _i, _vp]),
    "mc_op_gemm_bf16_resid_sel": (_i, [_vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp]),
    "mc_op_gemm_bf16_rowsplit": (_i, [_vp, _l, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp]),
    "mc_op_gemm_bf16_gelu_split": (_i, [_vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp]),
    "mc_op_gemm_bf16_kernel": (_i, [_i, _i, _i, _i]),
    "mc_op_gemm_bf16_splitk": (_i, [_i, _i, _i, _i]),
    "mc_op_gemm_splitk_need": (_sz, [_i, _i, _i, _i]),
    "mc_op_set_splitk_workspace": (_i, [_vp, _sz]),
    "mc_op_attention": (_i, [_vp, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _i, _i, _i, _i, _i, _f, _vp]),
    "mc_op_quantize_rows_fp8": (_i, [_vp, _i, _l, _i, _i, _vp, _l, _vp, _vp]),
    "mc_op_gemm_fp8": (_i, [_vp, _l, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp]),
    "mc_op_quantize_rows_mx": (_i, [_vp, _i, _l, _i, _i, _vp, _l, _vp, _l, _vp]),
    "mc_op_gemm_mxfp8": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp]),
    "mc_op_attention_partial": (_i, [_vp, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "mc_op_attn_merge": (_i, [_vp, _vp, _i, _vp, _l, _i, _i, _i, _vp]),
    "mc_op_ln_modulate": (_i, [_vp, _l, _vp, _l, _vp, _vp, _i, _f, _vp, _l, _vp, _l, _i, _i, _vp]),
    "mc_op_rmsnorm_rope": (_i, [_vp, _l, _vp, _f, _vp, _i, _i, _i, _vp]),
    "mc_op_skip_add": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _vp]),
    "mc_op_residual_sub": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _vp]),
    "mc_op_calib_stats": (_i, [_vp, _l, _vp, _l, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "mc_op_cfg_euler": (_i, [_vp, _vp, _f, _f, _vp, _vp, _sz, _vp]),
    "mc_op_cast_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "mc_op_lincomb": (_i, [_vp, _vp, _i, _vp, _sz, _vp]),
    "mc_op_rope_table": (_i, [_i, _i, _i, _i, _i, _vp]),
    # include/magcache_mmdit.h
    "mc_mmdit_create": (_i, [_vp, C.POINTER(_vp)]),
    "mc_mmdit_destroy": (None, [_vp]),
    "mc_mmdit_workspace_bytes": (_sz, [_vp]),
    "mc_mmdit_set_workspace": (_i, [_vp, _vp, _sz]),
    "mc_mmdit_buffer_info": (_i, [_vp, C.c_char_p, C.POINTER(_sz), C.POINTER(_sz)]),
    "mc_mmdit_set_weight": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i, _vp]),
    "mc_mmdit_weights_missing": (_i, [_vp, C.c_char_p, _sz]),
    "mc_mmdit_set_rope": (_i, [_vp, _vp, _vp, _i, _vp]),
    "mc_mmdit_forward": (_i, [_vp, _vp, _d, _d, _vp, _i, _vp, _i, _vp, _vp]),
    "mc_mmdit_begin": (_i, [_vp, _vp, _d, _d, _vp, _i, _vp, _i, _vp]),
    "mc_mmdit_block_pre": (_i, [_vp, _i, _vp]),
    "mc_mmdit_block_attn_local": (_i, [_vp, _i, _vp]),
    "mc_mmdit_block_post": (_i, [_vp, _i, _vp]),
    "mc_mmdit_end": (_i, [_vp, _vp, _vp]),
    "mc_mmdit_unpatchify": (_i, [_vp, _vp, _vp, _vp]),
    "mc_mmdit_calib_stats": (_i, [_vp, C.POINTER(C.c_float), _vp]),
    "mc_mmdit_state_reset": (_i, [_vp]),
}


class McMmditConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("family", "dim", "num_heads", "n_double", "n_single", "in_channels", "out_channels",
                                       "txt_dim", "txt_len", "vec_dim", "img_tokens", "latent_f", "latent_h", "latent_w",
                                       "refiner_depth", "calibration", "sp_rank", "sp_size")]


MC_FAMILY_FLUX, MC_FAMILY_HUNYUAN = 0, 1

_lib = None


def load():
    """Load the shared library (once).  Raises if it is absent -- build it with
    `python -m magcache_amd.build` (hipcc --offload-arch=gfx950)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the HIP engine is not built (python -m magcache_amd.build). "
                          "magcache_amd has no CPU fallback by design.")
    # PyTorch-ROCm ships its own libamdhip64; it must be the HIP runtime of this process BEFORE our
    # library is mapped, otherwise libmagcache_hip.so binds /opt/rocm's copy and the two runtimes
    # do not share devices, streams or allocations ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # MAGCACHE_HIP_OPTIONS="key=value,key=value": process-wide mc_set_option calls (A/B runs of bench.py and the tools)
    for kv in filter(None, os.environ.get("MAGCACHE_HIP_OPTIONS", "").split(",")):
        key, _, val = kv.partition("=")
        check(lib.mc_set_option(key.strip().encode(), int(val)))
    return lib


def check(status):
    if status != MC_OK:
        raise MagCacheHipError(status, load().mc_last_error().decode())
