"""ctypes binding of libdk_hip.so (include/dk_hip.h).

The product path has no CPU fallback: if the HIP extension is missing or an entry point
fails, a ``DkHipError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# DK_HIP_LIB: another build of the same library, for same-box A/B measurements (scripts/ab_lib.sh); the in-tree one otherwise
LIB_PATH = os.environ.get("DK_HIP_LIB") or os.path.join(_HERE, "libdk_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dk_hip.h")


class DkHipError(RuntimeError):
    pass


class dk_gemm_desc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
        ("gate", C.c_void_p), ("res", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
        ("a_seg_len", C.c_int32), ("a_seg_stride", C.c_int32),
        ("c_seg_len", C.c_int32), ("c_seg_stride", C.c_int32),
        ("r_seg_len", C.c_int32), ("r_seg_stride", C.c_int32),
        ("gate_seg_len", C.c_int32), ("gate_stride", C.c_int32),
        ("alpha", C.c_float), ("epilogue", C.c_int32), ("ldw", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class dk_conv_desc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("bias", C.c_void_p),
        ("res", C.c_void_p), ("zeros", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("O", C.c_int32),
        ("ldy", C.c_int32), ("ldr", C.c_int32), ("upsample", C.c_int32), ("epilogue", C.c_int32),
    ]


class dk_conv_gn_desc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("res", C.c_void_p),
        ("gn_scale_shift", C.c_void_p), ("gn_silu", C.c_int32),
        ("x2", C.c_void_p), ("bias2", C.c_void_p), ("stats_partial", C.c_void_p),
        ("image_f32", C.c_void_p), ("image_u8", C.c_void_p), ("raw_bf16", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("O", C.c_int32), ("C2", C.c_int32),
        ("ldw", C.c_int32), ("ldy", C.c_int32), ("ldr", C.c_int32), ("upsample", C.c_int32), ("stats_groups", C.c_int32),
    ]


class dk_mmdit_config(C.Structure):
    _fields_ = [
        ("num_heads", C.c_int32), ("depth_multimodal", C.c_int32), ("depth_unified", C.c_int32),
        ("hidden_size", C.c_int32), ("mlp_ratio", C.c_int32),
        ("vae_latent_dim", C.c_int32), ("patch_size", C.c_int32), ("patchify_via_reshape", C.c_int32),
        ("use_qk_norm", C.c_int32), ("use_rope", C.c_int32), ("rope_axes_dim", C.c_int32 * 4),
        ("n_rope_axes", C.c_int32), ("rope_theta", C.c_int32),
        ("use_pos_embed", C.c_int32), ("max_latent_resolution", C.c_int32),
        ("pooled_text_embed_dim", C.c_int32), ("token_level_text_embed_dim", C.c_int32),
        ("frequency_embed_dim", C.c_int32), ("max_period", C.c_int32),
        ("embed_dtype", C.c_int32), ("layer_norm_eps", C.c_float),
        ("guidance_embed", C.c_int32), ("fp8_linears", C.c_int32), ("fp8_bf16_double_blocks", C.c_int32),
    ]


class dk_gemm_fp8_desc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("A_scales", C.c_void_p), ("W", C.c_void_p), ("w_scale", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("gate", C.c_void_p), ("res", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
        ("a_seg_len", C.c_int32), ("a_seg_stride", C.c_int32), ("a_row0", C.c_int32), ("a_rows", C.c_int32),
        ("c_seg_len", C.c_int32), ("c_seg_stride", C.c_int32),
        ("r_seg_len", C.c_int32), ("r_seg_stride", C.c_int32),
        ("gate_seg_len", C.c_int32), ("gate_stride", C.c_int32),
        ("epilogue", C.c_int32), ("c_mx8", C.c_int32),
        ("C_scales", C.c_void_p), ("c_rows", C.c_int32), ("c_row0", C.c_int32), ("c_col0", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class dk_vae_config(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("block_out_channels", C.c_int32 * 4),
        ("n_blocks", C.c_int32), ("layers_per_block", C.c_int32), ("resnet_groups", C.c_int32),
        ("group_norm_eps", C.c_float),
    ]


DK_EPI_BIAS, DK_EPI_BIAS_GELU, DK_EPI_GATE_RES, DK_EPI_RES, DK_EPI_BIAS_SILU = range(5)

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/dk_hip.h declares
class dk_gemm_plan_t(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kernel", "tile_rows", "tiles", "workgroups", "split_tiles", "k_pieces", "ks", "n_cu", "launches")]


SIGNATURES = {
    "dk_abi_version": (_i32, []),
    "dk_last_error": (C.c_char_p, []),
    "dk_gemm_bf16": (_i32, [C.POINTER(dk_gemm_desc), _vp]),
    "dk_gemm_plan": (_i32, [C.POINTER(dk_gemm_desc), C.POINTER(dk_gemm_desc), C.POINTER(dk_gemm_plan_t)]),
    "dk_gemm_workspace_bytes": (C.c_size_t, []),
    "dk_attention_workspace_bytes": (C.c_size_t, []),
    "dk_attention_set_workspace": (C.c_int, [C.c_void_p, C.c_size_t]),
    "dk_conv3x3_bf16": (_i32, [C.POINTER(dk_conv_desc), _vp]),
    "dk_conv3x3_gn_bf16": (_i32, [C.POINTER(dk_conv_gn_desc), _vp]),
    "dk_groupnorm_table_bf16": (_i32, [_vp, _i32, C.c_int64, _i32, _i32, _vp, _vp, C.c_float, _vp, _i32, _vp, _vp]),
    "dk_attention_bf16": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "dk_attention_d512_tp": (_i32, [_i32]),
    "dk_attention_d512_bf16": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.c_float, _vp, _vp]),
    "dk_attention_bias_bf16": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _i64, _i32, _vp]),
    "dk_embedding_bf16": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "dk_layernorm_bf16": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _f32, _vp]),
    "dk_t5_rmsnorm_bf16": (_i32, [_vp, _vp, _i32, _i32, _vp, _f32, _vp]),
    "dk_text_elementwise": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "dk_t5_bias_bf16": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "dk_gemm_fp8": (_i32, [C.POINTER(dk_gemm_fp8_desc), _vp]),
    "dk_mx_scale_bytes": (_sz, [_i64, _i32]),
    "dk_quantize_mx8": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _i64, _i32, _i32, _vp]),
    "dk_ln_modulate_mx8": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _f32, _vp, _i32, _vp, _i64, _i32, _vp]),
    "dk_weight_pitch_fp8": (_i32, [_i32]),
    "dk_mmdit_set_guidance": (_i32, [_vp, _f32]),
    "dk_ln_modulate_bf16": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "dk_qk_norm_rope_bf16": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _f32, _vp, _i32, _i32, _i32, _vp]),
    "dk_rope_table_f32": (_i32, [_vp, _i32, _i32, _i32, C.POINTER(_i32), _i32, _f32, _vp]),
    "dk_timestep_embedding_bf16": (_i32, [_vp, _i32, _i32, _f32, _i32, _vp, _vp]),
    "dk_latent_to_tokens": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dk_euler_cfg_step": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp]),
    "dk_affine_f32": (_i32, [_vp, _vp, _i64, _f32, _f32, _vp]),
    "dk_groupnorm_scratch_floats": (_sz, [_i32, _i32]),
    "dk_groupnorm_bf16": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _f32, _i32, _vp, _vp]),
    "dk_softmax_rows_bf16": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "dk_transpose_bf16": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "dk_mmdit_create": (_i32, [C.POINTER(dk_mmdit_config), C.POINTER(_vp)]),
    "dk_mmdit_destroy": (None, [_vp]),
    "dk_mmdit_bind": (_i32, [_vp, C.c_char_p, _vp]),
    "dk_mmdit_mod_rows": (_i32, [_vp]),
    "dk_mmdit_mod_offset": (_i32, [_vp, _i32, _i32]),
    "dk_mmdit_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32, _i32, _i32]),
    "dk_mmdit_prepare": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "dk_mmdit_cache_modulation_params": (_i32, [_vp, _vp, _fp, _i32, _vp]),
    "dk_mmdit_forward": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "dk_mmdit_cache_context": (_i32, [_vp, _vp, _vp]),
    "dk_mmdit_run_blocks": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "dk_mmdit_debug_buffer": (_vp, [_vp, _i32]),
    "dk_vae_create": (_i32, [C.POINTER(dk_vae_config), C.POINTER(_vp)]),
    "dk_vae_destroy": (None, [_vp]),
    "dk_vae_bind": (_i32, [_vp, C.c_char_p, _vp]),
    "dk_vae_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "dk_vae_decode": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dk_vae_encoder_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "dk_vae_encode": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "dk_latent_sample_f32": (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _vp]),
    "dk_profile_enable": (_i32, [_i32]),
    "dk_profile_read": (_i32, [_i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dk_tune_set": (_i32, [C.c_char_p, _i32]),
    "dk_weight_pitch": (_i32, [_i32]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libdk_hip.so, failing loudly if it has not been built
    (``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C diffusionkit_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DkHipError(
            f"{LIB_PATH} not found: the HIP extension is not built and there is no CPU fallback. "
            "Run `make -C diffusionkit_amd/csrc` (needs hipcc, cross-compiles for gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise DkHipError(f"{LIB_PATH} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if lib.dk_abi_version() != 5:
        raise DkHipError("libdk_hip.so ABI version mismatch")
    _lib = lib
    return lib


_attn_ws = {}        # (host thread, device index) -> buffer (kept alive while the library may point at it)
_attn_ws_device = {}  # host thread -> device index whose buffer the library's thread-local pointer holds right now


def ensure_attention_workspace(device) -> None:
    """Hand the library its attention workspace for the calling host thread (dk_attention_set_workspace: the partial results of the
    key-split workgroups of csrc/attention5.hip).  The library keeps ONE pointer per host thread, so the pointer is installed again whenever
    the thread moves to another device (cuda:0 -> cuda:1 -> cuda:0 must not leave device-1 memory behind device-0 launches).
    Engines carry their own region (dk_mmdit_* install it for the duration of a call); this one serves the stand-alone ops.
    Optional for correctness: without it no launch is split."""
    import threading

    import torch
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    tid = threading.get_ident()
    if _attn_ws_device.get(tid) == index:
        return
    lib = load()
    n = int(lib.dk_attention_workspace_bytes())
    buf = _attn_ws.get((tid, index))
    if buf is None:
        buf = _attn_ws[(tid, index)] = torch.empty(max(n, 256), dtype=torch.uint8, device=torch.device("cuda", index))
    check(lib.dk_attention_set_workspace(buf.data_ptr(), n), "dk_attention_set_workspace")
    _attn_ws_device[tid] = index


def forget_attention_workspace() -> None:
    """The calling thread's pointer was overwritten from outside (ops.attention(workspace=...), lab): the next ensure_attention_workspace
    installs the regular buffer again.  Other threads' entries stay -- their native thread-locals still point at their buffers."""
    import threading
    _attn_ws_device.pop(threading.get_ident(), None)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().dk_last_error().decode("utf-8", "replace")
        raise DkHipError(f"{what or 'dk call'} failed (rc={rc}): {msg}")
