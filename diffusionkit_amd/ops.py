"""Operator-level Python wrappers over the C ABI (include/dk_hip.h).

Each function is a thin argument marshaller: tensors must already be on the GPU, bf16
(unless stated) and contiguous; the call is enqueued on the current stream.  They mirror one
MLX op of the reference hot path each and are what the parity tests drive.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib
from ._lib import (DK_EPI_BIAS, DK_EPI_BIAS_GELU, DK_EPI_BIAS_SILU, DK_EPI_GATE_RES, DK_EPI_RES)  # noqa: F401
from .engine import _ptr, _require_cuda, _stream

Tensor = torch.Tensor
BF = torch.bfloat16


def tune(key: str, value: int) -> None:
    """dk_tune_set: kernel-variant knobs for A/B measurements and parity tests (-1 = automatic)."""
    _lib.check(_lib.load().dk_tune_set(key.encode(), int(value)), "dk_tune_set")


_gemm_ws = {}


def gemm_workspace(device) -> Tensor:
    """Zero-initialised scratch of the GEMM's remainder-wave K split (dk_gemm_workspace_bytes), one per device."""
    key = str(device)
    if key not in _gemm_ws:
        _gemm_ws[key] = torch.zeros(_lib.load().dk_gemm_workspace_bytes(), dtype=torch.uint8, device=device)
    return _gemm_ws[key]


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, epilogue: int = DK_EPI_BIAS,
           gate: Optional[Tensor] = None, res: Optional[Tensor] = None, gate_seg_len: int = 0,
           alpha: float = 1.0, out: Optional[Tensor] = None, workspace: Optional[Tensor] = None) -> Tensor:
    """nn.Linear (+ fused epilogue).  x: [M, K]; w: [N, K]; gate: [n_batch, N]; res: [M, N]."""
    lib = _lib.load()
    for n, t in (("x", x), ("w", w)):
        _require_cuda(t, n, BF)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=BF, device=x.device)
    d = _lib.dk_gemm_desc()
    d.A, d.W, d.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias, d.gate, d.res = _ptr(bias), _ptr(gate), _ptr(res)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldc, d.ldr = x.stride(0), out.stride(0), (res.stride(0) if res is not None else 0)
    d.gate_seg_len = gate_seg_len
    d.gate_stride = gate.stride(0) if gate is not None else 0
    d.alpha, d.epilogue = alpha, epilogue
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel()
    _lib.check(lib.dk_gemm_bf16(C.byref(d), _stream()), "dk_gemm_bf16")
    return out


def gemm_desc_call(**kw) -> None:
    """Raw descriptor call (segment mappings etc.); keyword names = dk_gemm_desc fields,
    tensors are converted to pointers."""
    lib = _lib.load()
    d = _lib.dk_gemm_desc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    _lib.check(lib.dk_gemm_bf16(C.byref(d), _stream()), "dk_gemm_bf16")


_zero_pages = {}


def zero_page(device) -> Tensor:
    key = str(device)
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(256, dtype=BF, device=device)
    return _zero_pages[key]


def conv3x3(x: Tensor, w: Tensor, bias: Optional[Tensor], upsample: bool = False, res: Optional[Tensor] = None,
            downsample: bool = False) -> Tensor:
    """nn.Conv2d k3 on NHWC; w: [O,3,3,C] (or flattened [O, 9C]); C multiple of 64.  Default: stride 1, pad 1
    (``upsample``: over the nearest-x2 view of x); ``downsample``: stride 2 over x padded by one zero row /
    column at the bottom / right (vae.py:141-143)."""
    assert not (upsample and downsample)
    lib = _lib.load()
    _require_cuda(x, "x", BF)
    _require_cuda(w, "w", BF)
    B, Hs, Ws, Cc = x.shape
    H, W_ = (Hs * 2, Ws * 2) if upsample else (Hs // 2, Ws // 2) if downsample else (Hs, Ws)
    if downsample:
        assert Hs % 2 == 0 and Ws % 2 == 0, "stride-2 conv needs even input sizes"
    O = w.shape[0]
    ldy = (O + 3) // 4 * 4
    y = torch.empty(B, H, W_, ldy, dtype=BF, device=x.device)
    d = _lib.dk_conv_desc()
    d.x, d.w, d.y, d.bias, d.res = x.data_ptr(), w.data_ptr(), y.data_ptr(), _ptr(bias), _ptr(res)
    d.zeros = zero_page(x.device).data_ptr()
    d.B, d.H, d.W, d.C, d.O = B, H, W_, Cc, O
    d.ldy, d.ldr = ldy, (res.shape[-1] if res is not None else 0)
    d.upsample = 2 if downsample else int(upsample)
    d.epilogue = DK_EPI_RES if res is not None else DK_EPI_BIAS
    _lib.check(lib.dk_conv3x3_bf16(C.byref(d), _stream()), "dk_conv3x3_bf16")
    return y[..., :O]


def attention_d512(q: Tensor, k: Tensor, v: Tensor, scale: Optional[float] = None) -> Tensor:
    """single-head attention over head_dim 512 (VAE mid block, vae.py:28-57), flash-style; q / k / v: [B, T, 512] bf16."""
    lib = _lib.load()
    for n, t in (("q", q), ("k", k), ("v", v)):
        _require_cuda(t, n, BF)
    B, T, Cc = q.shape
    assert Cc == 512, f"attention_d512: head_dim {Cc}, the kernel is built for 512"
    for n, t in (("q", q), ("k", k), ("v", v)):
        assert t.shape == q.shape and t.is_contiguous(), f"attention_d512: {n} must be a dense [B, T, 512] tensor (row pitch 512)"
    out = torch.empty_like(q)
    vt = torch.empty(B * 512 * lib.dk_attention_d512_tp(T), dtype=BF, device=q.device)
    scale = scale if scale is not None else 1.0 / math.sqrt(Cc)
    _lib.check(lib.dk_attention_d512_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, T, Cc, Cc, scale, vt.data_ptr(),
                                          _stream()), "dk_attention_d512_bf16")
    return out


def groupnorm_table(x: Optional[Tensor], gamma: Tensor, beta: Tensor, groups: int, eps: float, partials: Optional[Tensor] = None,
                    shape=None) -> Tensor:
    """(scale | shift) table [B, 2, C] fp32 of nn.GroupNorm over NHWC ``x`` -- or over the tensor whose output-statistics
    ``partials`` [B, n, G, 2] a ``conv3x3_gn`` launch produced (then ``shape`` = (B, H*W, C))."""
    lib = _lib.load()
    if x is not None:
        _require_cuda(x, "x", BF)
        B, HW, Cc = x.shape[0], x.shape[1] * x.shape[2], x.shape[3]
        n_part = 0
        scratch = torch.empty(lib.dk_groupnorm_scratch_floats(B, groups), dtype=torch.float32, device=x.device)
    else:
        B, HW, Cc = shape
        n_part = partials.shape[1]
        scratch = torch.empty(B * max(1024, n_part) * 2 * groups + B * groups * 2, dtype=torch.float32, device=partials.device)
        scratch[:partials.numel()] = partials.reshape(-1)
    ss = torch.empty(B, 2, Cc, dtype=torch.float32, device=gamma.device)
    _lib.check(lib.dk_groupnorm_table_bf16(_ptr(x), B, HW, Cc, groups, gamma.data_ptr(), beta.data_ptr(), eps, scratch.data_ptr(),
                                           n_part, ss.data_ptr(), _stream()), "dk_groupnorm_table_bf16")
    return ss


def conv3x3_gn(x: Tensor, w: Tensor, bias: Tensor, gn_table: Optional[Tensor] = None, silu: bool = True, res: Optional[Tensor] = None,
               x2: Optional[Tensor] = None, bias2: Optional[Tensor] = None, stats_groups: int = 0, upsample: bool = False,
               image: bool = False):
    """norm -> silu -> conv3x3 as one launch (csrc/conv_halo.hip): ``x`` raw NHWC, ``gn_table`` from ``groupnorm_table``;
    ``w`` [O, 9 C (+ C2)] K-major.  Returns y, or (y, partials) with ``stats_groups``, or (image_f32, image_u8, raw) with ``image``."""
    lib = _lib.load()
    _require_cuda(x, "x", BF)
    _require_cuda(w, "w", BF)
    B, Hs, Ws, Cc = x.shape
    H, W_ = (Hs * 2, Ws * 2) if upsample else (Hs, Ws)
    O = w.shape[0]
    w2 = w.reshape(O, -1)
    d = _lib.dk_conv_gn_desc()
    d.x, d.w, d.bias, d.res = x.data_ptr(), w2.data_ptr(), bias.data_ptr(), _ptr(res)
    d.gn_scale_shift, d.gn_silu = _ptr(gn_table), int(silu)
    d.x2, d.bias2 = _ptr(x2), _ptr(bias2)
    d.B, d.H, d.W, d.C, d.O, d.C2 = B, H, W_, Cc, O, (x2.shape[-1] if x2 is not None else 0)
    d.ldw, d.upsample = w2.shape[1], int(upsample)
    out = None
    if image:
        img = torch.empty(B, H, W_, 3, dtype=torch.float32, device=x.device)
        u8 = torch.empty(B, H, W_, 3, dtype=torch.uint8, device=x.device)
        raw = torch.empty(B, H, W_, 4, dtype=BF, device=x.device)
        d.image_f32, d.image_u8, d.raw_bf16 = img.data_ptr(), u8.data_ptr(), raw.data_ptr()
        out = (img, u8, raw)
    else:
        y = torch.empty(B, H, W_, O, dtype=BF, device=x.device)
        d.y, d.ldy, d.ldr = y.data_ptr(), O, (res.shape[-1] if res is not None else 0)
        out = y
        if stats_groups:
            part = torch.empty(B, (H // 16) * (W_ // 16), stats_groups, 2, dtype=torch.float32, device=x.device)
            d.stats_partial, d.stats_groups = part.data_ptr(), stats_groups
            out = (y, part)
    _lib.check(lib.dk_conv3x3_gn_bf16(C.byref(d), _stream()), "dk_conv3x3_gn_bf16")
    return out


def attention(qkv: Tensor, H: int, D: int, scale: Optional[float] = None, workspace: Optional[Tensor] = None) -> Tensor:
    """SDPA over a token-major [B, S, 3*H*D] projection buffer -> [B, S, H*D].  ``workspace``: lab only (trace buffer of attention4.hip's
    DK4_TRACE builds, dk_attention_set_workspace for the duration of the call)."""
    lib = _lib.load()
    _require_cuda(qkv, "qkv", BF)
    if workspace is None:
        _lib.ensure_attention_workspace(qkv.device)
    B, S, ld = qkv.shape
    h = H * D
    out = torch.empty(B, S, h, dtype=BF, device=qkv.device)
    scale = scale if scale is not None else 1.0 / math.sqrt(D)
    base = qkv.data_ptr()
    if workspace is not None:
        _lib.check(lib.dk_attention_set_workspace(workspace.data_ptr(), workspace.numel()), "dk_attention_set_workspace")
    try:
        _lib.check(lib.dk_attention_bf16(base, base + 2 * h, base + 4 * h, out.data_ptr(), B, H, S, D, ld, h, scale, _stream()),
                   "dk_attention_bf16")
    finally:
        if workspace is not None:
            lib.dk_attention_set_workspace(None, 0)
            _lib.forget_attention_workspace()  # (this thread's next call hands the library its regular workspace again)
    return out


def ln_modulate(x: Tensor, shift: Tensor, scale: Tensor, eps: float = 1e-6) -> Tensor:
    """x: [B, S, h]; shift/scale: [B, h] -> [B, S, h]."""
    lib = _lib.load()
    _require_cuda(x, "x", BF)
    B, S, h = x.shape
    out = torch.empty_like(x)
    _lib.check(lib.dk_ln_modulate_bf16(x.data_ptr(), h, out.data_ptr(), h, B * S, h, shift.data_ptr(), scale.data_ptr(),
                                       shift.stride(0), S, B * S, 0, eps, _stream()), "dk_ln_modulate_bf16")
    return out


def qk_norm_rope_(qkv: Tensor, H: int, D: int, qw: Optional[Tensor], kw: Optional[Tensor],
                  rope: Optional[Tensor], pos_off: int = 0, eps: float = 1e-6) -> Tensor:
    """In place on qkv [B, S, 3*H*D]; rope: f32 [S_pos, D/2, 2]."""
    lib = _lib.load()
    _require_cuda(qkv, "qkv", BF)
    B, S, ld = qkv.shape
    _lib.check(lib.dk_qk_norm_rope_bf16(qkv.data_ptr(), ld, 0, H * D, B * S, H, D, _ptr(qw), _ptr(kw), eps, _ptr(rope),
                                        S, S, pos_off, _stream()), "dk_qk_norm_rope_bf16")
    return qkv


def rope_table(S_txt: int, gh: int, gw: int, axes, theta: float, device) -> Tensor:
    lib = _lib.load()
    half = sum(a // 2 for a in axes)
    t = torch.empty(S_txt + gh * gw, half, 2, dtype=torch.float32, device=device)
    arr = (C.c_int32 * len(axes))(*axes)
    _lib.check(lib.dk_rope_table_f32(t.data_ptr(), S_txt, gh, gw, arr, len(axes), float(theta), _stream()), "dk_rope_table_f32")
    return t


def timestep_embedding(t: Tensor, dim: int, max_period: float, embed_dtype: int) -> Tensor:
    lib = _lib.load()
    _require_cuda(t, "t", torch.float32)
    out = torch.empty(t.numel(), dim, dtype=BF, device=t.device)
    _lib.check(lib.dk_timestep_embedding_bf16(t.data_ptr(), t.numel(), dim, float(max_period), embed_dtype, out.data_ptr(),
                                              _stream()), "dk_timestep_embedding_bf16")
    return out


def groupnorm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, silu: bool) -> Tensor:
    """x: NHWC bf16 [B,H,W,C]."""
    lib = _lib.load()
    _require_cuda(x, "x", BF)
    B, H, W_, Cc = x.shape
    y = torch.empty_like(x)
    scratch = torch.empty(lib.dk_groupnorm_scratch_floats(B, groups), dtype=torch.float32, device=x.device)
    _lib.check(lib.dk_groupnorm_bf16(x.data_ptr(), y.data_ptr(), B, H * W_, Cc, groups, gamma.data_ptr(), beta.data_ptr(), eps,
                                     int(silu), scratch.data_ptr(), _stream()), "dk_groupnorm_bf16")
    return y


def softmax_rows_(x: Tensor) -> Tensor:
    """In-place row softmax; ``x`` may be a column slice ``buf[:, :cols]`` of a wider row-major buffer, whose remaining
    columns are then written as zeros (row stride a multiple of 8)."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == BF and x.dim() == 2 and x.stride(1) == 1):
        raise _lib.DkHipError("softmax_rows_: bf16 GPU matrix with unit column stride expected")
    _lib.check(lib.dk_softmax_rows_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), _stream()), "dk_softmax_rows_bf16")
    return x


def transpose(x: Tensor) -> Tensor:
    lib = _lib.load()
    _require_cuda(x, "x", BF)
    y = torch.empty(x.shape[1], x.shape[0], dtype=BF, device=x.device)
    _lib.check(lib.dk_transpose_bf16(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], _stream()), "dk_transpose_bf16")
    return y


# ---- fp8 path (include/dk_hip.h: dk_gemm_fp8 / dk_quantize_mx8 / dk_ln_modulate_mx8) ----------------------------------
def mx_scale_bytes(rows: int, k: int) -> int:
    return int(_lib.load().dk_mx_scale_bytes(rows, k))


def quantize_mx8(x: Tensor, out: Optional[Tensor] = None, scales: Optional[Tensor] = None, out_row0: int = 0, out_col0: int = 0):
    """bf16 [M, h] -> MX-fp8: (e4m3 bytes uint8 [rows, ldo], scale side array uint8).  Fresh buffers hold exactly the M rows."""
    _require_cuda(x, "x", BF)
    M, h = x.shape
    if out is None:
        out = torch.zeros(M, h, dtype=torch.uint8, device=x.device)
    if scales is None:
        scales = torch.zeros(mx_scale_bytes(out.shape[0], out.shape[1]), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().dk_quantize_mx8(x.data_ptr(), x.stride(0), M, h, out.data_ptr(), out.stride(0), scales.data_ptr(),
                                           out.shape[0], out_row0, out_col0, _stream()), "dk_quantize_mx8")
    return out, scales


def ln_modulate_mx8(x: Tensor, shift: Tensor, scale: Tensor, mod_seg_len: int = 0, eps: float = 1e-6):
    """dk_ln_modulate_bf16 with an MX-fp8 output row: (e4m3 bytes [M, h], scale side array)."""
    for n, t in (("x", x), ("shift", shift), ("scale", scale)):
        _require_cuda(t, n, BF)
    M, h = x.shape
    out = torch.zeros(M, h, dtype=torch.uint8, device=x.device)
    scales = torch.zeros(mx_scale_bytes(M, h), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().dk_ln_modulate_mx8(x.data_ptr(), x.stride(0), M, h, shift.data_ptr(), scale.data_ptr(), shift.stride(0),
                                              mod_seg_len, eps, out.data_ptr(), h, scales.data_ptr(), M, 0, _stream()), "dk_ln_modulate_mx8")
    return out, scales


def gemm_fp8(a8: Tensor, a_scales: Tensor, w8: Tensor, w_scale: Tensor, bias: Optional[Tensor] = None, epilogue: int = DK_EPI_BIAS,
             gate: Optional[Tensor] = None, res: Optional[Tensor] = None, gate_seg_len: int = 0, M: Optional[int] = None,
             k: Optional[int] = None, out_mx8: bool = False, workspace: Optional[Tensor] = None):
    """dk_gemm_fp8 over the first M rows / k columns of an MX-fp8 activation buffer a8 [rows, lda] and an e4m3 weight w8 [N, ldw].
    Returns bf16 [M, N], or with ``out_mx8`` (e4m3 bytes [M, N], scale side array).  ``workspace`` (ops.gemm_workspace): lets a launch of at most
    half a round of tiles with a long reduction be cut along K, as the engines do."""
    lib = _lib.load()
    for n, t, dt in (("a8", a8, torch.uint8), ("a_scales", a_scales, torch.uint8), ("w8", w8, torch.uint8), ("w_scale", w_scale, torch.float32)):
        _require_cuda(t, n, dt)
    M = a8.shape[0] if M is None else M
    K = a8.shape[1] if k is None else k
    N = w8.shape[0]
    d = _lib.dk_gemm_fp8_desc()
    d.A, d.A_scales, d.W, d.w_scale = a8.data_ptr(), a_scales.data_ptr(), w8.data_ptr(), w_scale.data_ptr()
    d.bias, d.gate, d.res = _ptr(bias), _ptr(gate), _ptr(res)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldw = a8.stride(0), w8.stride(0)
    d.a_row0, d.a_rows = 0, a8.shape[0]
    d.ldr = res.stride(0) if res is not None else 0
    d.gate_seg_len = gate_seg_len
    d.gate_stride = gate.stride(0) if gate is not None else 0
    d.epilogue = epilogue
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel()
    if out_mx8:
        out = torch.zeros(M, N, dtype=torch.uint8, device=a8.device)
        sc = torch.zeros(mx_scale_bytes(M, N), dtype=torch.uint8, device=a8.device)
        d.C, d.ldc, d.c_mx8, d.C_scales, d.c_rows, d.c_row0, d.c_col0 = out.data_ptr(), N, 1, sc.data_ptr(), M, 0, 0
        _lib.check(lib.dk_gemm_fp8(C.byref(d), _stream()), "dk_gemm_fp8")
        return out, sc
    out = torch.empty(M, N, dtype=BF, device=a8.device)
    d.C, d.ldc = out.data_ptr(), N
    _lib.check(lib.dk_gemm_fp8(C.byref(d), _stream()), "dk_gemm_fp8")
    return out
