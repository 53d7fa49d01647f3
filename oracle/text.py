"""CPU ORACLE (test infrastructure, NOT product code) -- text conditioning in front of the hot path.

PyTorch-CPU restatement of the reference's CLIP text encoders
(python/src/diffusionkit/mlx/clip.py:28-120), its T5 encoder (python/src/diffusionkit/mlx/t5.py:14-243,
316-325) and the conditioning assembly of encode_text (python/src/diffusionkit/mlx/__init__.py:197-251
for SD3, :642-671 for FLUX).  The exact-math mode is pinned against the reference's own clip.py / t5.py / tokenizer.py executed on
the MLX stand-in (tests/test_reference_mlx_golden.py, < 5e-6) and against Hugging Face transformers' CLIPTextModelWithProjection /
T5EncoderModel, the upstream both follow (tests/test_text_oracle.py); MLX's low-precision arithmetic is unpinned (oracle/mmdit.py header).  Weights use the reference's names
(model_io.py:565-646).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

from .mmdit import Prec, gelu_erf, linear

Tensor = torch.Tensor


# ---- CLIP -----------------------------------------------------------------------------------------
def layer_norm_affine(x: Tensor, w: Tensor, b: Tensor, eps: float, P: Prec) -> Tensor:
    """nn.LayerNorm -> mx.fast.layer_norm(x, weight, bias, eps): fp32 statistics, one rounding."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return P.r((x - mu) * torch.rsqrt(var + eps) * w + b)


def quick_gelu(x: Tensor, P: Prec) -> Tensor:
    """nn.gelu_fast_approx = x * sigmoid(1.702 x) (clip.py:11)."""
    return P.r(x * torch.sigmoid(1.702 * x))


def clip_causal_mask(n: int, low_precision: bool) -> Tensor:
    """CLIPTextModel._get_mask (clip.py:83-89): (q < k) * (-6e4 for half types, -1e9 for fp32)."""
    idx = torch.arange(n)
    return (idx[:, None] < idx[None]).to(torch.float32) * (-6e4 if low_precision else -1e9)


def mha(x: Tensor, w: Dict[str, Tensor], prefix: str, heads: int, mask: Tensor, P: Prec) -> Tensor:
    """mlx.nn.MultiHeadAttention as used by clip.py:37-42,48-50: biased q/k/v/out projections, queries scaled by
    1/sqrt(D) before the product, additive mask, softmax, weighted sum."""
    B, N, dm = x.shape
    D = dm // heads
    q = linear(x, w[prefix + ".query_proj.weight"], w[prefix + ".query_proj.bias"], P)
    k = linear(x, w[prefix + ".key_proj.weight"], w[prefix + ".key_proj.bias"], P)
    v = linear(x, w[prefix + ".value_proj.weight"], w[prefix + ".value_proj.bias"], P)
    q, k, v = (t.reshape(B, N, heads, D).transpose(1, 2) for t in (q, k, v))
    s = P.r(P.r(q * (1.0 / math.sqrt(D))) @ k.transpose(-1, -2))
    s = P.r(s + P.r(mask))
    a = P.r(torch.softmax(s, dim=-1))
    y = P.r(a @ v).transpose(1, 2).reshape(B, N, dm)
    return linear(y, w[prefix + ".out_proj.weight"], w[prefix + ".out_proj.bias"], P)


class OracleCLIPText:
    """CLIPTextModel (clip.py:62-120).  Returns (pooled_output, last_hidden_state, hidden_states)."""

    def __init__(self, cfg, weights: Dict[str, Tensor], prec: Optional[Prec] = None):
        self.cfg, self.w, self.P = cfg, weights, prec or Prec()

    def __call__(self, tokens: Tensor):
        c, w, P = self.cfg, self.w, self.P
        B, N = tokens.shape
        eos = tokens.argmax(-1)  # clip.py:94: the EOS token has the largest id of the vocabulary
        x = P.r(w["token_embedding.weight"][tokens] + w["position_embedding.weight"][:N])
        mask = clip_causal_mask(N, P.act is not None)
        act = quick_gelu if c.hidden_act == "quick_gelu" else gelu_erf
        hidden: List[Tensor] = []
        for i in range(c.num_layers):
            p = f"layers.{i}"
            y = layer_norm_affine(x, w[p + ".layer_norm1.weight"], w[p + ".layer_norm1.bias"], 1e-5, P)
            x = P.r(mha(y, w, p + ".attention", c.num_heads, mask, P) + x)
            y = layer_norm_affine(x, w[p + ".layer_norm2.weight"], w[p + ".layer_norm2.bias"], 1e-5, P)
            y = act(linear(y, w[p + ".linear1.weight"], w[p + ".linear1.bias"], P), P)
            x = P.r(linear(y, w[p + ".linear2.weight"], w[p + ".linear2.bias"], P) + x)
            hidden.append(x)
        last = layer_norm_affine(x, w["final_layer_norm.weight"], w["final_layer_norm.bias"], 1e-5, P)
        pooled = last[torch.arange(B), eos]
        if "text_projection.weight" in w:
            pooled = linear(pooled, w["text_projection.weight"], None, P)
        return pooled, last, hidden


# ---- T5 -------------------------------------------------------------------------------------------
def t5_relative_position_bucket(rel: np.ndarray, num_buckets: int = 32, max_distance: int = 128) -> np.ndarray:
    """_relative_position_bucket, bidirectional (t5.py:14-58): float32 log, int16 truncation."""
    nb = num_buckets // 2
    out = (rel > 0).astype(np.int64) * nb
    a = np.abs(rel)
    max_exact = nb // 2
    scale = np.float32((nb - max_exact) / np.log(max_distance / max_exact))
    safe = np.maximum(a, 1).astype(np.float32)  # |rel| = 0 takes the exact branch below
    large = max_exact + (np.log(safe / np.float32(max_exact)) * scale).astype(np.int16)
    large = np.minimum(large, nb - 1)
    return out + np.where(a < max_exact, a, large)


def t5_rms_norm(x: Tensor, w: Tensor, eps: float, P: Prec) -> Tensor:
    """The reference's RMSNorm (t5.py:131-151): x * rsqrt(sum((x / sqrt(h))^2) + eps) cast to x's dtype, times weight.
    x is the fp32 residual stream (t5.py:199-204), so the cast in between changes nothing."""
    h = x.shape[-1]
    n = x * torch.rsqrt(((x * (1.0 / math.sqrt(h))) ** 2).sum(-1, keepdim=True) + eps)
    return w * n


class OracleT5Encoder:
    """SD3T5Encoder (t5.py:316-325) = wte + TransformerEncoder (t5.py:207-243).  The residual stream is fp32 from the
    first layer on (t5.py:199-204); attention inputs are cast to the weight dtype, the gated MLP runs on the fp32
    normalised stream."""

    def __init__(self, cfg, weights: Dict[str, Tensor], prec: Optional[Prec] = None):
        self.cfg, self.w, self.P = cfg, weights, prec or Prec()

    def bias(self, n: int) -> Tensor:
        """RelativePositionBias.__call__ (t5.py:71-88): [H, n, n]."""
        c = self.cfg
        rel = np.arange(n)[None, :] - np.arange(n)[:, None]  # memory - context
        b = t5_relative_position_bucket(rel, c.relative_attention_num_buckets, c.relative_attention_max_distance)
        return self.w["encoder.relative_attention_bias.embeddings.weight"][torch.from_numpy(b)].permute(2, 0, 1)

    def __call__(self, tokens: Tensor) -> Tensor:
        c, w, P = self.cfg, self.w, self.P
        B, N = tokens.shape
        H, D = c.num_heads, c.d_kv
        x = w["wte.weight"][tokens].to(torch.float32)
        bias = self.bias(N)
        for i in range(c.num_layers):
            p = f"encoder.layers.{i}"
            y = P.r(t5_rms_norm(x, w[p + ".ln1.weight"], c.layer_norm_epsilon, P))  # .astype(weight dtype), t5.py:200
            q = linear(y, w[p + ".attention.query_proj.weight"], None, P).reshape(B, N, H, D).transpose(1, 2)
            k = linear(y, w[p + ".attention.key_proj.weight"], None, P).reshape(B, N, H, D).transpose(1, 2)
            v = linear(y, w[p + ".attention.value_proj.weight"], None, P).reshape(B, N, H, D).transpose(1, 2)
            s = P.r(P.r(q @ k.transpose(-1, -2)) + P.r(bias))  # no 1/sqrt(D) in T5 (t5.py:122-124)
            a = P.r(torch.softmax(s.float(), dim=-1))
            y = P.r(a @ v).transpose(1, 2).reshape(B, N, H * D)
            y = linear(y, w[p + ".attention.out_proj.weight"], None, P)
            x = x + y  # fp32 (t5.py:202-203)
            y = t5_rms_norm(x, w[p + ".ln2.weight"], c.layer_norm_epsilon, P)  # fp32
            g = gelu_erf(y @ w[p + ".dense.wi_0.weight"].t(), Prec())  # nn.gelu (exact) on the fp32 stream, t5.py:166-178
            y = (g * (y @ w[p + ".dense.wi_1.weight"].t())) @ w[p + ".dense.wo.weight"].t()
            x = x + y
        return P.r(t5_rms_norm(x, w["encoder.ln.weight"], c.layer_norm_epsilon, P))  # .astype(t), t5.py:243


# ---- encode_text ----------------------------------------------------------------------------------
def pad_tokens(rows: List[List[int]], pad: int) -> Tensor:
    n = max(len(r) for r in rows)
    return torch.tensor([r + [pad] * (n - len(r)) for r in rows], dtype=torch.long)


def sd3_conditioning(clip_l, clip_g, t5, tokens_l: Tensor, tokens_g: Tensor, tokens_t5: Optional[Tensor]):
    """DiffusionPipeline.encode_text (mlx/__init__.py:197-251) after tokenisation."""
    pl, _, hl = clip_l(tokens_l)
    pg, _, hg = clip_g(tokens_g)
    cond = torch.cat([hl[-2], hg[-2]], dim=-1)
    pooled = torch.cat([pl, pg], dim=-1)
    cond = torch.cat([cond, torch.zeros(cond.shape[0], cond.shape[1], 4096 - cond.shape[2])], dim=-1)
    t5c = t5(tokens_t5) if (t5 is not None and tokens_t5 is not None) else torch.zeros_like(cond)
    return torch.cat([cond, t5c], dim=1), pooled


def flux_conditioning(clip_l, t5, tokens_l: Tensor, tokens_t5: Tensor, t5_max_length: int):
    """FluxPipeline.encode_text (mlx/__init__.py:642-671): prompt row only, T5 tokens zero-padded to the maximum length."""
    pooled, _, _ = clip_l(tokens_l[:1])
    padded = torch.zeros(1, t5_max_length, dtype=torch.long)
    padded[:, :tokens_t5.shape[1]] = tokens_t5[:1]
    return t5(padded), pooled
