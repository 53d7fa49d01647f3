"""Text conditioning in front of the hot path: CLIP-L / CLIP-G text encoders, the T5-XXL
encoder, the CLIP BPE tokenizer and the conditioning assembly of ``encode_text``.

Host-side mirror of python/src/diffusionkit/mlx/clip.py:28-120, t5.py:60-243,316-325, tokenizer.py:14-118 and
mlx/__init__.py:176-251,642-671.  The encoders run once per prompt: their Linear layers and attention use the MMDiT
kernels through the C ABI (dk_gemm_bf16, dk_attention_bias_bf16), the rest the small kernels of csrc/text_ops.hip.
Build dtype is bf16 (the reference: fp16 weights, and for T5 an fp32 residual stream, kept fp32 here as well; its
gated MLP runs on bf16 casts of the fp32 normalised stream instead of fp32 x fp16 products).
"""
from __future__ import annotations

import ctypes as C
import json
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops
from .engine import _ptr, _require_cuda, _stream
from .weights import _Init

Tensor = torch.Tensor
BF = torch.bfloat16


# ---------------------------------------------------------------------------------------------------
# configurations (reference config.py:143-153; T5: google/t5-v1_1-xxl, model_io.py:928)
# ---------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class CLIPTextModelConfig:
    num_layers: int = 23
    model_dims: int = 1024
    num_heads: int = 16
    max_length: int = 77
    vocab_size: int = 49408
    projection_dim: Optional[int] = None
    hidden_act: str = "quick_gelu"


CLIP_L = CLIPTextModelConfig(num_layers=12, model_dims=768, num_heads=12, projection_dim=768, hidden_act="quick_gelu")
CLIP_G = CLIPTextModelConfig(num_layers=32, model_dims=1280, num_heads=20, projection_dim=1280, hidden_act="gelu")


@dataclass(frozen=True)
class T5EncoderConfig:
    vocab_size: int = 32128
    d_model: int = 4096
    d_kv: int = 64
    num_heads: int = 64
    d_ff: int = 10240
    num_layers: int = 24
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    feed_forward_proj: str = "gated-gelu"


T5_XXL = T5EncoderConfig()


def tiny_clip(act: str = "quick_gelu", proj: Optional[int] = 64) -> CLIPTextModelConfig:
    return CLIPTextModelConfig(num_layers=3, model_dims=128, num_heads=2, max_length=77, vocab_size=512, projection_dim=proj,
                               hidden_act=act)


def tiny_t5() -> T5EncoderConfig:
    return T5EncoderConfig(vocab_size=384, d_model=256, d_kv=64, num_heads=4, d_ff=512, num_layers=2)


# ---------------------------------------------------------------------------------------------------
# seeded synthetic weights with the reference's module-tree names (model_io.py:565-646)
# ---------------------------------------------------------------------------------------------------
def synth_clip_weights(cfg: CLIPTextModelConfig, seed: int = 2468, device="cpu", dtype=BF, shapes_only=False) -> Dict[str, Tensor]:
    I = _Init(device, seed, dtype, shapes_only)
    d = cfg.model_dims
    I.normal("token_embedding.weight", (cfg.vocab_size, d))
    I.normal("position_embedding.weight", (cfg.max_length, d))
    for i in range(cfg.num_layers):
        p = f"layers.{i}"
        for ln in ("layer_norm1", "layer_norm2"):
            I.normal(f"{p}.{ln}.weight", (d,), mean=1.0)
            I.normal(f"{p}.{ln}.bias", (d,))
        for n in ("query_proj", "key_proj", "value_proj", "out_proj"):
            I.linear(f"{p}.attention.{n}", d, d)
        I.linear(f"{p}.linear1", 4 * d, d)
        I.linear(f"{p}.linear2", d, 4 * d)
    I.normal("final_layer_norm.weight", (d,), mean=1.0)
    I.normal("final_layer_norm.bias", (d,))
    if cfg.projection_dim is not None:
        I.normal("text_projection.weight", (cfg.projection_dim, d))
    return I.out


def synth_t5_weights(cfg: T5EncoderConfig, seed: int = 1357, device="cpu", dtype=BF, shapes_only=False) -> Dict[str, Tensor]:
    I = _Init(device, seed, dtype, shapes_only)
    d, inner = cfg.d_model, cfg.d_kv * cfg.num_heads
    I.normal("wte.weight", (cfg.vocab_size, d), std=1.0)
    I.normal("encoder.relative_attention_bias.embeddings.weight", (cfg.relative_attention_num_buckets, cfg.num_heads), std=0.5)
    for i in range(cfg.num_layers):
        p = f"encoder.layers.{i}"
        I.normal(f"{p}.ln1.weight", (d,), mean=1.0)
        I.normal(f"{p}.ln2.weight", (d,), mean=1.0)
        I.normal(f"{p}.attention.query_proj.weight", (inner, d), std=0.3 / math.sqrt(d))
        I.normal(f"{p}.attention.key_proj.weight", (inner, d), std=1.0 / math.sqrt(d))
        I.normal(f"{p}.attention.value_proj.weight", (inner, d), std=1.0 / math.sqrt(d))
        I.normal(f"{p}.attention.out_proj.weight", (d, inner), std=1.0 / math.sqrt(inner))
        I.normal(f"{p}.dense.wi_0.weight", (cfg.d_ff, d), std=1.0 / math.sqrt(d))
        I.normal(f"{p}.dense.wi_1.weight", (cfg.d_ff, d), std=1.0 / math.sqrt(d))
        I.normal(f"{p}.dense.wo.weight", (d, cfg.d_ff), std=1.0 / math.sqrt(cfg.d_ff))
    I.normal("encoder.ln.weight", (d,), mean=1.0)
    return I.out


# ---------------------------------------------------------------------------------------------------
# device helpers over the C ABI
# ---------------------------------------------------------------------------------------------------
def _embedding(table: Tensor, ids: Tensor, pos: Optional[Tensor] = None, want_f32: bool = False):
    lib = _lib.load()
    ids = ids.to(torch.int32).contiguous()
    n, dim = ids.numel(), table.shape[1]
    out = torch.empty(n, dim, dtype=BF, device=table.device)
    of = torch.empty(n, dim, dtype=torch.float32, device=table.device) if want_f32 else None
    _lib.check(lib.dk_embedding_bf16(table.data_ptr(), ids.data_ptr(), _ptr(pos), pos.shape[0] if pos is not None else 0, out.data_ptr(),
                                     _ptr(of), n, dim, table.shape[0], _stream()), "dk_embedding_bf16")
    return out, of


def _layernorm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    out = torch.empty_like(x)
    _lib.check(_lib.load().dk_layernorm_bf16(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], w.data_ptr(), b.data_ptr(), eps,
                                             _stream()), "dk_layernorm_bf16")
    return out


def _t5_rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    out = torch.empty(x.shape, dtype=BF, device=x.device)
    _lib.check(_lib.load().dk_t5_rmsnorm_bf16(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], w.data_ptr(), eps, _stream()),
               "dk_t5_rmsnorm_bf16")
    return out


def _elementwise(op: int, a: Tensor, b: Optional[Tensor] = None, r: Optional[Tensor] = None) -> Optional[Tensor]:
    y = torch.empty_like(a) if op != 2 else None
    _lib.check(_lib.load().dk_text_elementwise(a.data_ptr(), _ptr(b), _ptr(y), _ptr(r), a.numel(), op, _stream()), "dk_text_elementwise")
    return y


def attention_bias(qkv: Tensor, H: int, D: int, scale: float, bias: Tensor, per_head: bool) -> Tensor:
    """SDPA over a token-major [B, S, 3*H*D] buffer with an additive score bias [S, ldb] (shared) or [H, S, ldb]."""
    lib = _lib.load()
    B, S, ld = qkv.shape
    h = H * D
    out = torch.empty(B, S, h, dtype=BF, device=qkv.device)
    base = qkv.data_ptr()
    ldb = bias.shape[-1]
    _lib.check(lib.dk_attention_bias_bf16(base, base + 2 * h, base + 4 * h, out.data_ptr(), B, H, S, D, ld, h, scale, bias.data_ptr(),
                                          S * ldb if per_head else 0, ldb, _stream()), "dk_attention_bias_bf16")
    return out


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


# ---------------------------------------------------------------------------------------------------
# engines
# ---------------------------------------------------------------------------------------------------
@dataclass
class CLIPOutput:
    """clip.py:14-25"""
    pooled_output: Optional[Tensor] = None
    last_hidden_state: Optional[Tensor] = None
    hidden_states: Optional[List[Tensor]] = None


class CLIPTextEngine:
    """Drop-in for the reference ``CLIPTextModel`` (clip.py:62-120) on reference-named bf16 device tensors."""

    def __init__(self, config: CLIPTextModelConfig, weights: Dict[str, Tensor], device):
        _lib.load()
        self.config, self.device = config, torch.device(device)
        w = {k: v.to(self.device, BF).contiguous() for k, v in weights.items()}
        self.w = w
        # q | k | v in one GEMM per layer
        self.qkv_w = [torch.cat([w[f"layers.{i}.attention.{n}_proj.weight"] for n in ("query", "key", "value")], 0).contiguous()
                      for i in range(config.num_layers)]
        self.qkv_b = [torch.cat([w[f"layers.{i}.attention.{n}_proj.bias"] for n in ("query", "key", "value")], 0).contiguous()
                      for i in range(config.num_layers)]
        self._masks: Dict[int, Tensor] = {}

    def _mask(self, n: int) -> Tensor:
        """clip.py:83-89 with the half-precision constant (-6e4), columns padded to a multiple of 64."""
        if n not in self._masks:
            idx = torch.arange(n)
            m = torch.zeros(n, _pad64(n))
            m[:, :n] = (idx[:, None] < idx[None]).float() * -6e4
            self._masks[n] = m.to(self.device, BF)
        return self._masks[n]

    def __call__(self, tokens) -> CLIPOutput:
        c, w = self.config, self.w
        tokens = torch.as_tensor(tokens).to(self.device)
        B, N = tokens.shape
        d, H = c.model_dims, c.num_heads
        D = d // H
        eos = tokens.argmax(-1)
        x, _ = _embedding(w["token_embedding.weight"], tokens.reshape(-1), w["position_embedding.weight"][:N].contiguous())
        mask = self._mask(N)
        hidden: List[Tensor] = []
        for i in range(c.num_layers):
            p = f"layers.{i}"
            y = _layernorm(x, w[p + ".layer_norm1.weight"], w[p + ".layer_norm1.bias"])
            qkv = ops.linear(y, self.qkv_w[i], self.qkv_b[i])
            att = attention_bias(qkv.view(B, N, 3 * d), H, D, 1.0 / math.sqrt(D), mask, per_head=False).view(B * N, d)
            x = ops.linear(att, w[p + ".attention.out_proj.weight"], w[p + ".attention.out_proj.bias"], epilogue=ops.DK_EPI_RES, res=x)
            y = _layernorm(x, w[p + ".layer_norm2.weight"], w[p + ".layer_norm2.bias"])
            if c.hidden_act == "quick_gelu":
                y = _elementwise(0, ops.linear(y, w[p + ".linear1.weight"], w[p + ".linear1.bias"]))
            else:
                y = ops.linear(y, w[p + ".linear1.weight"], w[p + ".linear1.bias"], epilogue=ops.DK_EPI_BIAS_GELU)
            x = ops.linear(y, w[p + ".linear2.weight"], w[p + ".linear2.bias"], epilogue=ops.DK_EPI_RES, res=x)
            hidden.append(x.view(B, N, d))
        last = _layernorm(x, w["final_layer_norm.weight"], w["final_layer_norm.bias"]).view(B, N, d)
        pooled = last[torch.arange(B, device=self.device), eos].contiguous()
        if "text_projection.weight" in w:
            pooled = ops.linear(pooled, w["text_projection.weight"], None)
        return CLIPOutput(pooled_output=pooled, last_hidden_state=last, hidden_states=hidden)


def t5_relative_position_bucket(rel: np.ndarray, num_buckets: int, max_distance: int) -> np.ndarray:
    """t5.py:14-58 (bidirectional): host-side integer table, float32 log as in the reference."""
    nb = num_buckets // 2
    out = (rel > 0).astype(np.int64) * nb
    a = np.abs(rel)
    max_exact = nb // 2
    scale = np.float32((nb - max_exact) / np.log(max_distance / max_exact))
    safe = np.maximum(a, 1).astype(np.float32)  # |rel| = 0 takes the exact branch below
    large = max_exact + (np.log(safe / np.float32(max_exact)) * scale).astype(np.int16)
    return out + np.where(a < max_exact, a, np.minimum(large, nb - 1))


class T5EncoderEngine:
    """Drop-in for the reference ``SD3T5Encoder`` (t5.py:316-325): token ids [B, N] -> bf16 [B, N, d_model]."""

    def __init__(self, config: T5EncoderConfig, weights: Dict[str, Tensor], device):
        _lib.load()
        self.config, self.device = config, torch.device(device)
        w = {k: v.to(self.device, BF).contiguous() for k, v in weights.items()}
        self.w = w
        L = config.num_layers
        self.qkv_w = [torch.cat([w[f"encoder.layers.{i}.attention.{n}_proj.weight"] for n in ("query", "key", "value")], 0).contiguous()
                      for i in range(L)]
        self._bias: Dict[int, Tensor] = {}

    def _position_bias(self, n: int) -> Tensor:
        """RelativePositionBias (t5.py:61-88) as a [H, n, pad64(n)] table."""
        if n not in self._bias:
            c = self.config
            rel = np.arange(-(n - 1), n)  # k - q + n - 1 indexes this vector
            b = torch.from_numpy(t5_relative_position_bucket(rel, c.relative_attention_num_buckets, c.relative_attention_max_distance)
                                 .astype(np.int32)).to(self.device)
            out = torch.empty(c.num_heads, n, _pad64(n), dtype=BF, device=self.device)
            _lib.check(_lib.load().dk_t5_bias_bf16(self.w["encoder.relative_attention_bias.embeddings.weight"].data_ptr(), b.data_ptr(),
                                                   c.num_heads, n, _pad64(n), out.data_ptr(), _stream()), "dk_t5_bias_bf16")
            self._bias[n] = out
        return self._bias[n]

    def __call__(self, tokens) -> Tensor:
        c, w = self.config, self.w
        tokens = torch.as_tensor(tokens).to(self.device)
        B, N = tokens.shape
        H, D, d = c.num_heads, c.d_kv, c.d_model
        inner = H * D
        _, x = _embedding(w["wte.weight"], tokens.reshape(-1), None, want_f32=True)  # fp32 residual stream (t5.py:199-204)
        bias = self._position_bias(N)
        for i in range(c.num_layers):
            p = f"encoder.layers.{i}"
            y = _t5_rmsnorm(x, w[p + ".ln1.weight"], c.layer_norm_epsilon)
            qkv = ops.linear(y, self.qkv_w[i], None)
            att = attention_bias(qkv.view(B, N, 3 * inner), H, D, 1.0, bias, per_head=True).view(B * N, inner)  # no 1/sqrt(D) in T5
            _elementwise(2, ops.linear(att, w[p + ".attention.out_proj.weight"], None), r=x)
            y = _t5_rmsnorm(x, w[p + ".ln2.weight"], c.layer_norm_epsilon)
            g = ops.linear(y, w[p + ".dense.wi_0.weight"], None, epilogue=ops.DK_EPI_BIAS_GELU)  # nn.gelu = exact erf (t5.py:166)
            u = ops.linear(y, w[p + ".dense.wi_1.weight"], None)
            _elementwise(2, ops.linear(_elementwise(1, g, u), w[p + ".dense.wo.weight"], None), r=x)
        return _t5_rmsnorm(x, w["encoder.ln.weight"], c.layer_norm_epsilon).view(B, N, d)


# ---------------------------------------------------------------------------------------------------
# tokenizers
# ---------------------------------------------------------------------------------------------------
class Tokenizer:
    """CLIP byte-pair tokenizer with the behaviour of the reference's port (tokenizer.py:14-118): lower-casing,
    whitespace collapse, the CLIP split pattern, greedy lowest-rank merges, truncation to max_length - 2, BOS / EOS."""

    _PATTERN = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""

    def __init__(self, bpe_ranks: Dict[tuple, int], vocab: Dict[str, int], pad_with_eos: bool = False):
        import regex
        self._re = regex
        self.bpe_ranks, self.vocab = bpe_ranks, vocab
        self.pat = regex.compile(self._PATTERN, regex.IGNORECASE)
        self.pad_to_max_length = True
        self.max_length = 77
        self.pad_with_eos = pad_with_eos
        # a literal "<|startoftext|>" / "<|endoftext|>" inside a prompt: the reference caches the STRING as the merge result and then
        # iterates it, i.e. the marker falls apart into single characters (tokenizer.py:29, :103) -- mirrored
        self._cache = {self.bos: list(self.bos), self.eos: list(self.eos)}

    @classmethod
    def from_files(cls, vocab_json: str, merges_txt: str, pad_with_eos: bool = False) -> "Tokenizer":
        """model_io.py:941-962: vocab.json + merges.txt (first line is a header)."""
        with open(vocab_json, encoding="utf-8") as f:
            vocab = json.load(f)
        with open(merges_txt, encoding="utf-8") as f:
            merges = f.read().strip().split("\n")[1:49152 - 256 - 2 + 1]
        ranks = {tuple(m.split()): i for i, m in enumerate(merges)}
        return cls(ranks, vocab, pad_with_eos)

    bos = "<|startoftext|>"
    eos = "<|endoftext|>"

    @property
    def bos_token(self) -> int:
        return self.vocab[self.bos]

    @property
    def eos_token(self) -> int:
        return self.vocab[self.eos]

    def bpe(self, word: str) -> List[str]:
        if word in self._cache:
            return self._cache[word]
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            pairs = set(zip(parts, parts[1:]))
            best = min(pairs, key=lambda pr: self.bpe_ranks.get(pr, float("inf")))
            if best not in self.bpe_ranks:
                break
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and (parts[i], parts[i + 1]) == best:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._cache[word] = parts
        return parts

    def tokenize(self, text, prepend_bos: bool = True, append_eos: bool = True):
        if isinstance(text, list):
            return [self.tokenize(t, prepend_bos, append_eos) for t in text]
        clean = self._re.sub(r"\s+", " ", text.lower())
        ids = [self.vocab[t] for w in self._re.findall(self.pat, clean) for t in self.bpe(w)]
        room = self.max_length - int(prepend_bos) - int(append_eos)
        ids = ids[:room]
        return ([self.bos_token] if prepend_bos else []) + ids + ([self.eos_token] if append_eos else [])


class T5Tokenizer:
    """tokenizer.py:121-160: the sentencepiece tokenizer of google/t5-v1_1-xxl through transformers, from a local directory
    (there is no hub access here)."""

    def __init__(self, local_path: str, max_context_length: int):
        from transformers import AutoTokenizer
        self.max_length = max_context_length
        self._tokenizer = AutoTokenizer.from_pretrained(local_path, legacy=False, model_max_length=max_context_length)
        self.pad_to_max_length = True
        self.pad_with_eos = False

    def tokenize(self, s: str) -> List[int]:
        ids = self._tokenizer(s, return_tensors="np", return_attention_mask=False, max_length=self.max_length, truncation=True)["input_ids"]
        return [int(t) for t in ids[0]]


def tokenize_rows(tokenizer, text: str, negative_text: Optional[str] = None) -> torch.Tensor:
    """DiffusionPipeline._tokenize (mlx/__init__.py:176-195): prompt (padded to max_length when the tokenizer asks for it)
    and the negative prompt ("" when none is given), right-padded to a common length with EOS (CLIP-L) or 0."""
    pad = tokenizer.eos_token if tokenizer.pad_with_eos else 0
    rows = [list(tokenizer.tokenize(text))]
    if tokenizer.pad_to_max_length:
        rows[0].extend([pad] * (tokenizer.max_length - len(rows[0])))
    # the reference turns a missing negative prompt into "" FIRST and then tests for None (:177-178, :187): the second row is
    # always there -- the empty prompt when cfg_weight <= 1 -- which is what the CFG batch of two needs for 0 < cfg_weight <= 1
    rows.append(list(tokenizer.tokenize("" if negative_text is None else negative_text)))
    n = max(len(r) for r in rows)
    return torch.tensor([r + [pad] * (n - len(r)) for r in rows], dtype=torch.long)


# ---------------------------------------------------------------------------------------------------
# encode_text
# ---------------------------------------------------------------------------------------------------
class TextConditioner:
    """The text half of ``encode_text`` (mlx/__init__.py:197-251 SD3, :642-671 FLUX) on the engines above; plug it into a
    pipeline with ``pipe.set_text_encoder(conditioner)``."""

    def __init__(self, clip_l: CLIPTextEngine, tokenizer_l, t5: Optional[T5EncoderEngine] = None, t5_tokenizer=None,
                 clip_g: Optional[CLIPTextEngine] = None, tokenizer_g=None, flux: bool = False, t5_max_length: int = 256):
        self.clip_l, self.tokenizer_l = clip_l, tokenizer_l
        self.clip_g, self.tokenizer_g = clip_g, tokenizer_g
        self.t5, self.t5_tokenizer = t5, t5_tokenizer
        self.flux, self.t5_max_length = flux, t5_max_length

    def __call__(self, text: str, cfg_weight: float = 7.5, negative_text: str = ""):
        neg = negative_text if cfg_weight > 1 else None
        tokens_l = tokenize_rows(self.tokenizer_l, text, neg)
        if self.flux:
            pooled = self.clip_l(tokens_l[:1]).pooled_output  # the negative prompt is ignored (:650)
            tokens_t5 = tokenize_rows(self.t5_tokenizer, text, neg)
            padded = torch.zeros(1, self.t5_max_length, dtype=torch.long)
            padded[:, :tokens_t5.shape[1]] = tokens_t5[:1]
            return self.t5(padded), pooled
        tokens_g = tokenize_rows(self.tokenizer_g, text, neg)
        out_l, out_g = self.clip_l(tokens_l), self.clip_g(tokens_g)
        cond = torch.cat([out_l.hidden_states[-2], out_g.hidden_states[-2]], dim=-1)
        pooled = torch.cat([out_l.pooled_output, out_g.pooled_output], dim=-1)
        cond = torch.cat([cond, torch.zeros(cond.shape[0], cond.shape[1], 4096 - cond.shape[2], dtype=cond.dtype, device=cond.device)], dim=-1)
        if self.t5 is not None:
            t5c = self.t5(tokenize_rows(self.t5_tokenizer, text, neg))
        else:
            t5c = torch.zeros_like(cond)
        return torch.cat([cond, t5c], dim=1), pooled
