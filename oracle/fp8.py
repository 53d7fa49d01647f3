"""CPU ORACLE (test infrastructure, NOT product code) -- the number formats of the MI355X build's fp8 path.

No reference counterpart: the reference quantises checkpoints to 4 bits with MLX (mlx/model_io.py:728-734,772-775) and has no
fp8 path; BASELINE.json configs[3] ("FLUX.1-dev, fp8 weights / bf16 activations, CDNA4 fp8 MFMA") names this one.  This file
restates, in plain PyTorch on the host, the two quantisers the HIP path applies (diffusionkit_amd/weights.py:
quantize_weight_e4m3; diffusionkit_amd/csrc/dk_common.h: dk_mx8_quantize8), so that the oracle model can be run on exactly the
values the fp8 GEMM multiplies: parity of the fp8 engine is then stated against "the oracle with fake-quantised weights and
activations" (tight) and against the plain fp32 oracle (loose, documents what fp8 costs).

  weights      OCP e4m3 (float8_e4m3fn), one fp32 scale per output channel = row amax / 448, elements RNE(w / scale)
  activations  MX-fp8: e4m3 elements, one power-of-two (E8M0) scale per row and 32 consecutive columns = the smallest power of
               two s with amax / s <= 448, clamped to [2^-126, 2^127]; elements RNE(x / s) clamped to +-448
"""
from __future__ import annotations

from typing import Dict

import torch

Tensor = torch.Tensor
E4M3_MAX = 448.0


def fake_quant_weight(w: Tensor) -> Tensor:
    """[N, K] -> the fp32 values an e4m3 matrix with per-row scales represents."""
    wf = w.to(torch.float32)
    amax = wf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
    q = (wf / scale[:, None]).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)
    return q * scale[:, None]


def mx8_scale_exponent(amax: Tensor) -> Tensor:
    """Biased E8M0 exponent e (scale = 2^(e - 127)) of a block with maximum magnitude ``amax``: ceil(log2(amax / 448)) + 127
    evaluated on the fp32 bit pattern of amax * (1/448), clamped to [1, 254] (dk_mx8_quantize8)."""
    t = amax.to(torch.float32) * torch.tensor(1.0 / 448.0, dtype=torch.float32)
    e = (t.view(torch.int32) + 0x7FFFFF) >> 23
    return e.clamp(1, 254)


def mx8_fake_quant(x: Tensor) -> Tensor:
    """[..., K] (K a multiple of 32) -> the fp32 values its MX-fp8 encoding represents."""
    shape = x.shape
    xb = x.to(torch.float32).reshape(*shape[:-1], shape[-1] // 32, 32)
    e = mx8_scale_exponent(xb.abs().amax(dim=-1, keepdim=True))
    inv = torch.ldexp(torch.ones_like(xb[..., :1]), 127 - e)
    q = (xb * inv).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * torch.ldexp(torch.ones_like(inv), e - 127)).reshape(shape)


def mx8_encode(x: Tensor):
    """[M, K] -> (e4m3 bytes uint8 [M, K], E8M0 bytes uint8 [M, K / 32]) -- the encoding itself, for op-level tests."""
    M, K = x.shape
    xb = x.to(torch.float32).reshape(M, K // 32, 32)
    e = mx8_scale_exponent(xb.abs().amax(dim=-1, keepdim=True))
    inv = torch.ldexp(torch.ones_like(xb[..., :1]), 127 - e)
    q = (xb * inv).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(M, K), e.reshape(M, K // 32).to(torch.uint8)


def fake_quant_block_weights(cfg, named: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Reference-named weight dict -> fp32 dict in which the Linear matrices of the transformer blocks carry the values of the
    engine's fp8 packing (weights.py: pack_mmdit with weight_dtype = "fp8_e4m3"): per-output-channel scales, so the fused
    [q|k|v](|fc1) matrices quantise row by row like their parts; the single blocks' linear2 = [o_proj | fc2] shares one scale
    per row across both parts."""
    out = {k: v.to(torch.float32) for k, v in named.items()}
    bf = torch.bfloat16
    n_bf = min(int(getattr(cfg, "fp8_bf16_double_blocks", 0)), cfg.depth_multimodal)  # precision policy: these double blocks stay bf16

    def rowwise(prefix, names):
        for n in names:
            key = f"{prefix}.{n}.weight"
            if key in named:
                out[key] = fake_quant_weight(named[key].to(bf))

    for i in range(n_bf, cfg.depth_multimodal):
        for s in ("image_transformer_block", "text_transformer_block"):
            rowwise(f"multimodal_transformer_blocks.{i}.{s}", ("attn.q_proj", "attn.k_proj", "attn.v_proj", "attn.o_proj", "mlp.fc1", "mlp.fc2"))
    for i in range(cfg.depth_unified):
        p = f"unified_transformer_blocks.{i}.transformer_block"
        rowwise(p, ("attn.q_proj", "attn.k_proj", "attn.v_proj", "mlp.fc1"))
        o, f2 = named[p + ".attn.o_proj.weight"].to(bf), named[p + ".mlp.fc2.weight"].to(bf)
        both = fake_quant_weight(torch.cat([o, f2], dim=1))
        out[p + ".attn.o_proj.weight"], out[p + ".mlp.fc2.weight"] = both[:, :o.shape[1]], both[:, o.shape[1]:]
    return out


def policy_act_quant(cfg):
    """activation fake-quantiser for OracleMMDiT(act_quant=...) that follows the engine's precision policy: the Linears of the first
    cfg.fp8_bf16_double_blocks double-stream blocks see un-quantised activations (diffusionkit_amd/config.py: fp8_config)"""
    n_bf = min(int(getattr(cfg, "fp8_bf16_double_blocks", 0)), cfg.depth_multimodal)
    keep = tuple(f"multimodal_transformer_blocks.{i}." for i in range(n_bf))

    def aq(x, site=None):
        if site is not None and keep and site[0].startswith(keep):
            return x
        return mx8_fake_quant(x)
    aq.takes_site = True
    return aq
