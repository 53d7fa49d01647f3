"""Host-side checks of the fp8 path and the round-2 host logic (no GPU): the oracle's quantisers (known answers, idempotence),
the weight packer's fp8 layout, the dtype-agnostic weight blob, the scale side-array layout helper against the library's size
function, conditioning tiling for seed lists, and the guidance-embedding restatement."""
from dataclasses import replace

import pytest
import torch

from diffusionkit_amd import _lib
from diffusionkit_amd.config import FLUX_DEV, FLUX_SCHNELL, MODEL_CONFIG, tiny_flux
from diffusionkit_amd.weights import (blob_pack, blob_unpack, dequantize_weight_e4m3, pack_mmdit, quantize_weight_e4m3,
                                      synth_mmdit_weights)
from oracle import fp8 as o8
from oracle.mmdit import OracleMMDiT, Prec
from tests import _fp8 as f8


def test_mx8_scale_exponent_known_answers():
    # smallest power of two s with amax / s <= 448: 448 -> 2^0, 449 -> 2^1, 224 -> 2^-1, 1.75 -> 2^-8, 0 -> clamp 2^-126
    amax = torch.tensor([448.0, 449.0, 896.0, 224.0, 225.0, 1.75, 1.0, 0.0, 3.0e38])
    e = o8.mx8_scale_exponent(amax)
    assert e.tolist() == [127, 128, 128, 126, 127, 119, 119, 1, 127 + 120]


def test_mx8_fake_quant_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 256, generator=g) * torch.logspace(-3, 3, 64)[:, None]
    y = o8.mx8_fake_quant(x)
    assert torch.equal(o8.mx8_fake_quant(y), y)  # idempotent: quantised values are representable with the same scales
    # e4m3 has 3 mantissa bits: relative error of an element against its block maximum is bounded by 2^-4 * (2 / 1.75) ... use 1/8
    xb, yb = x.reshape(64, 8, 32), y.reshape(64, 8, 32)
    assert float(((xb - yb).abs() / xb.abs().amax(-1, keepdim=True)).max()) <= 1.0 / 8
    q, e = o8.mx8_encode(x)
    assert torch.equal(f8.mx8_decode(q, e), y)
    assert torch.equal(o8.mx8_fake_quant(torch.zeros(2, 64)), torch.zeros(2, 64))


def test_weight_quantiser_matches_oracle_restatement():
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(96, 320, generator=g) * 0.02).to(torch.bfloat16)
    w[5] = 0
    q, s = quantize_weight_e4m3(w)
    assert q.dtype == torch.uint8 and s.dtype == torch.float32 and float(s[5]) == 1.0
    assert torch.equal(dequantize_weight_e4m3(q, s), o8.fake_quant_weight(w))
    assert float((dequantize_weight_e4m3(q, s) - w.float()).abs().max()) <= float(w.float().abs().max()) / 16 + 1e-9


def test_pack_mmdit_fp8_layout():
    cfg = replace(tiny_flux(depth_multimodal=1, depth_unified=1, heads=2), weight_dtype="fp8_e4m3")
    named = synth_mmdit_weights(cfg, seed=3)
    p = pack_mmdit(cfg, named, "cpu")
    h, lib = cfg.hidden_size, _lib.load()
    b = "multimodal_transformer_blocks.0.image_transformer_block"
    assert p[b + ".attn.qkv.weight_fp8"].shape == (3 * h, h) and p[b + ".attn.qkv.weight_fp8"].dtype == torch.uint8
    assert p[b + ".attn.qkv.wscale"].shape == (3 * h,) and p[b + ".attn.qkv.wscale"].dtype == torch.float32
    assert b + ".attn.qkv.weight" not in p and p[b + ".attn.qkv.bias"].dtype == torch.bfloat16
    u = "unified_transformer_blocks.0.transformer_block"
    assert p[u + ".linear1.weight_fp8"].shape == (7 * h, h)
    assert p[u + ".linear2.weight_fp8"].shape == (h, lib.dk_weight_pitch_fp8(5 * h))
    assert p["adaLN.weight"].dtype == torch.bfloat16 and p["context_embedder.weight"].dtype == torch.bfloat16  # outside the blocks: bf16
    # the values the engine multiplies with are the oracle's fake-quantised weights
    fq = o8.fake_quant_block_weights(cfg, named)
    got = dequantize_weight_e4m3(p[u + ".linear2.weight_fp8"], p[u + ".linear2.wscale"], k=5 * h)
    assert torch.equal(got[:, :h], fq[u + ".attn.o_proj.weight"]) and torch.equal(got[:, h:], fq[u + ".mlp.fc2.weight"])
    got = dequantize_weight_e4m3(p[b + ".attn.qkv.weight_fp8"], p[b + ".attn.qkv.wscale"])
    assert torch.equal(got[h:2 * h], fq[b + ".attn.k_proj.weight"])
    # mixed-dtype blob round trip (the RCCL weight broadcast)
    blob, index = blob_pack(p)
    assert blob.dtype == torch.uint8
    back = blob_unpack(blob, index)
    assert set(back) == set(p) and all(back[k].dtype == p[k].dtype and torch.equal(back[k], p[k]) for k in p)


def test_scale_layout_helper_matches_library_size():
    lib = _lib.load()
    for rows, k in ((128, 128), (4352, 3072), (300, 15360 + 128), (1, 32)):
        e = torch.arange(rows * ((k + 31) // 32), dtype=torch.int64).remainder(251).to(torch.uint8).reshape(rows, -1)
        arr = f8.scales_to_array(e, rows)
        assert arr.numel() == lib.dk_mx_scale_bytes(rows, k)
        assert torch.equal(f8.array_to_scales(arr, rows, e.shape[1] * 32, rows=rows), e)
    # distinct (row, block) pairs never share a byte
    r = torch.arange(512)[:, None].expand(512, 24)
    kb = torch.arange(24)[None, :].expand(512, 24)
    idx = f8.scale_index(r, kb, f8.n_blk128(512)).reshape(-1)
    assert idx.unique().numel() == idx.numel()
    assert lib.dk_weight_pitch_fp8(3072) == 3072 and lib.dk_weight_pitch_fp8(12288) == 12288 + 128


def test_tile_conditioning_layouts():
    from diffusionkit_amd.pipeline import _tile_conditioning
    c = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)
    p = torch.arange(2 * 5, dtype=torch.float32).reshape(2, 5)
    # SD3 encode_text rows [prompt, negative], 3 seeds, CFG on -> [prompt x 3, negative x 3]
    cc, pp = _tile_conditioning(c, p, 3, True, False)
    assert cc.shape == (6, 3, 4) and torch.equal(cc[:3], c[0:1].expand(3, 3, 4)) and torch.equal(cc[3:], c[1:2].expand(3, 3, 4))
    assert torch.equal(pp[2], p[0]) and torch.equal(pp[5], p[1])
    # SD3, CFG off, 2 seeds: the prompt row for BOTH images (never the negative row)
    cc, pp = _tile_conditioning(c, p, 2, False, False)
    assert cc.shape == (2, 3, 4) and torch.equal(cc[0], c[0]) and torch.equal(cc[1], c[0]) and torch.equal(pp[1], p[0])
    # FLUX: one row tiled; rows already laid out per image pass through
    cc, pp = _tile_conditioning(c[:1], p[:1], 4, False, True)
    assert cc.shape == (4, 3, 4) and torch.equal(cc[3], c[0])
    cc, pp = _tile_conditioning(c, p, 2, False, True)
    assert torch.equal(cc, c) and torch.equal(pp, p)
    with pytest.raises(ValueError):
        _tile_conditioning(c, p[:1], 2, False, True)


def test_guidance_embedding_oracle_and_default_off():
    assert MODEL_CONFIG["argmaxinc/mlx-FLUX.1-dev"] is FLUX_SCHNELL and not FLUX_SCHNELL.guidance_embed and FLUX_DEV.guidance_embed  # quirk Q7 stays the default
    cfg = replace(tiny_flux(1, 1), guidance_embed=True)
    w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=5).items()}
    assert w["guidance_in.mlp.layers.0.weight"].shape == (cfg.hidden_size, cfg.frequency_embed_dim)
    pooled = torch.randn(1, cfg.pooled_text_embed_dim)
    tabs = []
    for g in (1.0, 3.5):
        m = OracleMMDiT(cfg, w, Prec(), guidance=g)
        m.cache_modulation_params(pooled, torch.tensor([500.0]))
        tabs.append(m._mod["final_layer"][500.0])
    assert not torch.allclose(tabs[0], tabs[1])
    # without the flag the guidance value is ignored and the extra weights are not even needed
    m = OracleMMDiT(replace(cfg, guidance_embed=False), w, Prec(), guidance=3.5)
    m.cache_modulation_params(pooled, torch.tensor([500.0]))
    m2 = OracleMMDiT(replace(cfg, guidance_embed=False), w, Prec())
    m2.cache_modulation_params(pooled, torch.tensor([500.0]))
    assert torch.equal(m._mod["final_layer"][500.0], m2._mod["final_layer"][500.0])
