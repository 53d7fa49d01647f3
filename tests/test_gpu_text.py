"""Text conditioning on the MI355X: the text-encoder kernels and engines against the CPU oracle
(oracle/text.py, itself pinned against Hugging Face transformers in tests/test_text_oracle.py)."""
import json
import math
import os

import pytest
import torch

from diffusionkit_amd.text import (CLIPTextEngine, T5EncoderEngine, TextConditioner, Tokenizer, attention_bias, synth_clip_weights,
                                   synth_t5_weights, tiny_clip, tiny_t5)
from oracle import text as ot
from oracle.mmdit import Prec
from tests._util import BF, TOL_SINGLE_OP, bf16r, max_abs, randn, rel_l2

pytestmark = pytest.mark.gpu


def g(t, dev):
    return t.to(dev, BF).contiguous()


def yardstick(hip, emu, exact, what):
    e_h, e_e = rel_l2(exact, hip), rel_l2(exact, emu)
    assert e_h <= 2.0 * e_e + 2e-3, f"{what}: hip-vs-fp32 {e_h:.3e} > 2*emu-vs-fp32 {e_e:.3e} + 2e-3"


@pytest.mark.parametrize("B,H,S,D,per_head", [(2, 2, 77, 64, False), (1, 4, 150, 64, True), (2, 3, 256, 64, True), (1, 2, 77, 128, False)])
def test_attention_with_score_bias(dev, B, H, S, D, per_head):
    """scores = scale * q.k + bias: CLIP's causal mask (one table for every head) and a per-head T5-style bias."""
    h = H * D
    qkv = randn(B, S, 3 * h, seed=60)
    ldb = (S + 63) // 64 * 64
    if per_head:
        bias = randn(H, S, S, seed=61, scale=2.0)
        scale = 1.0
    else:
        bias = ot.clip_causal_mask(S, True)[None].expand(H, S, S)
        scale = 1.0 / math.sqrt(D)
    bias = bf16r(bias)
    bd = torch.zeros(H if per_head else 1, S, ldb)
    bd[..., :S] = bias[: (H if per_head else 1)]
    y = attention_bias(g(qkv, dev), H, D, scale, g(bd, dev) if per_head else g(bd[0], dev), per_head)
    q, k, v = (qkv[..., i * h:(i + 1) * h].reshape(B, S, H, D).transpose(1, 2) for i in range(3))
    s = (q * scale) @ k.transpose(-1, -2) + bias[None]
    ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, S, h)
    assert rel_l2(ref, y.float()) < 2 * TOL_SINGLE_OP


def test_text_small_kernels(dev):
    from diffusionkit_amd import text as tx
    x = randn(37, 192, seed=62, scale=2.0) + 0.3
    w, b = randn(192, seed=63, scale=0.2) + 1.0, randn(192, seed=64, scale=0.1)
    y = tx._layernorm(g(x, dev), g(w, dev), g(b, dev))
    assert rel_l2(ot.layer_norm_affine(bf16r(x), bf16r(w), bf16r(b), 1e-5, Prec()), y.float()) < TOL_SINGLE_OP
    xf = randn(21, 256, seed=65, scale=30.0)
    y = tx._t5_rmsnorm(xf.to(dev), g(w[:192].repeat(2)[:256], dev), 1e-6)
    assert rel_l2(ot.t5_rms_norm(xf, bf16r(w[:192].repeat(2)[:256]), 1e-6, Prec()), y.float()) < TOL_SINGLE_OP
    table, pos = randn(50, 64, seed=66), randn(10, 64, seed=67)
    ids = torch.tensor([3, 49, 0, 7, 7, 12, 1, 2, 3, 4, 5, 6])
    e, ef = tx._embedding(g(table, dev), ids.to(dev), g(pos, dev), want_f32=True)
    ref = bf16r(bf16r(table)[ids] + bf16r(pos)[torch.arange(12) % 10])
    assert torch.equal(e.float().cpu(), ref) and torch.equal(ef.cpu(), ref)
    a, c = randn(1000, seed=68, scale=3.0), randn(1000, seed=69)
    assert rel_l2(ot.quick_gelu(bf16r(a), Prec()), tx._elementwise(0, g(a, dev)).float()) < TOL_SINGLE_OP
    assert rel_l2(bf16r(a) * bf16r(c), tx._elementwise(1, g(a, dev), g(c, dev)).float()) < TOL_SINGLE_OP
    r = randn(1000, seed=70).to(dev)
    r0 = r.clone()
    tx._elementwise(2, g(a, dev), r=r)
    assert torch.allclose(r.cpu(), r0.cpu() + bf16r(a))


@pytest.mark.parametrize("act,proj", [("quick_gelu", 64), ("gelu", 96), ("quick_gelu", None)])
def test_clip_text_engine(dev, act, proj):
    cfg = tiny_clip(act, proj)
    named = synth_clip_weights(cfg, seed=11)
    eng = CLIPTextEngine(cfg, named, dev)
    tokens = torch.randint(1, 500, (2, 77), generator=torch.Generator().manual_seed(3))
    tokens[0, 9:] = 511
    tokens[1, 76] = 511
    out = eng(tokens)
    wf = {k: v.float() for k, v in named.items()}
    res = {n: ot.OracleCLIPText(cfg, wf, P)(tokens) for n, P in (("fp32", Prec()), ("emu", Prec(BF)))}
    yardstick(out.hidden_states[-2].float(), res["emu"][2][-2], res["fp32"][2][-2], "clip hidden[-2]")
    yardstick(out.last_hidden_state.float(), res["emu"][1], res["fp32"][1], "clip last")
    yardstick(out.pooled_output.float(), res["emu"][0], res["fp32"][0], "clip pooled")
    assert len(out.hidden_states) == cfg.num_layers


@pytest.mark.parametrize("B,N", [(1, 64), (2, 150)])
def test_t5_encoder_engine(dev, B, N):
    cfg = tiny_t5()
    named = synth_t5_weights(cfg, seed=12)
    eng = T5EncoderEngine(cfg, named, dev)
    tokens = torch.randint(0, cfg.vocab_size, (B, N), generator=torch.Generator().manual_seed(4))
    out = eng(tokens)
    assert out.shape == (B, N, cfg.d_model) and out.dtype == BF
    wf = {k: v.float() for k, v in named.items()}
    res = {n: ot.OracleT5Encoder(cfg, wf, P)(tokens) for n, P in (("fp32", Prec()), ("emu", Prec(BF)))}
    yardstick(out.float(), res["emu"], res["fp32"], "t5 encoder")
    # the bias table the kernel builds from the host-side buckets == the oracle's
    tab = eng._position_bias(N)[:, :, :N].float().cpu()
    assert torch.equal(tab, bf16r(ot.OracleT5Encoder(cfg, wf).bias(N)))


def test_pipeline_with_text_encoders(dev, tmp_path):
    """generate_image(text=...) end to end on a FLUX-shaped toy: CLIP BPE tokenizer -> CLIP-L pooled + T5 tokens ->
    MMDiT -> VAE (mlx/__init__.py:642-671).  The T5 tokenizer is a stand-in with the reference's interface (no
    sentencepiece model is available offline)."""
    from diffusionkit_amd.config import tiny_flux, tiny_vae
    from diffusionkit_amd.pipeline import FluxPipeline
    from tests.test_text_oracle import _tiny_bpe
    vp, mp, vocab = _tiny_bpe(str(tmp_path))
    tok_l = Tokenizer.from_files(vp, mp, pad_with_eos=True)

    class ByteT5Tok:
        max_length, pad_to_max_length, pad_with_eos = 32, True, False

        def tokenize(self, s):
            return [3 + (b % 300) for b in s.encode()][: self.max_length - 1] + [1]

    ccfg, tcfg, cfg = tiny_clip("quick_gelu", 64), tiny_t5(), tiny_flux()
    clip_w = synth_clip_weights(ccfg, seed=21)
    # the toy vocabulary is smaller than the engine's: ids stay in range
    clip_l = CLIPTextEngine(ccfg, clip_w, dev)
    t5 = T5EncoderEngine(tcfg, synth_t5_weights(tcfg, seed=22), dev)
    cond = TextConditioner(clip_l, tok_l, t5=t5, t5_tokenizer=ByteT5Tok(), flux=True, t5_max_length=32)
    pipe = FluxPipeline(w16=True, a16=True, mmdit_config=cfg, vae_config=tiny_vae(), device=dev, text_len=32)
    pipe.set_text_encoder(cond)
    c, p = pipe.encode_text("the cat and the dog", cfg_weight=0.0)
    assert c.shape == (1, 32, tcfg.d_model) and p.shape == (1, 64)
    # oracle for the same prompt
    from diffusionkit_amd.text import tokenize_rows
    tl, tt = tokenize_rows(tok_l, "the cat and the dog", None), tokenize_rows(ByteT5Tok(), "the cat and the dog", None)
    f = lambda d: {k: v.float() for k, v in d.items()}
    res = {}
    for n, P in (("fp32", Prec()), ("emu", Prec(BF))):
        res[n] = ot.flux_conditioning(ot.OracleCLIPText(ccfg, f(clip_w), P), ot.OracleT5Encoder(tcfg, f(synth_t5_weights(tcfg, seed=22)), P),
                                      tl, tt, 32)
    yardstick(c.float(), res["emu"][0], res["fp32"][0], "flux conditioning")
    yardstick(p.float(), res["emu"][1], res["fp32"][1], "flux pooled")
    img, log = pipe.generate_image("the cat and the dog", num_steps=2, latent_size=(8, 8), seed=1, verbose=False)
    assert img.size == (64, 64) and log["text_encoding"]["synthetic"] is False


def test_sd3_conditioning_assembly(dev, tmp_path):
    """DiffusionPipeline.encode_text (mlx/__init__.py:197-251): [CLIP-L | CLIP-G | zeros to 4096] tokens followed by the T5
    tokens (zeros when T5 is off), pooled = [pooled_L | pooled_G]; prompt + negative prompt rows when cfg_weight > 1."""
    from diffusionkit_amd.text import tokenize_rows
    from tests.test_text_oracle import _tiny_bpe
    vp, mp, _ = _tiny_bpe(str(tmp_path))
    tok_l, tok_g = Tokenizer.from_files(vp, mp, pad_with_eos=True), Tokenizer.from_files(vp, mp, pad_with_eos=False)
    cl, cg = tiny_clip("quick_gelu", 64), tiny_clip("gelu", 96)
    wl, wg = synth_clip_weights(cl, seed=31), synth_clip_weights(cg, seed=32)
    cond = TextConditioner(CLIPTextEngine(cl, wl, dev), tok_l, clip_g=CLIPTextEngine(cg, wg, dev), tokenizer_g=tok_g, flux=False)
    c, p = cond("the cat", cfg_weight=5.0, negative_text="dog")
    assert c.shape == (2, 154, 4096) and p.shape == (2, 64 + 96)
    f = lambda d: {k: v.float() for k, v in d.items()}
    tl, tg = tokenize_rows(tok_l, "the cat", "dog"), tokenize_rows(tok_g, "the cat", "dog")
    res = {n: ot.sd3_conditioning(ot.OracleCLIPText(cl, f(wl), P), ot.OracleCLIPText(cg, f(wg), P), None, tl, tg, None)
           for n, P in (("fp32", Prec()), ("emu", Prec(BF)))}
    yardstick(c.float(), res["emu"][0], res["fp32"][0], "sd3 conditioning")
    yardstick(p.float(), res["emu"][1], res["fp32"][1], "sd3 pooled")
    assert torch.all(c[:, :, 256:] == 0) and torch.all(c[:, 77:] == 0)
    c1, p1 = cond("the cat", cfg_weight=0.0)  # cfg_weight <= 1: the second row is still there, built from "" (:177-187)
    assert c1.shape == (2, 154, 4096) and p1.shape == (2, 160)
    assert torch.equal(c1[0], c[0]) and not torch.equal(c1[1], c[1])
