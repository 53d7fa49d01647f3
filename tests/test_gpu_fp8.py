"""fp8 path (BASELINE.json configs[3]): MX-fp8 quantisers, the block-scaled fp8 GEMM and the fp8 MMDiT engine against the oracle
(oracle/fp8.py restates both quantisers; the oracle model then runs on exactly the values the fp8 GEMM multiplies).

Tolerances: the quantisers are integer / bit work on bf16 inputs -> bit-exact bytes and scales.  The GEMM accumulates exact
e4m3 x e4m3 x 2^k products in fp32 -> one bf16 output rounding, rel-L2 <= 3e-3 like the bf16 kernels (tests/_util.py).  Model
level: yardstick against the bf16-emulating oracle with the same fake-quantisation; against the un-quantised fp32 oracle the test
only REPORTS the distance (that is the price of fp8, not an error of the kernels) and asserts a loose floor.
"""
from dataclasses import replace

import pytest
import torch

from diffusionkit_amd import ops
from diffusionkit_amd._lib import DK_EPI_BIAS, DK_EPI_BIAS_GELU, DK_EPI_GATE_RES
from diffusionkit_amd.config import FLUX_SCHNELL, tiny_flux
from diffusionkit_amd.weights import dequantize_weight_e4m3, pack_mmdit, quantize_weight_e4m3, synth_mmdit_weights
from oracle import fp8 as o8
from oracle.mmdit import OracleMMDiT, Prec
from tests import _fp8 as f8
from tests._util import BF, TOL_SINGLE_OP, bf16r, psnr, randn, rel_l2

pytestmark = pytest.mark.gpu


# (2560, 384, 2432: h / 8 is not a multiple of 64 -- the last of a lane's 16-byte chunks lies past the row end for some lanes and relies on the
#  per-row buffer resource: zeros in, stores dropped)
@pytest.mark.parametrize("M,h", [(256, 256), (300, 3072), (4352, 3072), (128, 12288 // 8), (200, 2560), (130, 384), (64, 2432)])
def test_quantize_mx8_bit_exact(dev, M, h):
    x = randn(M, h, seed=M + h, scale=2.0)
    x[3, 5:40] = 0.0
    x[min(7, M - 1)] *= 300.0  # rows with very different magnitudes: per-block scales
    x[min(9, M - 1), :32] = 0.0  # an all-zero block
    x = bf16r(x)
    q, sc = ops.quantize_mx8(x.to(dev, BF))
    q_ref, e_ref = o8.mx8_encode(x)
    assert torch.equal(q.cpu(), q_ref)
    assert torch.equal(f8.array_to_scales(sc, M, h), e_ref)


def test_quantize_mx8_into_wider_buffer(dev):
    """the attention output of a single block goes into columns [0, h) of the [rows, 5h + pad] CAT buffer at a row offset"""
    M, h, rows, ld = 256, 512, 640, 2560 + 128
    x = bf16r(randn(M, h, seed=11, scale=3.0))
    out = torch.zeros(rows, ld, dtype=torch.uint8, device=dev)
    sc = torch.zeros(ops.mx_scale_bytes(rows, ld), dtype=torch.uint8, device=dev)
    ops.quantize_mx8(x.to(dev, BF), out=out, scales=sc, out_row0=128, out_col0=512)
    q_ref, e_ref = o8.mx8_encode(x)
    assert torch.equal(out[128:128 + M, 512:512 + h].cpu(), q_ref)
    assert int(out[:128].max()) == 0 and int(out[128:128 + M, :512].max()) == 0
    assert torch.equal(f8.array_to_scales(sc, M, h, rows=rows, row0=128, col0=512), e_ref)


@pytest.mark.parametrize("h", [1024, 2560, 384])
def test_ln_modulate_mx8(dev, h):
    from oracle.mmdit import affine_transform
    B, S = 2, 192
    x = randn(B * S, h, seed=1, scale=1.5)
    shift, scale = randn(B, h, seed=2, scale=0.3), randn(B, h, seed=3, scale=0.3)
    q, sc = ops.ln_modulate_mx8(x.to(dev, BF), shift.to(dev, BF), scale.to(dev, BF), mod_seg_len=S)
    got = f8.mx8_decode(q, f8.array_to_scales(sc, B * S, h))
    y = torch.cat([affine_transform(x[b * S:(b + 1) * S][None], shift[b][None, None], scale[b][None, None], 1e-6, Prec(BF))[0] for b in range(B)])
    want = o8.mx8_fake_quant(y)
    # the LayerNorm arithmetic may differ in the last bf16 bit on a few elements (rsqrt); the quantisation of equal inputs is exact
    assert rel_l2(want, got) < 4e-3
    assert float((got != want).float().mean()) < 0.02


def _fp8_problem(M, N, K, seed, w_pitch=None):
    a = bf16r(randn(M, K, seed=seed, scale=1.0))
    a[:, :K // 2] *= 4.0
    w = (randn(N, K, seed=seed + 1, scale=0.02)).to(BF)
    qa, ea = o8.mx8_encode(a)
    qw, ws = quantize_weight_e4m3(w)
    a_dq = f8.mx8_decode(qa, ea)
    w_dq = dequantize_weight_e4m3(qw, ws)
    if w_pitch:
        wp = torch.zeros(N, w_pitch, dtype=torch.uint8)
        wp[:, :K] = qw
        qw = wp
    return a, qa, ea, qw, ws, a_dq, w_dq


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 512, 256), (384, 256, 384), (4352, 3072, 3072), (1024, 3072, 12288)])
def test_gemm_fp8_bias(dev, M, N, K):
    a, qa, ea, qw, ws, a_dq, w_dq = _fp8_problem(M, N, K, seed=M + N + K, w_pitch=K + 128 if K >= 8192 else None)
    bias = randn(N, seed=5, scale=0.5)
    rows = (M + 127) // 128 * 128
    a8 = torch.zeros(rows, K, dtype=torch.uint8)
    a8[:M] = qa
    out = ops.gemm_fp8(a8.to(dev), f8.scales_to_array(ea, rows).to(dev), qw.to(dev), ws.to(dev), bias=bias.to(dev, BF), M=M, k=K)
    ref = a_dq @ w_dq.t() + bias
    assert rel_l2(ref, out.float()) < TOL_SINGLE_OP, rel_l2(ref, out.float())


@pytest.mark.parametrize("M,N,K,split", [(1280, 3072, 12288, True),    # FLUX 512 x 512 fc2: 60 tiles, four K ranges each
                                          (1280, 3072, 15360, True),    # ... linear2
                                          (2560, 3072, 12288, True),    # 768 x 768: 120 tiles, two K ranges
                                          (1280, 3072, 3072, False),    # o_proj: below the break-even, stays whole
                                          (4352, 3072, 12288, False)])  # 1024 x 1024: a round of the CUs, untouched
def test_gemm_fp8_small_launch_is_split_automatically(dev, M, N, K, split):
    """Round 6: the fp8 GEMM's K split (gemm256f8.hip: F8Split) -- a launch of at most half a round of 256 x 256 tiles with a long reduction is cut along K
    when the caller hands in the split workspace, as the engines do (FLUX below 1024 x 1024).  Raw fp32 accumulators are exchanged, so the result is the
    unsplit one up to the fp32 summation order; against the exact product of the dequantised operands, and against the same launch without the workspace."""
    a, qa, ea, qw, ws, a_dq, w_dq = _fp8_problem(M, N, K, seed=M + N + K, w_pitch=K + 128 if K >= 8192 else None)
    bias, gate, res = randn(N, seed=5, scale=0.5), randn(1, N, seed=6), randn(M, N, seed=7)
    rows = (M + 127) // 128 * 128
    a8 = torch.zeros(rows, K, dtype=torch.uint8)
    a8[:M] = qa
    wsp = ops.gemm_workspace(dev)
    args = (a8.to(dev), f8.scales_to_array(ea, rows).to(dev), qw.to(dev), ws.to(dev))
    kw = dict(bias=bias.to(dev, BF), epilogue=DK_EPI_GATE_RES, gate=gate.to(dev, BF), res=res.to(dev, BF), gate_seg_len=M, M=M, k=K)
    y = ops.gemm_fp8(*args, workspace=wsp, **kw)
    y0 = ops.gemm_fp8(*args, **kw)
    lin = bf16r(a_dq @ w_dq.t() + bias)
    ref = res + bf16r(gate * lin)
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP and rel_l2(ref, y0.float()) < TOL_SINGLE_OP
    if split:
        assert not torch.equal(y0, y), "the launch was expected to be cut along K (other summation order)"
        assert float((y0.float() != y.float()).float().mean()) < 0.05
    else:
        assert torch.equal(y0, y)
    assert int(wsp[-4096:].sum()) == 0  # the flag region (and the error word) is left zero


def test_gemm_fp8_gate_res(dev):
    M, N, K, S = 512, 256, 256, 256
    a, qa, ea, qw, ws, a_dq, w_dq = _fp8_problem(M, N, K, seed=21)
    bias, gate, res = randn(N, seed=5, scale=0.5), randn(M // S, N, seed=6), randn(M, N, seed=7)
    out = ops.gemm_fp8(qa.to(dev), f8.scales_to_array(ea).to(dev), qw.to(dev), ws.to(dev), bias=bias.to(dev, BF), epilogue=DK_EPI_GATE_RES,
                       gate=gate.to(dev, BF), res=res.to(dev, BF), gate_seg_len=S)
    lin = bf16r(a_dq @ w_dq.t() + bias)
    ref = res + bf16r(gate.repeat_interleave(S, 0) * lin)
    assert rel_l2(ref, out.float()) < TOL_SINGLE_OP


def test_gemm_fp8_gelu_mx8_output_feeds_next_gemm(dev):
    """fc1 -> GELU -> MX-fp8 (written by the GEMM tail) -> fc2: the chain of a transformer block's MLP"""
    from oracle.mmdit import gelu_erf
    M, h, r = 512, 256, 4
    a, qa, ea, qw1, ws1, a_dq, w1_dq = _fp8_problem(M, r * h, h, seed=31)
    b1 = randn(r * h, seed=8, scale=0.5)
    hq, hs = ops.gemm_fp8(qa.to(dev), f8.scales_to_array(ea).to(dev), qw1.to(dev), ws1.to(dev), bias=b1.to(dev, BF), epilogue=DK_EPI_BIAS_GELU,
                          out_mx8=True)
    hid = gelu_erf(bf16r(a_dq @ w1_dq.t() + b1), Prec(BF))
    want_q, want_e = o8.mx8_encode(hid)
    got = f8.mx8_decode(hq, f8.array_to_scales(hs, M, r * h))
    want = f8.mx8_decode(want_q, want_e)
    # the bf16 value in front of the quantiser may differ by one bf16 ulp on a few elements (fp32 summation order, erf polynomial)
    assert rel_l2(want, got) < 6e-3
    w2 = randn(h, r * h, seed=33, scale=0.02).to(BF)
    qw2, ws2 = quantize_weight_e4m3(w2)
    out = ops.gemm_fp8(hq, hs, qw2.to(dev), ws2.to(dev))
    ref = got @ dequantize_weight_e4m3(qw2, ws2).t()  # on the hidden values the kernel itself produced
    assert rel_l2(ref, out.float()) < TOL_SINGLE_OP


def _fp8_forward_case(cfg, dev, B, Hl, Wl, S_t, ts, step, guidance=None):
    from diffusionkit_amd.engine import MMDiTEngine
    named = synth_mmdit_weights(cfg, seed=1234)
    eng = MMDiTEngine(cfg, pack_mmdit(cfg, named, dev))
    if guidance is not None:
        eng.guidance = guidance
    text = randn(B, S_t, cfg.token_level_text_embed_dim, seed=3)
    pooled = randn(B, cfg.pooled_text_embed_dim, seed=4)
    lat = randn(B, Hl, Wl, 16, seed=5)
    eng.prepare(B, (Hl, Wl), S_t, len(ts))
    eng.cache_modulation_params(pooled.to(dev), ts)
    out = eng.forward_tokens(eng.patchify(lat.to(dev)), text.to(dev, BF), step)
    res = {}
    fq = o8.fake_quant_block_weights(cfg, named)
    plain = {k: v.float() for k, v in named.items()}
    paq = o8.policy_act_quant(cfg)  # (= mx8_fake_quant on every block Linear unless cfg carries a precision policy)
    for name, w, P, aq in (("fq_emu", fq, Prec(BF), paq), ("fq_fp32", fq, Prec(), paq), ("fp32", plain, Prec(), None)):
        m = OracleMMDiT(cfg, w, P, act_quant=aq, guidance=guidance)
        m.cache_modulation_params(pooled, torch.tensor(ts))
        taps = {}
        m(lat, text, ts[step], taps=taps)
        res[name] = taps["final"]
    return out.float().cpu(), res


@pytest.mark.parametrize("B", [1, 2])
def test_mmdit_fp8_tiny(dev, B):
    cfg = replace(tiny_flux(depth_multimodal=2, depth_unified=2, heads=2), weight_dtype="fp8_e4m3")
    out, res = _fp8_forward_case(cfg, dev, B, 32, 32, 128, [1000.0, 752.0, 500.0], 1)
    e_h, e_e = rel_l2(res["fq_fp32"], out), rel_l2(res["fq_fp32"], res["fq_emu"])
    assert e_h <= 2.0 * e_e + 2e-3, (e_h, e_e)
    assert psnr(res["fq_fp32"], out) > 35.0
    # what fp8 costs against the un-quantised model (reported; bounded by what the bf16-emulating fake-quant oracle reaches)
    print(f"fp8 tiny B={B}: PSNR vs fp32 un-quantised oracle {psnr(res['fp32'], out):.1f} dB (fake-quant oracle: fp32 arithmetic "
          f"{psnr(res['fp32'], res['fq_fp32']):.1f} dB, bf16-emulating {psnr(res['fp32'], res['fq_emu']):.1f} dB)")
    assert psnr(res["fp32"], out) > psnr(res["fp32"], res["fq_emu"]) - 3.0


def test_mmdit_fp8_precision_policy_tiny(dev):
    """MMDiTConfig.fp8_bf16_double_blocks (round 5): the first double-stream block keeps bf16 Linears (bf16 weights, bf16 activations, the
    bf16 GEMM), the second one and the single blocks run the fp8 path -- against the oracle with exactly those blocks fake-quantised, and
    closer to the un-quantised oracle than the all-fp8 engine"""
    base = tiny_flux(depth_multimodal=2, depth_unified=2, heads=2)
    cfg = replace(base, weight_dtype="fp8_e4m3", fp8_bf16_double_blocks=1)
    packed = pack_mmdit(cfg, synth_mmdit_weights(cfg, seed=1234), dev)
    assert "multimodal_transformer_blocks.0.image_transformer_block.attn.qkv.weight" in packed
    assert "multimodal_transformer_blocks.1.image_transformer_block.attn.qkv.weight_fp8" in packed
    out, res = _fp8_forward_case(cfg, dev, 1, 32, 32, 128, [1000.0, 752.0, 500.0], 1)
    e_h, e_e = rel_l2(res["fq_fp32"], out), rel_l2(res["fq_fp32"], res["fq_emu"])
    assert e_h <= 2.0 * e_e + 2e-3, (e_h, e_e)
    out_all, _ = _fp8_forward_case(replace(base, weight_dtype="fp8_e4m3"), dev, 1, 32, 32, 128, [1000.0, 752.0, 500.0], 1)
    p_pol, p_all = psnr(res["fp32"], out), psnr(res["fp32"], out_all)
    print(f"fp8 tiny, first double block bf16: {p_pol:.1f} dB against the un-quantised oracle, every block fp8: {p_all:.1f} dB")
    assert p_pol > p_all


def test_mmdit_fp8_rejects_unaligned_token_counts(dev):
    from diffusionkit_amd.engine import MMDiTEngine
    cfg = replace(tiny_flux(depth_multimodal=1, depth_unified=1, heads=2), weight_dtype="fp8_e4m3")
    eng = MMDiTEngine(cfg, pack_mmdit(cfg, synth_mmdit_weights(cfg, seed=1), dev))
    with pytest.raises(Exception, match="multiples of 128"):
        eng.prepare(1, (8, 12), 20, 2)


def test_flux_width_block_pair_fp8(dev):
    """FLUX geometry (h 3072, 24 heads, S = 256 + 4096), depth 1 + 1, fp8 weights: every fp8 launch at the BASELINE shapes
    (grouped text + image GEMMs, the column-split linear1 with its MX-fp8 second output, K = 15360 linear2)."""
    from tests.test_gpu_model import flux_pair_case
    out, res = flux_pair_case(dev, fp8=True)
    e_h, e_e = rel_l2(res["fq_fp32"], out), rel_l2(res["fq_fp32"], res["fq_emu"])
    print(f"fp8 flux width: hip-vs-fq {e_h:.3e}, emu-vs-fq {e_e:.3e}; PSNR vs un-quantised fp32 {psnr(res['fp32'], out):.1f} dB "
          f"(fake-quant oracle {psnr(res['fp32'], res['fq_fp32']):.1f} dB)")
    # round 4 (VERDICT r3 "weak"): the gate sits at the emulation's own level (measured 4.903e-2 against 4.856e-2) -- an operand or scale
    # error that costs 1 dB (x 1.12) on top of the format's noise fails; the quantisers themselves are bit-exact tests above
    assert e_h <= 1.1 * e_e + 5e-4, (e_h, e_e)
    assert psnr(res["fp32"], out) >= psnr(res["fp32"], res["fq_fp32"]) - 0.5  # and no further from the UN-quantised oracle than the fake-quant oracle is (36.7 / 36.8 dB)


def test_guidance_embedding(dev):
    """FLUX.1-dev guidance embedding (config.guidance_embed, mmdit.py:31-36): changes the output, follows the oracle restatement,
    and is off by default (quirk Q7)."""
    cfg = replace(tiny_flux(), guidance_embed=True)
    from diffusionkit_amd.engine import MMDiTEngine
    from tests.test_gpu_model import yardstick_ok
    named = synth_mmdit_weights(cfg, seed=1234)
    assert "guidance_in.mlp.layers.0.weight" in named and "guidance_in.mlp.layers.0.weight" not in synth_mmdit_weights(tiny_flux(), shapes_only=True)
    eng = MMDiTEngine(cfg, pack_mmdit(cfg, named, dev))
    text, pooled, lat = randn(1, 20, cfg.token_level_text_embed_dim, seed=3), randn(1, cfg.pooled_text_embed_dim, seed=4), randn(1, 8, 12, 16, seed=5)
    ts = [1000.0, 752.0]
    outs = {}
    for g in (3.5, 1.0):
        eng.guidance = g
        eng.prepare(1, (8, 12), 20, len(ts))
        eng.cache_modulation_params(pooled.to(dev), ts)
        outs[g] = eng.forward_tokens(eng.patchify(lat.to(dev)), text.to(dev, BF), 1).float().cpu()
        wf = {k: v.float() for k, v in named.items()}
        ref = {}
        for name, P in (("fp32", Prec()), ("emu", Prec(BF))):
            m = OracleMMDiT(cfg, wf, P, guidance=g)
            m.cache_modulation_params(pooled, torch.tensor(ts))
            taps = {}
            m(lat, text, ts[1], taps=taps)
            ref[name] = taps["final"]
        yardstick_ok(outs[g], ref["emu"], ref["fp32"], f"guidance {g}")
    assert rel_l2(outs[3.5], outs[1.0]) > 1e-3
