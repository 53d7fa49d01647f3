"""Operator-level parity (MI355X): every C-ABI op against the CPU oracle on the same seeded,
bf16-representable inputs.  Tolerance for one fp32-accumulated result rounded once to bf16:
rel-L2 <= 3e-3 (tests/_util.py)."""
import math

import pytest
import torch

from oracle import mmdit as om
from oracle import vae as ov
from oracle.mmdit import Prec
from tests._util import BF, TOL_SINGLE_OP, bf16r, max_abs, randn, rel_l2

pytestmark = pytest.mark.gpu


def g(x, dev):
    return x.to(dev, BF).contiguous()


# ---- GEMM ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (100, 200, 192), (1178, 1536, 1536),
                                   (4352, 3072, 3072), (7, 64, 256), (333, 3, 1152), (2, 344 * 128, 128)])
def test_gemm_bias(dev, M, N, K):
    from diffusionkit_amd import ops
    x, w, b = randn(M, K, seed=1), randn(N, K, seed=2, scale=0.05), randn(N, seed=3)
    if N % 4:
        out = torch.empty(M, (N + 3) // 4 * 4, dtype=BF, device=dev)
        y = ops.linear(g(x, dev), g(w, dev), g(b, dev), out=out)[:, :N]
    else:
        y = ops.linear(g(x, dev), g(w, dev), g(b, dev))
    ref = x @ w.t() + b
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP
    # transpose-detecting: asymmetric problem, and the worst element is within 2 bf16 ulps of the scale
    assert max_abs(ref, y.float()) < 0.02 * float(ref.abs().max()) + 1e-2


def test_gemm_identity_asymmetric(dev):
    """A = I, asymmetric W: catches operand / output transposes (guide rule 16)."""
    from diffusionkit_amd import ops
    K = 128
    x = torch.eye(K)
    w = bf16r(torch.arange(256 * K, dtype=torch.float32).reshape(256, K) % 251 - 100)
    y = ops.linear(g(x, dev), g(w, dev))
    assert torch.equal(y.float().cpu(), w.t().contiguous())


@pytest.mark.parametrize("epi", ["gelu", "silu", "gate_res", "res"])
def test_gemm_epilogues(dev, epi):
    from diffusionkit_amd import ops
    B, S, K, N = 2, 160, 256, 384
    M = B * S
    x, w, b = randn(M, K, seed=4), randn(N, K, seed=5, scale=0.05), randn(N, seed=6, scale=0.1)
    res, gate = randn(M, N, seed=7), randn(B, N, seed=8)
    acc = bf16r(x @ w.t() + b)
    if epi == "gelu":
        y = ops.linear(g(x, dev), g(w, dev), g(b, dev), epilogue=ops.DK_EPI_BIAS_GELU)
        ref = om.gelu_erf(acc, Prec())
    elif epi == "silu":
        y = ops.linear(g(x, dev), g(w, dev), g(b, dev), epilogue=ops.DK_EPI_BIAS_SILU)
        ref = acc * torch.sigmoid(acc)
    elif epi == "gate_res":
        y = ops.linear(g(x, dev), g(w, dev), g(b, dev), epilogue=ops.DK_EPI_GATE_RES, gate=g(gate, dev), res=g(res, dev),
                       gate_seg_len=S)
        ref = res + bf16r(gate.repeat_interleave(S, 0) * acc)
    else:
        y = ops.linear(g(x, dev), g(w, dev), g(b, dev), epilogue=ops.DK_EPI_RES, res=g(res, dev))
        ref = res + acc
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP


def test_gemm_segment_mapping_in_place(dev):
    """Image rows of a joint [B, S, h] buffer: A and C/res use (seg_len, seg_stride) addressing, C
    aliases the residual (the o_proj call of post_sdpa)."""
    from diffusionkit_amd import ops
    B, S_t, S_i, h = 2, 24, 136, 128
    S = S_t + S_i
    att, X = randn(B, S, h, seed=9), randn(B, S, h, seed=10)
    w, b, gate = randn(h, h, seed=11, scale=0.08), randn(h, seed=12, scale=0.1), randn(B, 3 * h, seed=13)
    Xd, attd, gd = g(X, dev), g(att, dev), g(gate, dev)
    esz = 2
    ops.gemm_desc_call(A=attd.data_ptr() + S_t * h * esz, W=g(w, dev), C=Xd.data_ptr() + S_t * h * esz, bias=g(b, dev),
                       gate=gd.data_ptr() + h * esz, res=Xd.data_ptr() + S_t * h * esz, M=B * S_i, N=h, K=h, lda=h, ldc=h, ldr=h,
                       a_seg_len=S_i, a_seg_stride=S, c_seg_len=S_i, c_seg_stride=S, r_seg_len=S_i, r_seg_stride=S,
                       gate_seg_len=S_i, gate_stride=3 * h, alpha=1.0, epilogue=ops.DK_EPI_GATE_RES)
    ref = X.clone()
    o = bf16r(att[:, S_t:] @ w.t() + b)
    ref[:, S_t:] = X[:, S_t:] + bf16r(gate[:, None, h:2 * h] * o)
    got = Xd.float().cpu()
    assert torch.equal(got[:, :S_t], X[:, :S_t])  # text rows untouched
    assert rel_l2(ref[:, S_t:], got[:, S_t:]) < TOL_SINGLE_OP


GEMM_MODES = {-1: "automatic choice", 128: "128^2 tiles", 9: "256^2 tiles (16x16x32 MFMA, LDS-DMA ring)",
              10: "256^2 tiles, one wave per SIMD, hand-scheduled asm body (gemm256v4.hip)"}


@pytest.mark.parametrize("mode", sorted(GEMM_MODES))
@pytest.mark.parametrize("epi", ["bias", "gelu", "gate_res"])
def test_gemm_kernel_variants(dev, mode, epi):
    """Both GEMM kernels behind dk_tune_set("gemm", mode) against the oracle, on a shape both accept (two row segments of 512),
    with a multi-tile K loop (10 K-tiles: the ring of two activation and three weight slots wraps several times)."""
    from diffusionkit_amd import ops
    B, S, K, N = 2, 512, 640, 768
    M = B * S
    x, w, b = randn(M, K, seed=14), randn(N, K, seed=15, scale=0.05), randn(N, seed=16, scale=0.1)
    res, gate = randn(M, N, seed=17), randn(B, N, seed=18)
    acc = bf16r(x @ w.t() + b)
    ws = ops.gemm_workspace(dev)
    try:
        ops.tune("gemm", mode)
        if epi == "bias":
            y = ops.linear(g(x, dev), g(w, dev), g(b, dev), workspace=ws)
            ref = acc
        elif epi == "gelu":
            y = ops.linear(g(x, dev), g(w, dev), g(b, dev), epilogue=ops.DK_EPI_BIAS_GELU, workspace=ws)
            ref = om.gelu_erf(acc, Prec())
        else:
            y = ops.linear(g(x, dev), g(w, dev), g(b, dev), epilogue=ops.DK_EPI_GATE_RES, gate=g(gate, dev), res=g(res, dev),
                           gate_seg_len=S, workspace=ws)
            ref = res + bf16r(gate.repeat_interleave(S, 0) * acc)
    finally:
        ops.tune("gemm", -1)
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP, GEMM_MODES[mode]
    assert max_abs(ref, y.float()) < 0.02 * float(ref.abs().max()) + 1e-2
    # the flag region of the K-split workspace must be left zero (flags are reset by their consumer)
    assert int(ws[-4096:].sum()) == 0


@pytest.mark.parametrize("mf", [8, 7])
@pytest.mark.parametrize("epi", ["bias", "gelu", "gate_res"])
@pytest.mark.parametrize("M,N,K", [(1100, 384, 640), (2356, 2432, 2432)])
def test_gemm_256_half_column_tile(dev, M, N, K, epi, mf):
    """N % 256 == 128 on the 256^2 kernel (SD3.5-large's h = 38 * 64 = 2432, config.py:72-74: q / k / v, o_proj and fc2 end in half
    a column tile): every output column against the oracle, and the bytes BEHIND the last column of each row untouched"""
    from diffusionkit_amd import ops
    x, w, b = randn(M, K, seed=21), randn(N, K, seed=22, scale=0.03), randn(N, seed=23, scale=0.1)
    res, gate = randn(M, N, seed=24), randn(2, N, seed=25)
    seg = (M + 1) // 2
    acc = bf16r(x @ w.t() + b)
    out = torch.full((M, N + 128), 7.0, dtype=BF, device=dev)  # row pitch N + 128: a store into the missing half tile would land here
    ws = ops.gemm_workspace(dev)
    try:
        ops.tune("gemm", 9)
        ops.tune("gemm_mf", mf)
        if epi == "bias":
            ops.linear(g(x, dev), g(w, dev), g(b, dev), out=out[:, :N], workspace=ws)
            ref = acc
        elif epi == "gelu":
            ops.linear(g(x, dev), g(w, dev), g(b, dev), out=out[:, :N], epilogue=ops.DK_EPI_BIAS_GELU, workspace=ws)
            ref = om.gelu_erf(acc, Prec())
        else:
            ops.linear(g(x, dev), g(w, dev), g(b, dev), out=out[:, :N], epilogue=ops.DK_EPI_GATE_RES, gate=g(gate, dev), res=g(res, dev),
                       gate_seg_len=seg, workspace=ws)
            ref = res + bf16r(gate.repeat_interleave(seg, 0)[:M] * acc)
    finally:
        ops.tune("gemm", -1)
        ops.tune("gemm_mf", -1)
    y = out[:, :N].float().cpu()
    assert rel_l2(ref, y) < TOL_SINGLE_OP
    assert max_abs(ref, y) < 0.02 * float(ref.abs().max()) + 1e-2
    assert bool((out[:, N:] == 7.0).all())


def test_gemm_256_joint_stream_in_place(dev):
    """The 256^2 kernel on the image rows of a joint [B, S, h] buffer (segment maps evaluated once per tile), C aliasing the
    residual as in post_sdpa."""
    from diffusionkit_amd import ops
    B, S_t, S_i, h = 2, 256, 512, 256
    S = S_t + S_i
    att, X = randn(B, S, h, seed=9), randn(B, S, h, seed=10)
    w, b, gate = randn(h, h, seed=11, scale=0.08), randn(h, seed=12, scale=0.1), randn(B, 3 * h, seed=13)
    ref = X.clone()
    o = bf16r(att[:, S_t:] @ w.t() + b)
    ref[:, S_t:] = X[:, S_t:] + bf16r(gate[:, None, h:2 * h] * o)
    ws = ops.gemm_workspace(dev)
    for mode in (9, 10, -1):
        Xd, attd, gd = g(X, dev), g(att, dev), g(gate, dev)
        try:
            ops.tune("gemm", mode)
            ops.gemm_desc_call(A=attd.data_ptr() + S_t * h * 2, W=g(w, dev), C=Xd.data_ptr() + S_t * h * 2, bias=g(b, dev),
                               gate=gd.data_ptr() + h * 2, res=Xd.data_ptr() + S_t * h * 2, M=B * S_i, N=h, K=h, lda=h, ldc=h, ldr=h,
                               a_seg_len=S_i, a_seg_stride=S, c_seg_len=S_i, c_seg_stride=S, r_seg_len=S_i, r_seg_stride=S,
                               gate_seg_len=S_i, gate_stride=3 * h, alpha=1.0, epilogue=ops.DK_EPI_GATE_RES,
                               workspace=ws.data_ptr(), workspace_bytes=ws.numel())
        finally:
            ops.tune("gemm", -1)
        got = Xd.float().cpu()
        assert torch.equal(got[:, :S_t], X[:, :S_t])
        assert rel_l2(ref[:, S_t:], got[:, S_t:]) < TOL_SINGLE_OP, mode


@pytest.mark.parametrize("mf", [8, 7, 104])
@pytest.mark.parametrize("epi", ["bias", "gate_res", "res"])
@pytest.mark.parametrize("B,S_t,S_i", [(2, 589, 64), (3, 77, 11), (1, 300, 0)])
def test_gemm_v3_ragged_and_straddling_segments(dev, B, S_t, S_i, epi, mf):
    """(mf 104: the same case on gemm256v4.hip, dk_tune_set("gemm", 10))  The 16x16x32-MFMA 256^2 kernel on the text stream of a joint [B, S_t + S_i] buffer: M = B * S_t is not
    a multiple of 256 and (B > 1) tiles straddle the row segments of every map (A, C, residual, gate), so
    the per-lane row walk of the tail and the clamped DMA rows are exercised; rows of the other stream and
    the rows behind M must stay untouched."""
    from diffusionkit_amd import ops
    h, N = 192, 512
    S = S_t + S_i
    att, X = randn(B, S, h, seed=40), randn(B, S, N, seed=41)
    w, b, gate = randn(N, h, seed=42, scale=0.08), randn(N, seed=43, scale=0.1), randn(B, 2 * N, seed=44)
    Xd, attd, gd = g(X, dev), g(att, dev), g(gate, dev)
    kw = dict(A=attd, W=g(w, dev), C=Xd, bias=g(b, dev), M=B * S_t, N=N, K=h, lda=h, ldc=N,
              a_seg_len=S_t, a_seg_stride=S, c_seg_len=S_t, c_seg_stride=S, alpha=1.0)
    o = bf16r(att[:, :S_t] @ w.t() + b)
    ref = X.clone()
    if epi == "bias":
        kw.update(epilogue=ops.DK_EPI_BIAS)
        ref[:, :S_t] = o
    elif epi == "res":
        kw.update(epilogue=ops.DK_EPI_RES, res=Xd, ldr=N, r_seg_len=S_t, r_seg_stride=S)
        ref[:, :S_t] = X[:, :S_t] + o
    else:
        kw.update(epilogue=ops.DK_EPI_GATE_RES, res=Xd, ldr=N, r_seg_len=S_t, r_seg_stride=S,
                  gate=gd.data_ptr() + N * 2, gate_seg_len=S_t, gate_stride=2 * N)
        ref[:, :S_t] = X[:, :S_t] + bf16r(gate[:, None, N:] * o)
    try:
        ops.tune("gemm", 10 if mf == 104 else 9)
        ops.tune("gemm_mf", -1 if mf == 104 else mf)  # 256-row and 224-row tiles
        ops.gemm_desc_call(**kw)
    finally:
        ops.tune("gemm", -1)
        ops.tune("gemm_mf", -1)
    got = Xd.float().cpu()
    assert torch.equal(got[:, S_t:], X[:, S_t:])  # rows of the other stream untouched
    assert rel_l2(ref[:, :S_t], got[:, :S_t]) < TOL_SINGLE_OP
    assert max_abs(ref[:, :S_t], got[:, :S_t]) < 0.02 * float(ref.abs().max()) + 1e-2


@pytest.mark.parametrize("M,N,K", [(4352, 3072, 3072), (224, 256, 64), (449, 512, 192), (4608, 256, 128)])
def test_gemm_v3_tile_heights_agree(dev, M, N, K):
    """224-row tiles (gemm_mf 7) against 256-row tiles (8): the same fp32 accumulation order per output element, so the results
    are bit-identical; M exactly one / two tiles, one row over, and the FLUX shapes the automatic choice switches on."""
    from diffusionkit_amd import ops
    x, w, b = randn(M, K, seed=60), randn(N, K, seed=61, scale=0.05), randn(N, seed=62, scale=0.1)
    outs = {}
    for mf in (8, 7, -1):
        try:
            ops.tune("gemm", 9)
            ops.tune("gemm_mf", mf)
            outs[mf] = ops.linear(g(x, dev), g(w, dev), g(b, dev))
        finally:
            ops.tune("gemm", -1)
            ops.tune("gemm_mf", -1)
    assert rel_l2(bf16r(x @ w.t() + b), outs[8].float()) < TOL_SINGLE_OP
    assert torch.equal(outs[8], outs[7]) and torch.equal(outs[8], outs[-1])


@pytest.mark.parametrize("mf", [8, 7])
@pytest.mark.parametrize("epi", ["bias", "gelu", "gate_res"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 512, 128), (1024, 768, 192), (777, 256, 256), (2048, 1024, 448), (4352, 3072, 3072), (224, 256, 128), (4480, 512, 2048)])
def test_gemm_v4_equals_v3(dev, M, N, K, epi, mf):
    """gemm256v4.hip (one wave per SIMD, asm body: scripts/gen_gemm256v4.py) against gemm256v3.hip: same MFMA, same K order per output
    element, same staged bf16 image and read-back -- bit-identical outputs.  One to 48 K-tiles (1, 2: the peeled bodies only; 3: one
    pass of the steady-state loop; 7: odd count), ragged M (the last row tile takes the CUT tail path: tile-uniform maps with a row limit),
    256- and 224-row tiles (gemm_mf 8 / 7: both kernels have both), every fused epilogue; and against the oracle."""
    from diffusionkit_amd import ops
    x, w, b = randn(M, K, seed=70), randn(N, K, seed=71, scale=0.05), randn(N, seed=72, scale=0.1)
    res, gate = randn(M, N, seed=73), randn(1, N, seed=74)
    acc = bf16r(x @ w.t() + b)
    kw = {}
    if epi == "gelu":
        kw, ref = dict(epilogue=ops.DK_EPI_BIAS_GELU), om.gelu_erf(acc, Prec())
    elif epi == "gate_res":
        kw, ref = dict(epilogue=ops.DK_EPI_GATE_RES, gate=g(gate, dev), res=g(res, dev), gate_seg_len=M), res + bf16r(gate * acc)
    else:
        ref = acc
    outs = {}
    for mode in (9, 10):
        try:
            ops.tune("gemm", mode)
            ops.tune("gemm_mf", mf)
            ops.tune("gemm_split", 0)
            outs[mode] = ops.linear(g(x, dev), g(w, dev), g(b, dev), **kw)
        finally:
            ops.tune("gemm", -1)
            ops.tune("gemm_mf", -1)
            ops.tune("gemm_split", -1)
    assert rel_l2(ref, outs[10].float()) < TOL_SINGLE_OP
    assert torch.equal(outs[9], outs[10])


# ---- conv ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,C,O,ups,res", [(1, 16, 16, 64, 128, False, False), (2, 8, 24, 128, 64, False, True),
                                               (1, 16, 8, 64, 64, True, False), (1, 32, 32, 128, 3, False, False)])
def test_conv3x3(dev, B, H, W, C, O, ups, res):
    from diffusionkit_amd import ops
    x = randn(B, H, W, C, seed=20)
    w = randn(O, 3, 3, C, seed=21, scale=0.05)
    b = randn(O, seed=22, scale=0.1)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    r = randn(B, Ho, Wo, O, seed=23) if res else None
    y = ops.conv3x3(g(x, dev), g(w, dev), g(b, dev), upsample=ups, res=g(r, dev) if res else None)
    xin = ov.upsample_nearest(x) if ups else x
    ref = ov.conv2d_nhwc(xin, w, b, Prec())
    if res:
        ref = ref + r
    assert y.shape == ref.shape
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP


@pytest.mark.parametrize("B,H,W,C,O,ups,res,mode", [(1, 40, 24, 64, 256, False, False, 9),    # ragged M (960 pixels), forced
                                                    (2, 16, 24, 128, 256, False, True, 9),   # two images: no tap crosses an image border
                                                    (1, 12, 16, 64, 512, True, False, 9),    # nearest-x2 view folded into the gather
                                                    (1, 224, 224, 64, 256, False, True, -1),  # 196 tiles: the automatic choice
                                                    (1, 112, 112, 64, 256, True, False, -1)])
@pytest.mark.parametrize("mf", [8, 7])
def test_conv3x3_on_256_tile_kernel(dev, B, H, W, C, O, ups, res, mode, mf):
    """Convolutions with O % 256 == 0 on dk_gemm256v3_kernel<MF, CONV = true> (im2col rows recomputed per DMA piece, padding taps
    through out-of-range buffer offsets): against the oracle and against the 128^2-tile kernel (same products, other summation order)."""
    from diffusionkit_amd import ops
    x = randn(B, H, W, C, seed=20)
    w = randn(O, 3, 3, C, seed=21, scale=0.05)
    b = randn(O, seed=22, scale=0.1)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    r = randn(B, Ho, Wo, O, seed=23) if res else None
    args = (g(x, dev), g(w, dev), g(b, dev))
    kw = dict(upsample=ups, res=g(r, dev) if res else None)
    try:
        ops.tune("gemm", mode)
        ops.tune("gemm_mf", mf)
        y = ops.conv3x3(*args, **kw)
        ops.tune("gemm", 128)
        y128 = ops.conv3x3(*args, **kw)
    finally:
        ops.tune("gemm", -1)
        ops.tune("gemm_mf", -1)
    xin = ov.upsample_nearest(x) if ups else x
    ref = ov.conv2d_nhwc(xin, w, b, Prec())
    if res:
        ref = ref + r
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP
    assert max_abs(y128.float(), y.float()) <= 0.02 * float(ref.abs().max()) + 1e-2
    assert float((y128.float() != y.float()).float().mean()) < 0.05


@pytest.mark.parametrize("M,N,K,epi", [(1024, 768, 640, "gate_res"),     # 12 tiles: 4 equal pieces per tile
                                       (1178, 512, 384, "bias"),         # ragged M, 10 tiles
                                       (3328, 2560, 256, "gelu"),        # 130 tiles > half the CUs: finisher + 1 producer
                                       (4352, 3072, 512, "gate_res")])   # 204 tiles, 4 producer pieces in turn per spare CU
@pytest.mark.parametrize("mf", [8, 7])
def test_gemm_v3_remainder_split(dev, M, N, K, epi, mf):
    """dk_tune_set("gemm_split", 1): the tiles beyond the last full wave of the CUs are cut along K into a
    finisher piece and producer pieces (fp32 slabs + flags in the caller's workspace).  Same results as
    the unsplit kernel up to the fp32 summation order; the flag region must be left zero."""
    from diffusionkit_amd import ops
    x, w, b = randn(M, K, seed=50), randn(N, K, seed=51, scale=0.05), randn(N, seed=52, scale=0.1)
    res, gate = randn(M, N, seed=53), randn(1, N, seed=54)
    acc = bf16r(x @ w.t() + b)
    ws = ops.gemm_workspace(dev)
    kw = {}
    if epi == "gelu":
        kw, ref = dict(epilogue=ops.DK_EPI_BIAS_GELU), om.gelu_erf(acc, Prec())
    elif epi == "gate_res":
        kw, ref = dict(epilogue=ops.DK_EPI_GATE_RES, gate=g(gate, dev), res=g(res, dev), gate_seg_len=M), res + bf16r(gate * acc)
    else:
        ref = acc
    try:
        ops.tune("gemm", 9)
        ops.tune("gemm_mf", mf)
        ops.tune("gemm_split", 1)
        y = ops.linear(g(x, dev), g(w, dev), g(b, dev), workspace=ws, **kw)
        ops.tune("gemm_split", 0)
        y0 = ops.linear(g(x, dev), g(w, dev), g(b, dev), workspace=ws, **kw)
    finally:
        ops.tune("gemm", -1)
        ops.tune("gemm_split", -1)
        ops.tune("gemm_mf", -1)
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP
    assert max_abs(y0.float(), y.float()) <= 0.02 * float(ref.abs().max()) + 1e-2  # a bf16 ulp where the summation order differs
    assert float((y0.float() != y.float()).float().mean()) < 0.02
    assert int(ws[-4096:].sum()) == 0


@pytest.mark.parametrize("M,N,K,epi,split", [(1280, 3072, 12288, "gate_res", True),   # FLUX 512 x 512 fc2: 60 tiles of 256 rows, four K ranges each
                                              (1280, 3072, 15360, "gate_res", True),   # ... linear2
                                              (2560, 3072, 12288, "bias", True),       # 768 x 768: 120 tiles, two K ranges
                                              (1280, 3072, 3072, "gate_res", True),    # o_proj at 512 x 512: 36 K-tile steps saved per workgroup, just above the break-even (32)
                                              (2560, 3072, 3072, "gate_res", False),   # ... at 768 x 768 (two ranges: 24 saved): below it, stays whole
                                              (4352, 3072, 12288, "gate_res", False)])  # 1024 x 1024: a full round, untouched
def test_gemm_small_launch_is_split_automatically(dev, M, N, K, epi, split):
    """Round 6 (the reference CLI's 512 x 512 default, generate_images.py:15-30): a block Linear of at most half a round of 256 x 256 tiles is cut
    along K by the AUTOMATIC choice (gemm.hip: dk_use_v4 -> gemm256v3.hip's split) when the caller hands in the split workspace, as the engines do.
    Against the fp32 oracle, and against the same launch with the split switched off: bit-identical where the rule leaves the launch whole,
    equal up to the fp32 summation order where it is cut."""
    from diffusionkit_amd import ops
    x, w, b = randn(M, K, seed=60), randn(N, K, seed=61, scale=0.02), randn(N, seed=62, scale=0.1)
    res, gate = randn(M, N, seed=63), randn(1, N, seed=64)
    acc = bf16r(x @ w.t() + b)
    ws = ops.gemm_workspace(dev)
    if epi == "gate_res":
        kw, ref = dict(epilogue=ops.DK_EPI_GATE_RES, gate=g(gate, dev), res=g(res, dev), gate_seg_len=M), res + bf16r(gate * acc)
    else:
        kw, ref = {}, acc
    y = ops.linear(g(x, dev), g(w, dev), g(b, dev), workspace=ws, **kw)
    try:
        ops.tune("gemm_split", 0)
        y0 = ops.linear(g(x, dev), g(w, dev), g(b, dev), workspace=ws, **kw)
    finally:
        ops.tune("gemm_split", -1)
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP
    assert rel_l2(ref, y0.float()) < TOL_SINGLE_OP
    if split:
        assert not torch.equal(y0, y), "the launch was expected to be cut along K (other summation order)"
        assert float((y0.float() != y.float()).float().mean()) < 0.05
    else:
        assert torch.equal(y0, y)
    assert int(ws[-4096:].sum()) == 0  # the flag region is left zero


@pytest.mark.parametrize("B,H,W,C,O", [(1, 16, 16, 64, 128), (2, 8, 24, 128, 64), (1, 64, 32, 64, 64)])
def test_conv3x3_stride2_downsample(dev, B, H, W, C, O):
    """EncoderDecoderBlock2D downsample (vae.py:141-143): pad bottom / right by one, conv k3 s2 p0."""
    from diffusionkit_amd import ops
    x = randn(B, H, W, C, seed=24)
    w = randn(O, 3, 3, C, seed=25, scale=0.05)
    b = randn(O, seed=26, scale=0.1)
    y = ops.conv3x3(g(x, dev), g(w, dev), g(b, dev), downsample=True)
    ref = ov.conv2d_s2_pad_br_nhwc(x, w, b, Prec())
    assert y.shape == ref.shape == (B, H // 2, W // 2, O)
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP


def test_latent_sample(dev):
    """encode_image_to_latents tail (__init__.py:588-594) incl. the logvar clip."""
    from diffusionkit_amd.config import tiny_vae_encoder
    from diffusionkit_amd.engine import VAEEncoderEngine
    from diffusionkit_amd.weights import pack_vae, synth_vae_encoder_weights
    cfg = tiny_vae_encoder()
    eng = VAEEncoderEngine(cfg, pack_vae(cfg, synth_vae_encoder_weights(cfg), dev))
    mom = randn(2, 4, 6, 32, seed=27, scale=1.0)
    mom[0, 0, 0, 16] = 1000.0   # logvar clipped to 20
    mom[0, 0, 1, 17] = -1000.0  # logvar clipped to -30
    noise = randn(2, 4, 6, 16, seed=28)
    got = eng.sample(g(mom, dev), noise.to(dev))
    ref = ov.sample_latent(bf16r(mom), noise)
    assert got.dtype == torch.float32 and got.shape == ref.shape
    assert torch.allclose(got.cpu(), ref, rtol=1e-5, atol=1e-6)


# ---- attention ----------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,S,D", [(1, 2, 128, 128), (1, 3, 200, 128), (2, 4, 333, 64), (1, 24, 1088, 128),
                                     (2, 24, 589 + 64, 64),
                                     # SD3-medium 1024^2 (BASELINE configs[2]): B 2, S = 4096 + 589 = 4685 -- 73.2 key tiles, ragged last query block
                                     (2, 4, 4096 + 589, 64),
                                     # the automatic choice at D = 128, S >= 2048 (phase-alternating kernel): ragged last key tile
                                     # and last query block, two images; a whole number of both
                                     (2, 3, 2048 + 17, 128), (1, 2, 2304, 128)])
def test_attention(dev, B, H, S, D):
    from diffusionkit_amd import ops
    h = H * D
    qkv = randn(B, S, 3 * h, seed=30)
    y = ops.attention(g(qkv, dev), H, D)
    q, k, v = (qkv[..., i * h:(i + 1) * h].reshape(B, S, H, D).transpose(1, 2) for i in range(3))
    ref = om.sdpa(q, k, v, 1.0 / math.sqrt(D), Prec()).transpose(1, 2).reshape(B, S, h)
    # P is rounded to bf16 before the PV product and the output once more
    assert rel_l2(ref, y.float()) < 6e-3
    assert max_abs(ref, y.float()) < 0.03


@pytest.mark.parametrize("mode", [4, 9])
@pytest.mark.parametrize("B,H,S,D", [(1, 2, 333, 128), (2, 3, 700, 64), (1, 2, 64, 128), (1, 2, 100, 64), (1, 2, 128, 128), (1, 2, 129, 128), (1, 2, 192, 128), (1, 2, 250, 128), (1, 3, 1088, 128),
                                     (2, 2, 589 + 64, 64), (1, 2, 64, 64), (1, 2, 128, 64), (1, 2, 129, 64), (1, 2, 192, 64), (1, 3, 1088 + 31, 64)])
def test_attention_kernel_variants(dev, mode, B, H, S, D):
    """The attention kernels behind dk_tune_set("attn", mode) (4: the VALU-lean kernel with the deferred rescale, 9: the
    phase-alternating kernel, D = 128; for D = 64 it falls back to the lean kernel) against
    the oracle; ragged tail tile, one to 18 key tiles (the phase-alternating kernel's two wave groups stage different tiles)."""
    from diffusionkit_amd import ops
    h = H * D
    qkv = randn(B, S, 3 * h, seed=32)
    try:
        ops.tune("attn", mode)
        y = ops.attention(g(qkv, dev), H, D)
    finally:
        ops.tune("attn", -1)
    q, k, v = (qkv[..., i * h:(i + 1) * h].reshape(B, S, H, D).transpose(1, 2) for i in range(3))
    ref = om.sdpa(q, k, v, 1.0 / math.sqrt(D), Prec()).transpose(1, 2).reshape(B, S, h)
    assert rel_l2(ref, y.float()) < 6e-3
    assert max_abs(ref, y.float()) < 0.03


@pytest.mark.parametrize("B,H,S", [(1, 2, 768), (2, 3, 1024), (1, 2, 2304), (1, 24, 4352)])
def test_attention5_one_wave_per_simd(dev, B, H, S):
    """attention5.hip (dk_tune_set("attn", 10): one wave per SIMD, 4 waves x 64 queries, K / V by LDS-DMA, generated asm tile loop; the
    automatic choice at D = 128, S >= 2048, S % 256 == 0) against the oracle and against the phase-alternating kernel: 12 key tiles (one
    pass of the four-tile loop), 16, 36, and FLUX.1-schnell's 4352 tokens x 24 heads (408 query blocks: two rounds of the CUs)."""
    from diffusionkit_amd import ops
    D, h = 128, H * 128
    qkv = randn(B, S, 3 * h, seed=33)
    outs = {}
    for mode in (9, 10):
        try:
            ops.tune("attn", mode)
            outs[mode] = ops.attention(g(qkv, dev), H, D)
        finally:
            ops.tune("attn", -1)
    hs = min(H, 3)  # (the oracle over the first heads; every head against the other kernel)
    q, k, v = (qkv[..., i * h:i * h + hs * D].reshape(B, S, hs, D).transpose(1, 2) for i in range(3))
    ref = om.sdpa(q, k, v, 1.0 / math.sqrt(D), Prec()).transpose(1, 2).reshape(B, S, hs * D)
    assert rel_l2(ref, outs[10][..., :hs * D].float()) < 6e-3
    assert max_abs(ref, outs[10][..., :hs * D].float()) < 0.03
    assert max_abs(outs[9].float(), outs[10].float()) < 4e-3  # (same algorithm; the row sums are added in another order)


@pytest.mark.parametrize("split", [2, 3, 4])
def test_attention5_key_split_of_the_last_round(dev, split):
    """attention5.hip's key-split jobs (dk_tune_set("attn_split", n): the query blocks of a launch's last, partial round of the CUs in n key
    ranges each, partial results through the workspace, dk_attn5_merge_kernel) against the unsplit launch and the oracle; a spiked key in the
    last range (the ranges end at different exponent offsets: the merge weights l_i 2^(mc_i - max mc))."""
    from diffusionkit_amd import ops
    B, H, S, D = 1, 2, 3072, 128
    h = H * D
    qkv = randn(B, S, 3 * h, seed=34, scale=0.7)
    qkv[0, 3000, h:h + D] = bf16r(qkv[0, 77, :D] * 5.0)  # key 3000 aligned with query 77 of head 0
    outs = {}
    try:
        ops.tune("attn", 10)
        for s_ in (0, split):
            ops.tune("attn_split", s_)
            outs[s_] = ops.attention(g(qkv, dev), H, D)
    finally:
        ops.tune("attn", -1)
        ops.tune("attn_split", -1)
    q, k, v = (qkv[..., i * h:(i + 1) * h].reshape(B, S, H, D).transpose(1, 2) for i in range(3))
    ref = om.sdpa(q, k, v, 1.0 / math.sqrt(D), Prec()).transpose(1, 2).reshape(B, S, h)
    assert rel_l2(ref, outs[split].float()) < 6e-3
    assert not torch.equal(outs[0], outs[split])  # (the switch does select the split launch: the partials are rounded to bf16)
    assert max_abs(outs[0].float(), outs[split].float()) < 0.02 * float(ref.abs().max()) + 4e-3


def test_attention5_batch_consistency(dev):
    """whether a query block goes through the key-split jobs (bf16-rounded partial results + merge) depends on the launch's last round of
    the CUs, hence on the batch: FLUX's 24 heads x 4352 tokens are 408 blocks at B = 1 (last round 152 blocks: not split) and 1632 at B = 4 (last
    round 96 blocks: two key ranges each).  An image's attention output is therefore NOT bit-identical between batch sizes -- pinned here: every
    image of the batch equals its own single-image launch to the rounding of the partials, and the images the last round does not reach bit for bit."""
    from diffusionkit_amd import ops
    B, H, S, D = 4, 24, 4352, 128
    h = H * D
    qkv = g(randn(B, S, 3 * h, seed=36), dev)
    y4 = ops.attention(qkv, H, D)
    worst, exact = 0.0, 0
    for i in range(B):
        y1 = ops.attention(qkv[i:i + 1].contiguous(), H, D)
        exact += int(torch.equal(y1[0], y4[i]))
        worst = max(worst, float((y1[0].float() - y4[i].float()).abs().max()))
    print(f"[attention5] B = 4 against four B = 1 launches: {exact} of 4 images bit-identical, worst |diff| {worst:.3e}")
    assert exact >= 3  # (the 96 blocks of the last round are the last heads of the last image)
    assert worst <= 4e-3  # measured 9.8e-4: bf16 roundings of the two partial results of values of magnitude < 0.25


def test_attention5_spiked_key_forces_rescale(dev):
    """a key that dominates late in the sequence: attention5.hip records the rescale factor at the decision and applies it once the previous
    tile's P.V is complete (guide rule 26); fp64 reference; a second spike in the very last tile"""
    from diffusionkit_amd import ops
    B, H, S, D = 1, 1, 1024, 128
    qkv = randn(B, S, 3 * D, seed=31, scale=0.5)
    qkv[0, 700, D:2 * D] = bf16r(qkv[0, 7, :D] * 6.0)
    qkv[0, 1023, D:2 * D] = bf16r(qkv[0, 300, :D] * 6.0)
    q, k, v = (qkv[..., i * D:(i + 1) * D].double() for i in range(3))
    p = torch.softmax(q[0] @ k[0].t() / math.sqrt(D), dim=-1)
    ref = (p @ v[0])[None]
    assert float(p[7, 700]) > 0.9 and float(p[300, 1023]) > 0.9
    try:
        ops.tune("attn", 10)
        y = ops.attention(g(qkv, dev), H, D)
    finally:
        ops.tune("attn", -1)
    assert rel_l2(ref, y.float()) < 6e-3


@pytest.mark.parametrize("B,T", [(1, 64), (2, 96), (1, 100), (1, 1000), (2, 4096)])
def test_attention_d512_flash(dev, B, T):
    """the VAE mid block's single-head D = 512 attention (vae.py:28-57) on the role-split flash kernel (attention512.hip): ragged
    query blocks and key tiles, two images, against the oracle's SDPA"""
    from diffusionkit_amd import ops
    D = 512
    q, k, v = (randn(B, T, D, seed=s, scale=sc) for s, sc in ((34, 1.0), (35, 1.0), (36, 1.0)))
    y = ops.attention_d512(g(q, dev), g(k, dev), g(v, dev))
    ref = om.sdpa(q[:, None], k[:, None], v[:, None], 1.0 / math.sqrt(D), Prec())[:, 0]
    assert rel_l2(ref, y.float()) < 6e-3
    assert max_abs(ref, y.float()) < 0.03


def test_attention_d512_spiked_key(dev):
    """a key that dominates late forces the deferred rescale across the S -> PV hand-off (guide rule 26); fp64 reference"""
    from diffusionkit_amd import ops
    T, D = 320, 512
    q, k, v = (randn(1, T, D, seed=s, scale=0.5) for s in (37, 38, 39))
    k[0, 250] = bf16r(q[0, 7] * 6.0)  # key 250 aligned with query 7
    p = torch.softmax(q[0].double() @ k[0].double().t() / math.sqrt(D), dim=-1)
    ref = (p @ v[0].double())[None]
    assert float(p[7, 250]) > 0.9
    y = ops.attention_d512(g(q, dev), g(k, dev), g(v, dev))
    assert rel_l2(ref, y.float()) < 6e-3


def test_attention_spiked_key_forces_rescale(dev):
    """A key that dominates late in the sequence forces the online-softmax rescale path
    (guide rule 26); fp64 reference."""
    from diffusionkit_amd import ops
    B, H, S, D = 1, 1, 320, 128
    h = H * D
    qkv = randn(B, S, 3 * h, seed=31, scale=0.5)
    qkv[0, 250, h:2 * h] = bf16r(qkv[0, 7, :h] * 6.0)  # key 250 aligned with query 7
    q, k, v = (qkv[..., i * h:(i + 1) * h].double() for i in range(3))
    p = torch.softmax(q[0] @ k[0].t() / math.sqrt(D), dim=-1)
    ref = (p @ v[0])[None]
    assert float(p[7, 250]) > 0.9
    for mode in (4, 9):  # the deferred-rescale kernels (threshold path; 9: phase-alternating kernel)
        try:
            ops.tune("attn", mode)
            y = ops.attention(g(qkv, dev), H, D)
        finally:
            ops.tune("attn", -1)
        assert rel_l2(ref, y.float()) < 6e-3, mode


@pytest.mark.parametrize("mode", [4])
def test_attention_spiked_key_forces_rescale_d64(dev, mode):
    """The same at D = 64 (two heads, the spike in head 0 only); fp64 reference."""
    from diffusionkit_amd import ops
    B, H, S, D = 1, 2, 400, 64
    h = H * D
    qkv = randn(B, S, 3 * h, seed=33, scale=0.5)
    qkv[0, 330, h:h + D] = bf16r(qkv[0, 9, :D] * 8.0)  # head 0: key 330 aligned with query 9
    q, k, v = (qkv[..., i * h:(i + 1) * h].double().reshape(S, H, D).transpose(0, 1) for i in range(3))
    p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(D), dim=-1)
    ref = (p @ v).transpose(0, 1).reshape(1, S, h)
    assert float(p[0, 9, 330]) > 0.9
    try:
        ops.tune("attn", mode)
        y = ops.attention(g(qkv, dev), H, D)
    finally:
        ops.tune("attn", -1)
    assert rel_l2(ref, y.float()) < 6e-3


# ---- normalisation / elementwise ------------------------------------------------------------------
@pytest.mark.parametrize("B,S,h", [(1, 100, 256), (2, 77, 1536), (1, 300, 3072), (1, 5, 2432)])
def test_ln_modulate(dev, B, S, h):
    from diffusionkit_amd import ops
    x, shift, scale = randn(B, S, h, seed=40, scale=3.0) + 0.5, randn(B, h, seed=41), randn(B, h, seed=42, scale=0.5)
    x = bf16r(x)
    y = ops.ln_modulate(g(x, dev), g(shift, dev), g(scale, dev))
    P = Prec(BF)
    ref = P.r(om.layer_norm(x, 1e-6) * P.r(1.0 + scale[:, None]) + shift[:, None])  # fused batch-1 form
    assert rel_l2(ref, y.float()) < 2e-3


@pytest.mark.parametrize("D,norm,rope", [(128, True, True), (64, True, False), (128, False, True)])
def test_qk_norm_rope(dev, D, norm, rope):
    from diffusionkit_amd import ops
    from diffusionkit_amd.config import FLUX_SCHNELL
    B, H, S_t, gh, gw = 2, 3, 5, 4, 6
    S = S_t + gh * gw
    h = H * D
    qkv = randn(B, S, 3 * h, seed=50)
    qw, kw = bf16r(1 + randn(D, seed=51, scale=0.1)), bf16r(1 + randn(D, seed=52, scale=0.1))
    tab_dev = ops.rope_table(S_t, gh, gw, (16, 56, 56), 10000.0, dev) if rope else None
    d = g(qkv, dev)
    ops.qk_norm_rope_(d, H, D, g(qw, dev) if norm else None, g(kw, dev) if norm else None, tab_dev)
    P = Prec(BF)
    q, k, v = (qkv[..., i * h:(i + 1) * h].reshape(B, S, H, D).transpose(1, 2) for i in range(3))
    if norm:
        q, k = om.rms_norm(q, qw, 1e-6, P), om.rms_norm(k, kw, 1e-6, P)
    if rope:
        tab = om.rope_table(FLUX_SCHNELL, S_t, gh, gw)
        assert max_abs(tab, tab_dev) < 2e-5  # device table vs oracle table
        q, k = om.rope_apply(q, tab, P), om.rope_apply(k, tab, P)
    ref = torch.cat([t.transpose(1, 2).reshape(B, S, h) for t in (q, k, v)], dim=-1)
    got = d.float().cpu()
    assert torch.equal(got[..., 2 * h:], qkv[..., 2 * h:])  # v untouched
    assert rel_l2(ref, got) < 2e-3


def test_timestep_embedding(dev):
    from diffusionkit_amd import ops
    from diffusionkit_amd.config import FLUX_SCHNELL, SD3_2b
    for cfg, code, dt in ((FLUX_SCHNELL, 0, BF), (SD3_2b, 1, torch.float16)):
        t = torch.tensor([1000.0, 752.0, 500.0, 250.0, 8.9296875, 0.0])
        y = ops.timestep_embedding(t.to(dev), 256, 10000.0, code)
        ref = bf16r(om.timestep_embedding(t, cfg, Prec(dt)))  # the engine stores the embedding as bf16
        # cos/sin of large bf16-rounded arguments: device and host libm may differ by 1 ulp of the
        # low-precision output in a few entries
        diff = (ref - y.float().cpu()).abs()
        assert float(diff.max()) < 1.6e-2
        assert float((diff > 1e-6).float().mean()) < 0.05


@pytest.mark.parametrize("C,G,HW,silu", [(64, 32, (8, 8), True), (128, 32, (16, 16), False), (512, 32, (8, 8), True),
                                         (256, 32, (32, 32), True)])
def test_groupnorm(dev, C, G, HW, silu):
    from diffusionkit_amd import ops
    B = 2
    x = bf16r(randn(B, HW[0], HW[1], C, seed=60, scale=2.0) + 0.7)
    gamma, beta = bf16r(1 + randn(C, seed=61, scale=0.1)), randn(C, seed=62, scale=0.1)
    y = ops.groupnorm(g(x, dev), g(gamma, dev), g(beta, dev), G, 1e-5, silu)
    P = Prec(BF)
    ref = ov.group_norm_nhwc(x, gamma, beta, G, 1e-5, P)
    if silu:
        ref = ov.silu(ref, P)
    assert rel_l2(ref, y.float()) < 3e-3


@pytest.mark.parametrize("B,H,W,C,O,res,sc,gn", [(1, 16, 16, 64, 128, False, 0, True),     # one tile: every border is padding
                                                 (2, 32, 48, 128, 128, True, 0, True),     # 12 tiles, two 64-channel chunks, residual
                                                 (1, 48, 32, 256, 128, False, 256, True),  # resnet with shortcut: 4 + 4 chunks
                                                 (1, 32, 32, 64, 256, True, 0, False),     # plain conv (no table), two N-tiles
                                                 (1, 64, 64, 128, 128, False, 128, True)])
def test_conv3x3_gn_halo(dev, B, H, W, C, O, res, sc, gn):
    """norm -> silu -> conv (+ residual / + 1x1 shortcut as extra reduction columns) in one launch (conv_halo.hip): against the
    oracle's GroupNorm + SiLU + conv, against the unfused device path (dk_groupnorm_bf16 then dk_conv3x3_bf16: same arithmetic in
    front of the MFMAs, other summation order), and the output statistics against a statistics pass over the stored output."""
    from diffusionkit_amd import ops
    G, eps = 32, 1e-5
    x = bf16r(randn(B, H, W, C, seed=80, scale=1.5) + 0.3)
    gamma, beta = bf16r(1 + randn(C, seed=81, scale=0.1)), randn(C, seed=82, scale=0.1)
    w = randn(O, 3, 3, C, seed=83, scale=0.05)
    b = randn(O, seed=84, scale=0.1)
    r = randn(B, H, W, O, seed=85) if res else None
    x2 = randn(B, H, W, sc, seed=86) if sc else None
    ws = randn(O, sc, seed=87, scale=0.05) if sc else None
    bs = randn(O, seed=88, scale=0.1) if sc else None
    P = Prec(BF)
    act = ov.silu(ov.group_norm_nhwc(x, gamma, beta, G, eps, P), P) if gn else x
    ref = ov.conv2d_nhwc(act, w, b, Prec())
    if res:
        ref = ref + r
    if sc:
        ref = ref + (x2 @ ws.t() + bs)
    xd = g(x, dev)
    tab = ops.groupnorm_table(xd, g(gamma, dev), g(beta, dev), G, eps) if gn else None
    wk = w.reshape(O, -1)
    if sc:
        wk = torch.cat([wk, ws], dim=1)
    y, part = ops.conv3x3_gn(xd, g(wk, dev), g(b, dev), gn_table=tab, silu=True, res=g(r, dev) if res else None,
                             x2=g(x2, dev) if sc else None, bias2=g(bs, dev) if sc else None, stats_groups=G)
    assert y.shape == ref.shape
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP
    # the unfused device path
    actd = ops.groupnorm(xd, g(gamma, dev), g(beta, dev), G, eps, True) if gn else xd
    if sc:
        scd = ops.linear(g(x2, dev).reshape(-1, sc), g(ws, dev), g(bs, dev)).reshape(B, H, W, O)
        y0 = ops.conv3x3(actd, g(w, dev), g(b, dev), res=scd)
    else:
        y0 = ops.conv3x3(actd, g(w, dev), g(b, dev), res=g(r, dev) if res else None)
    assert max_abs(y0.float(), y.float()) <= 0.02 * float(ref.abs().max()) + 1e-2
    # output statistics: the table built from the conv's partials == the table of a statistics pass over y
    g2, b2 = bf16r(1 + randn(O, seed=89, scale=0.1)), randn(O, seed=90, scale=0.1)
    t_part = ops.groupnorm_table(None, g(g2, dev), g(b2, dev), G, eps, partials=part, shape=(B, H * W, O))
    t_pass = ops.groupnorm_table(y.contiguous(), g(g2, dev), g(b2, dev), G, eps)
    assert rel_l2(t_pass.cpu(), t_part.cpu()) < 1e-5


@pytest.mark.parametrize("B,H,W,C,O,res,gn,ups", [(1, 16, 16, 128, 256, False, True, False),   # one tile: every border is padding; one pass of the chunk loop
                                                   (2, 32, 48, 128, 256, True, True, False),    # 12 tiles, two images, residual
                                                   (1, 48, 48, 256, 512, False, True, False),   # an inner tile, two column tiles, four chunks
                                                   (1, 32, 32, 512, 256, True, True, False),    # eight chunks
                                                   (2, 32, 48, 128, 256, False, False, True),   # plain input through the nearest-x2 view
                                                   (1, 64, 32, 256, 256, True, False, False),   # plain input, residual
                                                   (2, 32, 48, 128, 128, True, True, False),    # 128-channel tiles (waves 4 x 1, three-slot weight ring)
                                                   (1, 48, 48, 256, 128, False, True, False),   # ... four chunks, an inner tile
                                                   (1, 32, 32, 256, 128, False, False, True)])  # ... plain input through the nearest-x2 view
def test_conv256v4_equals_conv_halo(dev, B, H, W, C, O, res, gn, ups):
    """conv256v4.hip (one wave per SIMD, 16 x 16 pixels x 256 channels per workgroup, asm body: scripts/gen_conv256v4.py) against conv_halo.hip
    on the same inputs: same products, same chunk-major fp32 order, same rounding points (GroupNorm-apply -> bf16 -> SiLU -> bf16 on the way
    into LDS; bf16(acc + bias) + residual on the way out) -- bit-identical outputs; the GroupNorm partials are sums in another order
    (1e-6); and against the oracle's GroupNorm + SiLU + conv."""
    from diffusionkit_amd import ops
    G, eps = 32, 1e-5
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    x = bf16r(randn(B, Hs, Ws, C, seed=180, scale=1.5) + 0.3)
    gamma, beta = bf16r(1 + randn(C, seed=181, scale=0.1)), randn(C, seed=182, scale=0.1)
    w = randn(O, 3, 3, C, seed=183, scale=0.05)
    b = randn(O, seed=184, scale=0.1)
    r = randn(B, H, W, O, seed=185) if res else None
    P = Prec(BF)
    act = ov.silu(ov.group_norm_nhwc(x, gamma, beta, G, eps, P), P) if gn else x
    ref = ov.conv2d_nhwc(ov.upsample_nearest(act) if ups else act, w, b, Prec())
    if res:
        ref = ref + r
    xd = g(x, dev)
    tab = ops.groupnorm_table(xd, g(gamma, dev), g(beta, dev), G, eps) if gn else None
    sg = 0 if ups else G
    outs = {}
    for mode in (0, 2):
        try:
            ops.tune("conv_v4", mode)
            outs[mode] = ops.conv3x3_gn(xd, g(w, dev).reshape(O, -1), g(b, dev), gn_table=tab, silu=True, res=g(r, dev) if res else None, stats_groups=sg,
                                        upsample=ups)
        finally:
            ops.tune("conv_v4", 1)
    y0, y1 = (outs[0][0], outs[2][0]) if sg else (outs[0], outs[2])
    assert rel_l2(ref, y1.float()) < TOL_SINGLE_OP
    assert torch.equal(y0, y1)
    if sg:
        assert rel_l2(outs[0][1].double().cpu(), outs[2][1].double().cpu()) < 1e-6


def test_conv3x3_halo_upsample(dev):
    """the upsampling conv (vae.py:20-25,146) on the halo kernel: the nearest-x2 view folded into the halo addressing, no norm"""
    from diffusionkit_amd import ops
    B, Hs, Ws, C, O = 2, 16, 24, 128, 256
    x = randn(B, Hs, Ws, C, seed=96)
    w = randn(O, 3, 3, C, seed=97, scale=0.05)
    b = randn(O, seed=98, scale=0.1)
    ref = ov.conv2d_nhwc(ov.upsample_nearest(x), w, b, Prec())
    y = ops.conv3x3_gn(g(x, dev), g(w, dev).reshape(O, -1), g(b, dev), gn_table=None, upsample=True)
    assert y.shape == ref.shape == (B, 2 * Hs, 2 * Ws, O)
    assert rel_l2(ref, y.float()) < TOL_SINGLE_OP


def test_conv_out_image_tail_halo(dev):
    """conv_norm_out -> silu -> conv_out -> clip / uint8 (vae.py:381,384,397-399; __init__.py:581-584,525-526) in one launch"""
    from diffusionkit_amd import ops
    B, H, W, C, G, eps = 2, 32, 48, 128, 32, 1e-5
    x = bf16r(randn(B, H, W, C, seed=91, scale=1.5) + 0.3)
    gamma, beta = bf16r(1 + randn(C, seed=92, scale=0.1)), randn(C, seed=93, scale=0.1)
    w = randn(3, 3, 3, C, seed=94, scale=0.05)
    b = randn(3, seed=95, scale=0.1)
    P = Prec(BF)
    act = ov.silu(ov.group_norm_nhwc(x, gamma, beta, G, eps, P), P)
    ref = bf16r(ov.conv2d_nhwc(act, w, b, Prec()))
    xd = g(x, dev)
    tab = ops.groupnorm_table(xd, g(gamma, dev), g(beta, dev), G, eps)
    img, u8, raw = ops.conv3x3_gn(xd, g(w, dev).reshape(3, -1), g(b, dev), gn_table=tab, image=True)
    assert rel_l2(ref, raw[..., :3].float()) < TOL_SINGLE_OP
    want = torch.clip((raw[..., :3].float() / 2 + 0.5).to(BF).float(), 0, 1)  # bf16 products, as dk_image_post_kernel and the reference
    assert torch.equal(img, want) and torch.equal(u8, (want.to(BF) * 255).to(BF).to(torch.uint8))
    assert torch.all(raw[..., 3].float() == 0)


def test_softmax_and_transpose(dev):
    from diffusionkit_amd import ops
    x = randn(64, 256, seed=70, scale=3.0)
    y = ops.softmax_rows_(g(x, dev).clone())
    assert rel_l2(torch.softmax(x, -1), y.float()) < TOL_SINGLE_OP
    z = ops.transpose(g(x, dev))
    assert torch.equal(z.float().cpu(), x.t())
    # ragged row length inside a padded row (the VAE attention's score matrix for an odd token count): the padding is
    # ignored on input (NaNs there must not leak) and comes out as zeros
    buf = g(x, dev).clone()
    buf[:, 100:] = float("nan")
    y = ops.softmax_rows_(buf[:, :100])
    full = buf.float().cpu()
    assert rel_l2(torch.softmax(x[:, :100], -1), full[:, :100]) < TOL_SINGLE_OP
    assert torch.all(full[:, 100:] == 0)
    # rows longer than the register cache (latents beyond 128 x 128)
    xl = randn(3, 16384 + 4096, seed=71, scale=3.0)
    bl = g(xl, dev).clone()
    ops.softmax_rows_(bl[:, :20000])
    fl = bl.float().cpu()
    assert rel_l2(torch.softmax(bf16r(xl)[:, :20000], -1), fl[:, :20000]) < TOL_SINGLE_OP and torch.all(fl[:, 20000:] == 0)


@pytest.mark.parametrize("flux,cfg_on", [(True, False), (False, True), (True, True)])
def test_patchify_and_euler_step(dev, flux, cfg_on):
    """dk_latent_to_tokens + dk_euler_cfg_step against the oracle's patchify / unpatchify / CFG /
    Euler restatement."""
    from diffusionkit_amd import _lib
    from diffusionkit_amd.config import tiny_flux, tiny_sd3
    from diffusionkit_amd.engine import MMDiTEngine, _stream  # noqa: F401
    from oracle.mmdit import OracleMMDiT
    cfg = tiny_flux() if flux else tiny_sd3()
    lib = _lib.load()
    n_img, Hl, Wl, C, p = 2, 8, 12, 16, 2
    x = torch.randn(n_img, Hl, Wl, C, generator=torch.Generator().manual_seed(80))
    dup = 2 if cfg_on else 1
    S_i, F = (Hl // p) * (Wl // p), p * p * C
    xd = x.to(dev).contiguous()
    tok = torch.empty(n_img * dup, S_i, F, dtype=BF, device=dev)
    _lib.check(lib.dk_latent_to_tokens(xd.data_ptr(), tok.data_ptr(), n_img, dup, Hl, Wl, C, p, int(flux), _stream()))
    orc = OracleMMDiT(cfg, {"x_embedder.proj.weight": torch.eye(F).reshape(F, *((1, 1, F) if flux else (p, p, C))),
                            "x_embedder.proj.bias": torch.zeros(F)}, Prec())
    ref_tok = orc._patch_embed(bf16r(x))  # identity projection => the patch feature order itself
    assert torch.equal(tok.float().cpu()[:n_img], ref_tok)
    if cfg_on:
        assert torch.equal(tok[:n_img], tok[n_img:])
    out = randn(n_img * dup, S_i, F, seed=81)
    sigma, sigma_next, w = 0.75, 0.5, 5.0
    _lib.check(lib.dk_euler_cfg_step(xd.data_ptr(), g(out, dev).data_ptr(), F, tok.data_ptr(), n_img, int(cfg_on), Hl, Wl, C, p,
                                     int(flux), sigma, sigma_next, w, _stream()))
    xb = bf16r(x)
    o = orc._unpatch(out, Hl, Wl)
    den = xb - o[:n_img] * sigma
    if cfg_on:
        den_neg = xb - o[n_img:] * sigma
        den = den_neg + w * (den - den_neg)
    ref = x + (x - den) / sigma * (sigma_next - sigma)
    assert max_abs(ref, xd) < 1e-5
    assert torch.equal(tok.float().cpu()[:n_img], orc._patch_embed(bf16r(xd.cpu())))


def test_errors_do_not_cross_the_abi(dev):
    from diffusionkit_amd import _lib, ops
    with pytest.raises(_lib.DkHipError, match="multiple of 64"):
        ops.linear(g(randn(8, 40), dev), g(randn(8, 40), dev))
    with pytest.raises(_lib.DkHipError, match="head_dim"):
        ops.attention(g(randn(1, 8, 3 * 96), dev), 1, 96)
