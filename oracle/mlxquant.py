"""CPU ORACLE (test infrastructure, NOT product code) -- MLX affine group quantisation (``mx.quantize`` / ``mx.dequantize``).

The reference loads its ``*-4bit-quantized`` checkpoints by calling ``nn.quantize(model)`` (group_size 64, bits 4: MLX's
defaults) and then ``load_weights`` / ``update`` with the stored ``weight`` (uint32) / ``scales`` / ``biases`` triplets
(python/src/diffusionkit/mlx/model_io.py:728-734,772-775).  The arithmetic lives in MLX (``mlx==0.17.3``, setup.py:32), which is
not vendored under /root/reference and cannot be imported here: PARITY UNPINNED for the bit layout.  What is restated is MLX's
published contract (docs of ``mx.quantize``): a row of ``w`` is cut into groups of ``group_size`` consecutive elements; each group
stores ``scale`` and ``bias`` with ``w_i ~= scale * q_i + bias``, ``q_i`` an unsigned ``bits``-bit integer; 32 / bits consecutive
``q`` share one uint32, element ``j`` of the pack in bits ``[bits * j, bits * (j + 1))`` (least significant first).

Plain Python / numpy loops: small cases only.
"""
import numpy as np


def dequantize(wq: np.ndarray, scales: np.ndarray, biases: np.ndarray, group_size: int = 64, bits: int = 4) -> np.ndarray:
    """``mx.dequantize``: wq uint32 [out, in * bits / 32], scales / biases [out, in / group_size] -> fp32 [out, in]"""
    per = 32 // bits
    out_f, packs = wq.shape
    n_in = packs * per
    w = np.zeros((out_f, n_in), np.float32)
    for r in range(out_f):
        for c in range(n_in):
            q = (int(wq[r, c // per]) >> (bits * (c % per))) & ((1 << bits) - 1)
            g = c // group_size
            w[r, c] = np.float32(scales[r, g]) * np.float32(q) + np.float32(biases[r, g])
    return w


def quantize(w: np.ndarray, group_size: int = 64, bits: int = 4):
    """An affine quantiser of the documented form (min / max of the group): scale = (max - min) / (2^bits - 1), bias = min,
    q = round((w - bias) / scale).  Used to BUILD test checkpoints; any (q, scale, bias) triplet is a valid file, so the loader's
    parity does not depend on MLX's tie-breaking inside ``mx.quantize``."""
    per = 32 // bits
    out_f, n_in = w.shape
    assert n_in % group_size == 0 and group_size % per == 0
    levels = (1 << bits) - 1
    wq = np.zeros((out_f, n_in // per), np.uint32)
    scales = np.zeros((out_f, n_in // group_size), np.float32)
    biases = np.zeros_like(scales)
    for r in range(out_f):
        for g in range(n_in // group_size):
            grp = w[r, g * group_size:(g + 1) * group_size].astype(np.float32)
            lo, hi = float(grp.min()), float(grp.max())
            sc = (hi - lo) / levels if hi > lo else 1.0
            scales[r, g], biases[r, g] = sc, lo
            for j, v in enumerate(grp):
                q = int(min(levels, max(0, round((float(v) - lo) / sc))))
                c = g * group_size + j
                wq[r, c // per] |= np.uint32(q << (bits * (c % per)))
    return wq, scales, biases
