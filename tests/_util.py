"""Shared helpers for the parity tests (oracle = checker, HIP path = thing under test)."""
import numpy as np
import torch

BF = torch.bfloat16


def bf16r(x: torch.Tensor) -> torch.Tensor:
    """values representable in bf16, kept as fp32 (what both sides see as input)"""
    return x.to(BF).to(torch.float32)


def rel_l2(ref: torch.Tensor, got: torch.Tensor) -> float:
    ref = ref.detach().to(torch.float64).cpu().reshape(-1)
    got = got.detach().to(torch.float64).cpu().reshape(-1)
    return float(torch.linalg.norm(ref - got) / (torch.linalg.norm(ref) + 1e-30))


def max_abs(ref, got) -> float:
    return float((ref.detach().double().cpu() - got.detach().double().cpu()).abs().max())


def psnr(ref, got) -> float:
    """reference metric: python/src/diffusionkit/utils.py:70-82"""
    ref = np.asarray(ref.detach().double().cpu()).ravel()
    got = np.asarray(got.detach().double().cpu()).ravel()
    peak = np.abs(ref).max()
    rmse = np.sqrt(np.mean((ref - got) ** 2))
    return float(20 * np.log10((peak + 1e-5) / (rmse + 1e-10)))


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16r(torch.randn(*shape, generator=g) * scale)


# tolerances (stated once, used everywhere):
#  - one bf16 output rounding of an fp32-accumulated result: relative error uniform in
#    +-2^-9 => rel-L2 ~ 1.1e-3; allow 3e-3 for single-op kernels.
#  - chained ops (block / model level): compared against the fp32 oracle with the bf16-emulating
#    oracle as yardstick: err(hip, fp32) <= 2 * err(emu, fp32) + 2e-3.
TOL_SINGLE_OP = 3e-3


def seeded_checkpoint(spec, seed):
    """Deterministic synthetic checkpoint for a list of (name, shape): matrices / conv kernels ~ N(0, 1 / fan_in), vectors that
    scale (1-D '...norm...weight') ~ 1 + 0.1 N(0, 1), every other vector ~ 0.1 N(0, 1).  Shared by the fixture generator
    (tests/golden/make_reference_torch_fixtures.py) and the tests that replay its checkpoints."""
    import math
    import torch
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for name, shape in spec:
        shape = tuple(int(s) for s in shape)
        t = torch.randn(*shape, generator=g)
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if name.endswith("pos_embed"):
                t = 0.1 * t
            else:
                t = t / math.sqrt(fan_in)
        elif "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * t
        else:
            t = 0.1 * t
        out[name] = t
    return out


def checkpoint_checksum(ckpt):
    """Order-dependent scalar digest of a checkpoint's values (float64 sum of value * running index weights)."""
    import torch
    acc, n = 0.0, 0
    for name in ckpt:
        v = ckpt[name].double().flatten()
        acc += float((v * torch.arange(1, v.numel() + 1, dtype=torch.float64).remainder(997.0)).sum()) * (1 + n % 7)
        n += 1
    return acc
