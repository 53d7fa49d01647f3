"""Writes tests/golden/kat_scalars.json: scalar known-answer values derived by hand from the
reference's formulas: python/src/diffusionkit/mlx/sampler.py:31-35,
mlx/__init__.py:553-574,729-747,769-771; python/src/diffusionkit/utils.py:65-67.
Deliberately uses plain numpy float64/float32 arithmetic, independent of both
diffusionkit_amd.sampler and oracle.pipeline, which the tests then check against it."""
import json
import os

import numpy as np


def sigma(t, shift):
    t = np.float32(t) / np.float32(1000.0)
    if shift == 1.0:
        return np.float32(t)
    s = np.float32(shift)
    return np.float32(s * t / (np.float32(1) + (s - np.float32(1)) * t))


def sigmas(shift, flux, n):
    smin = sigma(0 if flux else 1, shift)
    smax = sigma(1000, shift)
    start, end = float(np.float32(smax) * np.float32(1000)), float(np.float32(smin) * np.float32(1000))
    num = n + 1 if flux else n
    ts = np.linspace(start, end, num, dtype=np.float64).astype(np.float32)
    out = [float(sigma(t, shift)) for t in ts]
    if not flux:
        out.append(0.0)
    return out


def bf16_round(x):
    import torch
    return torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).to(torch.float32).tolist()


def fp16_round(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32).tolist()


np.random.seed(0)
noise = np.random.randn(1, 16, 4, 4)
kat = {
    "flux_sigmas_n4_shift1": sigmas(1.0, True, 4),
    "flux_timesteps_bf16_n4": bf16_round([s * 1000 for s in sigmas(1.0, True, 4)]),
    "sd3_sigma_min_shift3": float(sigma(1, 3.0)),
    "sd3_sigmas_n4_shift3": sigmas(3.0, False, 4),
    "sd3_timesteps_fp16_n4": fp16_round([s * 1000 for s in sigmas(3.0, False, 4)]),
    "sd3_sigmas_n50_shift3_first6": sigmas(3.0, False, 50)[:6],
    "sd3_sigmas_n50_shift3_last3": sigmas(3.0, False, 50)[-3:],
    "noise_seed0_nchw_0_0_0_first4": noise[0, 0, 0, :4].tolist(),
    "noise_seed0_nhwc_0_0_0_first4": noise.transpose(0, 2, 3, 1)[0, 0, 0, :4].tolist(),
    "latent_format": {"sd3": [1.5305, 0.0609], "flux": [0.3611, 0.1159]},
    "empty_latent_value": 0.0609,
    "psnr_example": {"ref": [1.0, -2.0, 3.0, 0.5], "proxy": [1.1, -2.0, 2.9, 0.5],
                     "value": float(20 * np.log10((3.0 + 1e-5) / (np.sqrt(np.mean(np.array([0.1, 0, 0.1, 0]) ** 2)) + 1e-10)))},
}
with open(os.path.join(os.path.dirname(__file__), "kat_scalars.json"), "w") as f:
    json.dump(kat, f, indent=1)
print(json.dumps(kat, indent=1))
