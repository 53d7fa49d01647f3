"""CPU ORACLE (test infrastructure, NOT product code) -- schedule, CFG and Euler loop.

Restates python/src/diffusionkit/mlx/sampler.py:10-77 and
python/src/diffusionkit/mlx/__init__.py:253-292 (denoise_latents), :553-584
(get_noise/get_sigmas/get_empty_latent/decode), :674-788 (CFGDenoiser, LatentFormat,
to_d, sample_euler).  Pinned by the reference's own DiffusionPipeline.denoise_latents executed on the MLX stand-in (SD3 with
CFG, FLUX, SD3 img2img: tests/test_reference_mlx_golden.py, rel-L2 < 2e-5) and by the scalar known-answer values in
tests/golden/kat_scalars.json; MLX's low-precision arithmetic is unpinned (see oracle/mmdit.py header).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .mmdit import OracleMMDiT, Prec

Tensor = torch.Tensor


# ---- sampler.py ---------------------------------------------------------------------
def sigma_of_timestep(t: Tensor, shift: float) -> Tensor:
    """sampler.py:31-35 / :66-70 (float32)."""
    t = t.to(torch.float32) / 1000.0
    if shift == 1.0:
        return t
    return shift * t / (1 + (shift - 1) * t)


def sigma_min_max(shift: float, flux: bool):
    """sampler.py:16-26 (table from arange(1,1001)) / :51-61 (arange(0,1001))."""
    table = sigma_of_timestep(torch.arange(0 if flux else 1, 1001), shift)
    return float(table[0]), float(table[-1])


def get_sigmas(shift: float, flux: bool, num_steps: int) -> Tensor:
    """mlx/__init__.py:559-571."""
    smin, smax = sigma_min_max(shift, flux)
    start = float(torch.tensor(smax, dtype=torch.float32) * 1000)
    end = float(torch.tensor(smin, dtype=torch.float32) * 1000)
    n = num_steps + 1 if flux else num_steps
    timesteps = torch.linspace(start, end, n, dtype=torch.float32)
    sigs = [float(sigma_of_timestep(t, shift)) for t in timesteps]
    if not flux:
        sigs.append(0.0)
    return torch.tensor(sigs, dtype=torch.float32)


def get_noise(seed: int, h: int, w: int, c: int = 16) -> Tensor:
    """mlx/__init__.py:553-557: numpy global RNG, NCHW draw, transposed to NHWC, fp32."""
    np.random.seed(seed)
    noise = np.random.randn(1, c, h, w)
    return torch.from_numpy(noise).to(torch.float32).permute(0, 2, 3, 1).contiguous()


def get_empty_latent(h: int, w: int) -> Tensor:
    """mlx/__init__.py:573-574"""
    return torch.ones(1, h, w, 16) * 0.0609


LATENT_FORMAT = {  # mlx/__init__.py:736-747 (scale_factor, shift_factor)
    "sd3": (1.5305, 0.0609),
    "flux": (0.3611, 0.1159),
}


def process_in(latent: Tensor, fmt: str) -> Tensor:
    """LatentFormat.process_in (mlx/__init__.py:729-730)"""
    scale, shift = LATENT_FORMAT[fmt]
    return (latent - shift) * scale


def process_out(latent: Tensor, fmt: str) -> Tensor:
    """LatentFormat.process_out (mlx/__init__.py:732-733)"""
    scale, shift = LATENT_FORMAT[fmt]
    return latent / scale + shift


# ---- CFGDenoiser + sample_euler -----------------------------------------------------
def cfg_denoise(model: OracleMMDiT, x_t: Tensor, timestep: float, sigma: float,
                conditioning: Tensor, cfg_weight: float, act: Prec) -> Tensor:
    """CFGDenoiser.__call__ (mlx/__init__.py:691-719). x_t fp32 [1,h,w,16]."""
    if cfg_weight <= 0:
        x_in = act.r(x_t)  # cast to activation dtype (quirk Q6)
    else:
        x_in = act.r(torch.cat([x_t] * 2, dim=0))
    out = model(x_in, conditioning, timestep)
    den = x_in - out * sigma  # promoted to fp32 by the fp32 sigma (sampler.py:37-39)
    if cfg_weight <= 0:
        return den
    text, neg = den[0:1], den[1:2]
    return neg + cfg_weight * (text - neg)


def sample_euler(model: OracleMMDiT, x: Tensor, sigmas: Tensor, conditioning: Tensor,
                 pooled: Tensor, cfg_weight: float, act: Prec, trace: Optional[list] = None,
                 t_act: Optional[Prec] = None) -> Tensor:
    """sample_euler (mlx/__init__.py:761-788): x stays fp32; model timesteps are
    sigma*1000 rounded to the pipeline's activation dtype (quirk Q1; :683,770): fp16 for
    DiffusionPipeline (:76-79), bf16 for FluxPipeline (:610-613).  ``t_act`` = that rounding when it differs
    from ``act`` (the MI355X engine keeps bf16 activations for SD3 but rounds its timesteps to fp16)."""
    timesteps = (t_act or act).r(sigmas * 1000.0)
    model.cache_modulation_params(pooled, timesteps)
    for i in range(len(sigmas) - 1):
        den = cfg_denoise(model, x, float(timesteps[i]), float(sigmas[i]), conditioning, cfg_weight, act)
        d = (x - den) / sigmas[i]
        x = x + d * (sigmas[i + 1] - sigmas[i])
        if trace is not None:
            trace.append(x.clone())
    return x


def read_image_array(rgb_u8: np.ndarray) -> Tensor:
    """read_image (mlx/__init__.py:536-551) for an HWC uint8 array whose sides are multiples of 64
    (the resize branch is host-side PIL code, not restated): RGB in [-1, 1], [1,H,W,3] fp32."""
    assert rgb_u8.shape[0] % 64 == 0 and rgb_u8.shape[1] % 64 == 0
    return torch.from_numpy((rgb_u8[:, :, :3].astype(np.float32) / 255) * 2 - 1.0)[None]


def encode_image_to_latents(encoder, image: Tensor, seed: int) -> Tensor:
    """mlx/__init__.py:586-594: posterior sample with the numpy noise of ``seed`` (shape of the mean)."""
    from .vae import sample_latent
    hidden = encoder(image)
    _, h, w, c2 = hidden.shape
    return sample_latent(hidden, get_noise(seed, h, w, c2 // 2))


def denoise_latents(model: OracleMMDiT, conditioning: Tensor, pooled: Tensor, num_steps: int,
                    cfg_weight: float, latent_size, seed: int, shift: float, flux: bool,
                    act: Prec, trace: Optional[list] = None, init_latent: Optional[Tensor] = None,
                    denoise: float = 1.0, t_act: Optional[Prec] = None) -> Tensor:
    """DiffusionPipeline.denoise_latents (mlx/__init__.py:253-292).  ``init_latent`` = the output of
    encode_image_to_latents for img2img (image_path given), else the empty latent and denoise = 1."""
    fmt = "flux" if flux else "sd3"
    if init_latent is None:
        x_T = get_empty_latent(*latent_size)
        denoise = 1.0
    else:
        x_T = process_in(init_latent, fmt)
        latent_size = tuple(x_T.shape[1:3])
    noise = get_noise(seed, *latent_size)
    sigmas = get_sigmas(shift, flux, num_steps)
    sigmas = sigmas[int(num_steps * (1 - denoise)):]
    noise_scaled = sigmas[0] * noise + (1.0 - sigmas[0]) * x_T  # sampler.py:41-42
    latent = sample_euler(model, noise_scaled, sigmas, conditioning, pooled, cfg_weight, act, trace, t_act)
    return process_out(latent, "flux" if flux else "sd3")


def image_psnr(reference_u8: np.ndarray, proxy_u8: np.ndarray) -> float:
    """python/src/diffusionkit/utils.py:52-67, the metric behind the reference's 20 dB image gate
    (tests/mlx/test_diffusion_pipeline.py:91-93): the arrays stay uint8, so the difference WRAPS modulo 256 before it is squared
    (and the square again), the peak is the reference's largest value and 'mse' is an RMSE."""
    reference = np.asarray(reference_u8, dtype=np.uint8).flatten()
    proxy = np.asarray(proxy_u8, dtype=np.uint8).flatten()
    peak = np.abs(reference).max()
    rmse = np.sqrt(np.mean((reference - proxy) ** 2))
    return float(20 * np.log10((peak + 1e-5) / (rmse + 1e-10)))


def compute_psnr(reference: np.ndarray, proxy: np.ndarray) -> float:
    """python/src/diffusionkit/utils.py:70-82 (note: 'mse' there is an RMSE)."""
    reference = np.asarray(reference, dtype=np.float64).flatten()
    proxy = np.asarray(proxy, dtype=np.float64).flatten()
    peak = np.abs(reference).max()
    rmse = np.sqrt(np.mean((reference - proxy) ** 2))
    return float(20 * np.log10((peak + 1e-5) / (rmse + 1e-10)))
