"""Flow-matching sigma schedules (host-side scalars, float32 like the reference).

reference: python/src/diffusionkit/mlx/sampler.py:10-42 (ModelSamplingDiscreteFlow),
:45-77 (FluxSampler); schedule assembly python/src/diffusionkit/mlx/__init__.py:559-571.

Everything here is O(num_steps) scalar work done once per image; it is evaluated in
numpy float32 so that the values agree with the reference's MLX float32 arrays.
"""
from __future__ import annotations

import math

import numpy as np

_F = np.float32


class _FlowSamplerBase:
    #: first table index (reference sampler.py:16 uses arange(1, 1001); FluxSampler :51 arange(0, 1001))
    _TABLE_START = 1

    def __init__(self, shift: float = 1.0):
        self.shift = shift
        timesteps = 1000
        ts = self.sigma(np.arange(self._TABLE_START, timesteps + 1, 1))
        self.sigmas = ts

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma):
        return np.asarray(sigma, dtype=_F) * _F(1000)

    def sigma(self, timestep):
        timestep = np.asarray(timestep).astype(_F) / _F(1000.0)
        if self.shift == 1.0:
            return timestep
        s = _F(self.shift)
        return s * timestep / (_F(1) + (s - _F(1)) * timestep)

    def calculate_denoised(self, sigma, model_output, model_input):
        # x0 = x - v * sigma (reference sampler.py:37-39 / :72-74)
        return model_input - model_output * sigma

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        return sigma * noise + (1.0 - sigma) * latent_image


class ModelSamplingDiscreteFlow(_FlowSamplerBase):
    """SD3 schedule helper (reference sampler.py:10-42)."""
    _TABLE_START = 1


class FluxSampler(_FlowSamplerBase):
    """FLUX schedule helper (reference sampler.py:45-77)."""
    _TABLE_START = 0


def _linspace_f32(start: float, stop: float, num: int) -> np.ndarray:
    # MLX linspace: arange(num) * ((stop-start)/(num-1)) + start, in float32.
    if num == 1:
        return np.asarray([start], dtype=_F)
    step = _F((stop - start) / (num - 1))
    return (np.arange(num, dtype=_F) * step + _F(start)).astype(_F)


def get_sigmas(sampler, num_steps: int) -> np.ndarray:
    """reference mlx/__init__.py:559-571 (incl. quirk Q13: for SD3 the shift is applied
    to endpoints that already went through ``sigma()``)."""
    start = float(sampler.timestep(sampler.sigma_max))
    end = float(sampler.timestep(sampler.sigma_min))
    is_flux = isinstance(sampler, FluxSampler)
    n = num_steps + 1 if is_flux else num_steps
    timesteps = _linspace_f32(start, end, n)
    sigs = [float(sampler.sigma(t)) for t in timesteps]
    if not is_flux:
        sigs += [0.0]
    return np.asarray(sigs, dtype=_F)


def max_denoise(sampler, sigmas) -> bool:
    """reference mlx/__init__.py:576-579"""
    max_sigma = float(sampler.sigma_max)
    sigma = float(sigmas[0])
    return math.isclose(max_sigma, sigma, rel_tol=1e-05) or sigma > max_sigma
