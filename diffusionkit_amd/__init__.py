"""diffusionkit_amd -- MI355X-native denoising engine behind DiffusionKit's pipeline API.

Drop-in for the hot path of ``diffusionkit.mlx`` (reference:
python/src/diffusionkit/mlx/__init__.py): ``DiffusionPipeline``, ``FluxPipeline``,
``MMDIT_CKPT``, ``T5_MAX_LENGTH``, ``CFGDenoiser``, ``sample_euler``.
Importing this package does not load the HIP library; constructing a pipeline or an engine
does, and raises ``DkHipError`` if it is missing (there is no CPU fallback).
"""
from .config import MMDIT_CKPT, T5_MAX_LENGTH, MMDiTConfig, VAEDecoderConfig, SD3_2b, SD3_8b, FLUX_SCHNELL  # noqa: F401
from ._lib import DkHipError  # noqa: F401


def __getattr__(name):
    # pipeline classes import torch lazily so that `import diffusionkit_amd` stays cheap
    if name in ("DiffusionPipeline", "FluxPipeline", "CFGDenoiser", "sample_euler", "LatentFormat",
                "SD3LatentFormat", "FluxLatentFormat", "to_d", "append_dims"):
        from . import pipeline
        return getattr(pipeline, name)
    if name in ("MMDiTEngine", "VAEDecoderEngine"):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)
