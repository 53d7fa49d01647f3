"""Shape/dtype presets for the denoising hot path.

Mirrors the fields of the reference's ``MMDiTConfig`` / ``VAEDecoderConfig``
(reference: python/src/diffusionkit/mlx/config.py:19-71, 74-111, 126-132) that the
hot path reads.  Fields the reference declares but never reads
(``upcast_*_blocks``, ``low_memory_mode``) are dropped.  ``guidance_embed`` is the reference's field
(config.py:71,109): its FLUX_DEV preset sets it, but model_io.py:109,756 selects FLUX_SCHNELL for the
FLUX.1-dev checkpoint (quirk Q7) and the one call site (mmdit.py:219-220) cannot run; MODEL_CONFIG keeps that
default, and ``FLUX_DEV`` below enables the published FLUX.1-dev conditioning for callers who pass it.
``weight_dtype`` has no reference counterpart: "fp8_e4m3" runs the block Linears on the fp8 MFMA
(BASELINE.json configs[3]).
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from enum import Enum
from typing import Optional, Tuple


class PositionalEncoding(Enum):
    LearnedInputEmbedding = 1
    PreSDPARope = 2


@dataclass(frozen=True)
class MMDiTConfig:
    num_heads: int = 24
    depth_multimodal: int = 24
    depth_unified: int = 0
    parallel_mlp_for_unified_blocks: bool = True
    mlp_ratio: int = 4
    vae_latent_dim: int = 16
    layer_norm_eps: float = 1e-6
    pos_embed_type: PositionalEncoding = PositionalEncoding.LearnedInputEmbedding
    rope_axes_dim: Optional[Tuple[int, ...]] = None
    rope_theta: int = 10000
    use_qk_norm: bool = False
    hidden_size_override: Optional[int] = None
    max_latent_resolution: int = 192
    patch_size: int = 2
    patchify_via_reshape: bool = False
    pooled_text_embed_dim: int = 2048
    token_level_text_embed_dim: int = 4096
    frequency_embed_dim: int = 256
    max_period: int = 10000
    # dtype the timestep embedding is evaluated in (reference config.dtype, quirk Q2)
    dtype: str = "bfloat16"
    guidance_embed: bool = False
    # storage / MFMA dtype of the transformer blocks' Linear weights: "bfloat16", or "fp8_e4m3" (per-output-channel scales,
    # MX-fp8 activations quantised on the fly; FLUX geometry only: head_dim 128, token counts multiples of 128)
    weight_dtype: str = "bfloat16"
    # fp8 precision policy (only read with weight_dtype = "fp8_e4m3"): the Linears of the first n double-stream blocks stay bf16 --
    # an error made in the first blocks travels through all the others (measured dB per block: DESIGN.md section 4)
    fp8_bf16_double_blocks: int = 0

    @property
    def hidden_size(self) -> int:
        return self.hidden_size_override or (64 * self.depth_multimodal)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.patch_size ** 2 * self.vae_latent_dim

    @property
    def is_flux(self) -> bool:
        return self.depth_unified > 0

    def param_count(self) -> int:
        """Approximate parameter count (linear weights only): 12 h^2 per stream / single block plus
        the adaLN Linears."""
        h = self.hidden_size
        per_stream = (4 + 2 * self.mlp_ratio) * h * h
        n = (2 * self.depth_multimodal + self.depth_unified) * per_stream
        n += self.num_modulation_rows() * h * h
        n += (self.token_level_text_embed_dim + self.pooled_text_embed_dim + self.frequency_embed_dim + 2 * h) * h
        return n

    def num_modulation_rows(self) -> int:
        """Number of hidden-size rows of adaLN output per (timestep, batch row).

        double block: 6 (image) + 6 (text); the last SD3 block's text stream has 2
        (reference mmdit.py:64-65,424-427); single block: 3; final layer: 2.
        """
        n = 0
        for i in range(self.depth_multimodal):
            skip_txt = (i == self.depth_multimodal - 1) and self.depth_unified < 1
            n += 6 + (2 if skip_txt else 6)
        n += 3 * self.depth_unified
        n += 2
        return n


# reference config.py:78-80 (fp16 there; this build computes in bf16 on MI355X)
SD3_2b = MMDiTConfig(depth_multimodal=24, num_heads=24, dtype="float16")

# reference config.py:74-76
SD3_8b = MMDiTConfig(depth_multimodal=38, num_heads=38, use_qk_norm=True)

# reference config.py:82-95
FLUX_SCHNELL = MMDiTConfig(
    num_heads=24,
    depth_multimodal=19,
    depth_unified=38,
    parallel_mlp_for_unified_blocks=True,
    hidden_size_override=3072,
    patchify_via_reshape=True,
    pos_embed_type=PositionalEncoding.PreSDPARope,
    rope_axes_dim=(16, 56, 56),
    pooled_text_embed_dim=768,
    use_qk_norm=True,
    dtype="bfloat16",
)


# reference config.py:97-111 (declared there, never selected: quirk Q7)
FLUX_DEV = replace(FLUX_SCHNELL, guidance_embed=True)


# fp8 precision policy of the FLUX family, measured at full depth on MI355X (scripts/fp8_policy_gpu.py, profiles/r05_fp8_policy_gpu.log:
# Euler direction of teacher-forced steps 1 / 2 / 49 / 50 against the fp32 oracle): every double-stream block kept in bf16 buys ~0.27 dB and
# costs ~0.36 ms per step; 12 of the 19 put every step at >= 35 dB (35.1 - 36.4 dB, 46.3 ms per step against 41.8 all-fp8 / 61.6 bf16)
FLUX_FP8_QUALITY_BLOCKS = 12


def fp8_config(cfg: MMDiTConfig, policy: str = "quality") -> MMDiTConfig:
    """``cfg`` on the fp8 path (BASELINE.json configs[3]).  policy "quality" (default): the first FLUX_FP8_QUALITY_BLOCKS double-stream blocks keep
    bf16 Linears (>= 35 dB per step, SURVEY.md section 8c iii); "speed": every block Linear in fp8 (31.6 - 32.5 dB per step)."""
    if policy not in ("quality", "speed"):
        raise ValueError(f"unknown fp8 policy {policy!r} (quality | speed)")
    # (reject here what dk_mmdit_create would only reject later -- the fp8 kernels are FLUX geometry: head_dim 128, width a multiple of 256)
    if cfg.head_dim != 128 or cfg.hidden_size % 256 != 0:
        raise ValueError(f"fp8_config: the fp8 path needs head_dim 128 and hidden_size % 256 == 0 (got {cfg.head_dim}, {cfg.hidden_size})")
    n = min(FLUX_FP8_QUALITY_BLOCKS, cfg.depth_multimodal) if policy == "quality" else 0
    out = replace(cfg, weight_dtype="fp8_e4m3", fp8_bf16_double_blocks=n)
    validate_fp8_policy(out)
    return out


def validate_fp8_policy(cfg: MMDiTConfig) -> None:
    """0 <= fp8_bf16_double_blocks <= depth_multimodal (the C engine and weights.pack_mmdit clamp independently: a value outside the range
    would only surface as a missing-tensor error)."""
    n = cfg.fp8_bf16_double_blocks
    if not (0 <= n <= cfg.depth_multimodal):
        raise ValueError(f"fp8_bf16_double_blocks = {n}: must lie in [0, depth_multimodal = {cfg.depth_multimodal}]")


def tiny_flux(depth_multimodal: int = 2, depth_unified: int = 2, heads: int = 2,
              head_dim: int = 128, text_dim: int = 256, pooled: int = 64) -> MMDiTConfig:
    """FLUX-shaped config small enough for the CPU oracle (head_dim stays 128 so the
    RoPE axes (16,56,56) and the D=128 attention kernel are the production ones)."""
    return replace(
        FLUX_SCHNELL,
        num_heads=heads,
        depth_multimodal=depth_multimodal,
        depth_unified=depth_unified,
        hidden_size_override=heads * head_dim,
        token_level_text_embed_dim=text_dim,
        pooled_text_embed_dim=pooled,
    )


def tiny_sd3(depth: int = 2, heads: int = 4, text_dim: int = 256, pooled: int = 64,
             max_res: int = 24) -> MMDiTConfig:
    """SD3-shaped config (head_dim 64, learned pos-emb, conv patchify)."""
    return replace(
        SD3_2b,
        num_heads=heads,
        depth_multimodal=depth,
        hidden_size_override=heads * 64,
        token_level_text_embed_dim=text_dim,
        pooled_text_embed_dim=pooled,
        max_latent_resolution=max_res,
    )


@dataclass(frozen=True)
class VAEDecoderConfig:
    """reference config.py:126-132"""
    in_channels: int = 16
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 3
    resnet_groups: int = 32
    group_norm_eps: float = 1e-5  # MLX nn.GroupNorm default (vae.py:34,72,78,381)


@dataclass(frozen=True)
class VAEEncoderConfig:
    """reference config.py:135-140 (the image -> latent half used by img2img, vae.py:404-467)"""
    in_channels: int = 3
    out_channels: int = 32  # mean | logvar of the 16 latent channels
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    resnet_groups: int = 32
    group_norm_eps: float = 1e-5


def tiny_vae_encoder() -> VAEEncoderConfig:
    return VAEEncoderConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1)


def tiny_vae() -> VAEDecoderConfig:
    return VAEDecoderConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1)


# reference mlx/__init__.py:37-53
MMDIT_CKPT = {
    "argmaxinc/mlx-stable-diffusion-3-medium": "argmaxinc/mlx-stable-diffusion-3-medium",
    "argmaxinc/mlx-stable-diffusion-3.5-large": "argmaxinc/mlx-stable-diffusion-3.5-large",
    "argmaxinc/mlx-stable-diffusion-3.5-large-4bit-quantized": "argmaxinc/mlx-stable-diffusion-3.5-large-4bit-quantized",
    "argmaxinc/mlx-FLUX.1-schnell": "argmaxinc/mlx-FLUX.1-schnell",
    "argmaxinc/mlx-FLUX.1-schnell-4bit-quantized": "argmaxinc/mlx-FLUX.1-schnell-4bit-quantized",
    "argmaxinc/mlx-FLUX.1-dev": "argmaxinc/mlx-FLUX.1-dev",
}

T5_MAX_LENGTH = {
    "argmaxinc/mlx-stable-diffusion-3-medium": 512,
    "argmaxinc/mlx-stable-diffusion-3.5-large": 512,
    "argmaxinc/mlx-stable-diffusion-3.5-large-4bit-quantized": 512,
    "argmaxinc/mlx-FLUX.1-schnell": 256,
    "argmaxinc/mlx-FLUX.1-schnell-4bit-quantized": 256,
    "argmaxinc/mlx-FLUX.1-dev": 512,
}

# reference model_io.py:104-111 (_CONFIG): which MMDiT preset a model_version selects
MODEL_CONFIG = {
    "argmaxinc/mlx-stable-diffusion-3-medium": SD3_2b,
    "argmaxinc/mlx-FLUX.1-schnell": FLUX_SCHNELL,
    "argmaxinc/mlx-FLUX.1-schnell-4bit-quantized": FLUX_SCHNELL,
    "argmaxinc/mlx-FLUX.1-dev": FLUX_SCHNELL,  # quirk Q7: dev runs without guidance embedding
    "argmaxinc/mlx-stable-diffusion-3.5-large": SD3_8b,
    "argmaxinc/mlx-stable-diffusion-3.5-large-4bit-quantized": SD3_8b,
}
