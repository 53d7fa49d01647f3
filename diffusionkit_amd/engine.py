"""Host-side mirrors of the reference's model objects, backed by libdk_hip.so.

``MMDiTEngine`` mirrors ``MMDiT`` (python/src/diffusionkit/mlx/mmdit.py:22-266:
``cache_modulation_params`` / ``__call__``), ``VAEDecoderEngine`` mirrors ``VAEDecoder``
(python/src/diffusionkit/mlx/vae.py:336-401).  PyTorch is used for device memory and
streams only; every FLOP runs in the HIP library.  There is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from .config import MMDiTConfig, PositionalEncoding, VAEDecoderConfig

Tensor = torch.Tensor
_EMBED_DTYPE = {"bfloat16": 0, "float16": 1, "float32": 2}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_cuda(t: Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise _lib.DkHipError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise _lib.DkHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.DkHipError(f"{name} must be contiguous")


class MMDiTEngine:
    """Drop-in for the reference ``MMDiT`` module on the hot path."""

    def __init__(self, config: MMDiTConfig, packed_weights: Dict[str, Tensor]):
        self.lib = _lib.load()
        self.config = config
        dev0 = next(iter(packed_weights.values())).device
        if dev0.type == "cuda":
            _lib.ensure_attention_workspace(dev0)  # (the key-split workgroups of the D = 128 attention kernel)
        c = _lib.dk_mmdit_config()
        c.num_heads, c.depth_multimodal, c.depth_unified = config.num_heads, config.depth_multimodal, config.depth_unified
        c.hidden_size, c.mlp_ratio = config.hidden_size, config.mlp_ratio
        c.vae_latent_dim, c.patch_size = config.vae_latent_dim, config.patch_size
        c.patchify_via_reshape = int(config.patchify_via_reshape)
        c.use_qk_norm = int(config.use_qk_norm)
        c.use_rope = int(config.pos_embed_type == PositionalEncoding.PreSDPARope)
        axes = list(config.rope_axes_dim or ())
        for i, a in enumerate(axes):
            c.rope_axes_dim[i] = a
        c.n_rope_axes, c.rope_theta = len(axes), config.rope_theta
        c.use_pos_embed = int(config.pos_embed_type == PositionalEncoding.LearnedInputEmbedding)
        c.max_latent_resolution = config.max_latent_resolution
        c.pooled_text_embed_dim = config.pooled_text_embed_dim
        c.token_level_text_embed_dim = config.token_level_text_embed_dim
        c.frequency_embed_dim, c.max_period = config.frequency_embed_dim, config.max_period
        c.embed_dtype = _EMBED_DTYPE[config.dtype]
        c.layer_norm_eps = config.layer_norm_eps
        c.guidance_embed = int(config.guidance_embed)
        if config.weight_dtype not in ("bfloat16", "fp8_e4m3"):
            raise _lib.DkHipError(f"unknown weight_dtype {config.weight_dtype!r} (bfloat16 | fp8_e4m3)")
        c.fp8_linears = int(config.weight_dtype == "fp8_e4m3")
        if c.fp8_linears:
            from .config import validate_fp8_policy
            validate_fp8_policy(config)
        c.fp8_bf16_double_blocks = int(config.fp8_bf16_double_blocks) if c.fp8_linears else 0
        h = C.c_void_p()
        _lib.check(self.lib.dk_mmdit_create(C.byref(c), C.byref(h)), "dk_mmdit_create")
        self._h = h
        self.weights = packed_weights  # keep the device tensors alive
        for name, t in packed_weights.items():
            want = torch.uint8 if name.endswith(".weight_fp8") else torch.float32 if name.endswith(".wscale") else torch.bfloat16
            _require_cuda(t, name, want)
            _lib.check(self.lib.dk_mmdit_bind(self._h, name.encode(), t.data_ptr()), f"bind {name}")
        self.guidance = 3.5  # FLUX.1-dev's default distilled-guidance strength; only read when config.guidance_embed
        self._shape = None
        self._ws = None
        self._n_cached = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.dk_mmdit_destroy(self._h)
            self._h = None

    def _check_pitched_weights(self) -> None:
        """The engine reads the long-reduction weights at ``dk_weight_pitch`` elements per row (include/dk_hip.h); the C ABI
        binds bare pointers, so the layout the packer produced is checked here, where the shapes are still known."""
        h, r = self.config.hidden_size, self.config.mlp_ratio
        for name, t in self.weights.items():
            fp8 = name.endswith(".weight_fp8")
            base = name[:-len(".weight_fp8" if fp8 else ".weight")] if (fp8 or name.endswith(".weight")) else ""
            k = r * h if base.endswith(".mlp.fc2") else (1 + r) * h if base.endswith(".linear2") else 0
            pitch = (self.lib.dk_weight_pitch_fp8 if fp8 else self.lib.dk_weight_pitch)(k) if k else 0
            if k and t.shape[1] != pitch:
                raise _lib.DkHipError(f"{name}: {t.shape[1]} elements per row, the engine expects dk_weight_pitch({k}) = "
                                      f"{pitch} (was the weight packed under a different pitch_min_k?)")

    # -- shape / workspace ---------------------------------------------------------------
    def prepare(self, batch: int, latent_size: Sequence[int], text_len: int, n_timesteps: int) -> None:
        shape = (batch, int(latent_size[0]), int(latent_size[1]), text_len, n_timesteps)
        if self._shape == shape:
            return
        self._check_pitched_weights()
        nbytes = self.lib.dk_mmdit_workspace_bytes(self._h, *shape)
        dev = next(iter(self.weights.values())).device
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(self.lib.dk_mmdit_prepare(self._h, *shape, self._ws.data_ptr(), self._ws.numel(), _stream()),
                   "dk_mmdit_prepare")
        self._shape = shape
        self._n_cached = 0
        self._ctx_cached = False

    @property
    def batch(self):
        return self._shape[0]

    def tokens_shape(self):
        b, hl, wl, _, _ = self._shape
        p = self.config.patch_size
        return (b, (hl // p) * (wl // p), self.config.patch_dim)

    def mod_rows(self) -> int:
        return self.lib.dk_mmdit_mod_rows(self._h)

    def mod_offset(self, kind: int, index: int) -> int:
        return self.lib.dk_mmdit_mod_offset(self._h, kind, index)

    # -- reference API ---------------------------------------------------------------------
    def cache_modulation_params(self, pooled_text_embeddings: Tensor, timesteps: Sequence[float]) -> None:
        """MMDiT.cache_modulation_params (mmdit.py:77-180); ``timesteps`` are host floats
        already rounded to the activation dtype (quirk Q1)."""
        if self._shape is None:
            raise _lib.DkHipError("prepare() must be called before cache_modulation_params()")
        pooled = pooled_text_embeddings.to(torch.bfloat16).contiguous()
        _require_cuda(pooled, "pooled_text_embeddings")
        if pooled.shape != (self._shape[0], self.config.pooled_text_embed_dim):
            raise _lib.DkHipError(f"pooled_text_embeddings shape {tuple(pooled.shape)} != (batch, pooled_dim)")
        ts = [float(t) for t in timesteps]
        if self.config.guidance_embed:
            _lib.check(self.lib.dk_mmdit_set_guidance(self._h, float(self.guidance)), "dk_mmdit_set_guidance")
        arr = (C.c_float * len(ts))(*ts)
        _lib.check(self.lib.dk_mmdit_cache_modulation_params(self._h, pooled.data_ptr(), arr, len(ts), _stream()),
                   "dk_mmdit_cache_modulation_params")
        self._pooled_keepalive = pooled
        self._n_cached = len(ts)

    def cache_context(self, text: Tensor) -> None:
        """Step-invariant hoist of ``context_embedder(text)`` (mmdit.py:195): embedded once here, ``forward_tokens(..., None, i)``
        then reuses it in every step (the reference recomputes the same values per call)."""
        _require_cuda(text, "text", torch.bfloat16)
        if tuple(text.shape) != (self._shape[0], self._shape[3], self.config.token_level_text_embed_dim):
            raise _lib.DkHipError(f"text shape {tuple(text.shape)} does not match the prepared problem")
        _lib.check(self.lib.dk_mmdit_cache_context(self._h, text.data_ptr(), _stream()), "dk_mmdit_cache_context")
        self._ctx_cached = True

    def forward_tokens(self, tokens_in: Tensor, text: Optional[Tensor], step_index: int, tokens_out: Optional[Tensor] = None) -> Tensor:
        """MMDiT.__call__ between patchify and unpatchify (mmdit.py:188-252).  ``text=None`` uses ``cache_context``'s result."""
        _require_cuda(tokens_in, "tokens_in", torch.bfloat16)
        if tuple(tokens_in.shape) != self.tokens_shape():
            raise _lib.DkHipError(f"tokens_in shape {tuple(tokens_in.shape)} != {self.tokens_shape()}")
        if text is None:
            if not getattr(self, "_ctx_cached", False):
                raise _lib.DkHipError("forward_tokens(text=None) needs cache_context(text) after prepare()")
        else:
            _require_cuda(text, "text", torch.bfloat16)
            if tuple(text.shape) != (self._shape[0], self._shape[3], self.config.token_level_text_embed_dim):
                raise _lib.DkHipError(f"text shape {tuple(text.shape)} does not match the prepared problem")
        if not (0 <= step_index < self._n_cached):
            raise KeyError(f"no cached modulation parameters for step {step_index}")  # reference: dict KeyError
        if tokens_out is None:
            tokens_out = torch.empty_like(tokens_in)
        _lib.check(self.lib.dk_mmdit_forward(self._h, tokens_in.data_ptr(), None if text is None else text.data_ptr(), step_index,
                                             tokens_out.data_ptr(), _stream()), "dk_mmdit_forward")
        return tokens_out

    def run_blocks(self, x: Tensor, step_index: int, first_block: int, n_blocks: int = 1) -> Tensor:
        """MultiModalTransformerBlock / UnifiedTransformerBlock.__call__ (mmdit.py:568-675, 693-751) on a caller-supplied joint
        residual stream ``x`` (bf16 [batch, S_t + S_i, h], text rows first): blocks ``first_block .. first_block + n_blocks - 1`` of
        the global order (double blocks, then single blocks) with the modulation cached for ``step_index``."""
        _require_cuda(x, "x", torch.bfloat16)
        b, hl, wl, s_t, _ = self._shape
        p = self.config.patch_size
        want = (b, s_t + (hl // p) * (wl // p), self.config.hidden_size)
        if tuple(x.shape) != want:
            raise _lib.DkHipError(f"x shape {tuple(x.shape)} != {want}")
        if not (0 <= step_index < self._n_cached):
            raise KeyError(f"no cached modulation parameters for step {step_index}")
        out = torch.empty_like(x)
        _lib.check(self.lib.dk_mmdit_run_blocks(self._h, x.data_ptr(), out.data_ptr(), step_index, first_block, n_blocks, _stream()),
                   "dk_mmdit_run_blocks")
        return out

    def patchify(self, latent: Tensor, dup: int = 1) -> Tensor:
        """LatentImageAdapter reshape (mmdit.py:292-300): f32 NHWC -> bf16 tokens."""
        _require_cuda(latent, "latent", torch.float32)
        n, hl, wl, c = latent.shape
        p = self.config.patch_size
        tok = torch.empty(n * dup, (hl // p) * (wl // p), p * p * c, dtype=torch.bfloat16, device=latent.device)
        _lib.check(self.lib.dk_latent_to_tokens(latent.data_ptr(), tok.data_ptr(), n, dup, hl, wl, c, p,
                                                int(self.config.patchify_via_reshape), _stream()), "dk_latent_to_tokens")
        return tok

    def __call__(self, latent_image_embeddings: Tensor, token_level_text_embeddings: Tensor, step_index: int) -> Tensor:
        """MMDiT.__call__ (mmdit.py:188-266) for NHWC latents; returns tokens-major FinalLayer
        output un-patchified to NHWC bf16 (used by parity tests; the step loop stays fused)."""
        lat = latent_image_embeddings.to(torch.float32).contiguous()
        tok = self.patchify(lat)
        text = token_level_text_embeddings
        if text.dim() == 4:  # reference passes [B, S_t, 1, T]
            text = text.squeeze(2)
        out = self.forward_tokens(tok, text.to(torch.bfloat16).contiguous(), step_index)
        return unpatchify_tokens(out, self.config, lat.shape[1], lat.shape[2])


def unpatchify_tokens(tok: Tensor, cfg: MMDiTConfig, hl: int, wl: int) -> Tensor:
    """Pure index shuffle used only by tests / the NHWC convenience call
    (mmdit.py:304-321, 975-988); the production step loop does this inside dk_euler_cfg_step."""
    b = tok.shape[0]
    p, c = cfg.patch_size, cfg.vae_latent_dim
    h, w = hl // p, wl // p
    if cfg.patchify_via_reshape:
        return tok.reshape(b, h, w, c, p, p).permute(0, 1, 4, 2, 5, 3).reshape(b, hl, wl, c)
    return tok.reshape(b, h, w, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(b, hl, wl, c)


class VAEDecoderEngine:
    """Drop-in for the reference ``VAEDecoder`` (+ the clip / uint8 tail)."""

    def __init__(self, config: VAEDecoderConfig, packed_weights: Dict[str, Tensor]):
        self.lib = _lib.load()
        self.config = config
        c = _lib.dk_vae_config()
        c.in_channels, c.out_channels = config.in_channels, config.out_channels
        for i, ch in enumerate(config.block_out_channels):
            c.block_out_channels[i] = ch
        c.n_blocks = len(config.block_out_channels)
        c.layers_per_block, c.resnet_groups = config.layers_per_block, config.resnet_groups
        c.group_norm_eps = config.group_norm_eps
        h = C.c_void_p()
        _lib.check(self.lib.dk_vae_create(C.byref(c), C.byref(h)), "dk_vae_create")
        self._h = h
        self.weights = packed_weights
        for name, t in packed_weights.items():
            _require_cuda(t, name, torch.bfloat16)
            _lib.check(self.lib.dk_vae_bind(self._h, name.encode(), t.data_ptr()), f"bind {name}")
        self._ws = None

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.dk_vae_destroy(self._h)
            self._h = None

    def decode(self, x: Tensor, want_raw: bool = False):
        """x: f32 [B,h,w,16] -> (image f32 [B,8h,8w,3] in [0,1], uint8 image, raw bf16 or None)."""
        x = x.to(torch.float32).contiguous()
        _require_cuda(x, "latent")
        b, h, w, _ = x.shape
        nbytes = self.lib.dk_vae_workspace_bytes(self._h, b, h, w)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        scale = 2 ** (len(self.config.block_out_channels) - 1)
        img = torch.empty(b, h * scale, w * scale, 3, dtype=torch.float32, device=x.device)
        u8 = torch.empty(b, h * scale, w * scale, 3, dtype=torch.uint8, device=x.device)
        raw = torch.empty(b, h * scale, w * scale, 4, dtype=torch.bfloat16, device=x.device) if want_raw else None
        _lib.check(self.lib.dk_vae_decode(self._h, x.data_ptr(), b, h, w, img.data_ptr(), u8.data_ptr(), _ptr(raw),
                                          self._ws.data_ptr(), self._ws.numel(), _stream()), "dk_vae_decode")
        return img, u8, raw

    def __call__(self, x: Tensor) -> Tensor:
        """VAEDecoder.__call__ (vae.py:386-401): returns the un-clipped decoder output [B,8h,8w,3]."""
        _, _, raw = self.decode(x, want_raw=True)
        return raw[..., :3]


class VAEEncoderEngine:
    """Drop-in for the reference ``VAEEncoder`` (vae.py:404-467): image in [-1, 1] -> moments."""

    def __init__(self, config, packed_weights: Dict[str, Tensor]):
        self.lib = _lib.load()
        self.config = config
        c = _lib.dk_vae_config()
        c.in_channels, c.out_channels = config.in_channels, config.out_channels
        for i, ch in enumerate(config.block_out_channels):
            c.block_out_channels[i] = ch
        c.n_blocks = len(config.block_out_channels)
        c.layers_per_block, c.resnet_groups = config.layers_per_block, config.resnet_groups
        c.group_norm_eps = config.group_norm_eps
        h = C.c_void_p()
        _lib.check(self.lib.dk_vae_create(C.byref(c), C.byref(h)), "dk_vae_create")
        self._h = h
        self.weights = packed_weights
        for name, t in packed_weights.items():
            _require_cuda(t, name, torch.bfloat16)
            _lib.check(self.lib.dk_vae_bind(self._h, name.encode(), t.data_ptr()), f"bind {name}")
        self._ws = None

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.dk_vae_destroy(self._h)
            self._h = None

    def encode(self, image: Tensor):
        """image: f32 [B,H,W,3] in [-1,1] -> moments bf16 [B,H/8,W/8,ldm] (mean | logvar in the first
        ``out_channels`` columns)."""
        image = image.to(torch.float32).contiguous()
        _require_cuda(image, "image")
        b, H, W, _ = image.shape
        nbytes = self.lib.dk_vae_encoder_workspace_bytes(self._h, b, H, W)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=image.device)
        down = 2 ** (len(self.config.block_out_channels) - 1)
        ldm = (self.config.out_channels + 3) // 4 * 4
        mom = torch.empty(b, H // down, W // down, ldm, dtype=torch.bfloat16, device=image.device)
        _lib.check(self.lib.dk_vae_encode(self._h, image.data_ptr(), b, H, W, mom.data_ptr(), ldm, None,
                                          self._ws.data_ptr(), self._ws.numel(), _stream()), "dk_vae_encode")
        return mom

    def __call__(self, image: Tensor) -> Tensor:
        """VAEEncoder.__call__ (vae.py:456-467): hidden [B,H/8,W/8,out_channels]."""
        return self.encode(image)[..., :self.config.out_channels]

    def sample(self, moments: Tensor, noise: Tensor) -> Tensor:
        """encode_image_to_latents tail (__init__.py:588-594): mean + exp(0.5 * clip(logvar)) * noise, f32."""
        _require_cuda(moments, "moments", torch.bfloat16)
        noise = noise.to(torch.float32).contiguous()
        _require_cuda(noise, "noise")
        L = self.config.out_channels // 2
        b, h, w, ldm = moments.shape
        assert noise.shape == (b, h, w, L), (noise.shape, (b, h, w, L))
        out = torch.empty(b, h, w, L, dtype=torch.float32, device=moments.device)
        _lib.check(self.lib.dk_latent_sample_f32(moments.data_ptr(), ldm, noise.data_ptr(), out.data_ptr(), b * h * w, L,
                                                 _stream()), "dk_latent_sample_f32")
        return out
