"""CPU ORACLE (test infrastructure, NOT product code) -- latent-decode VAE (+ the img2img encoder half).

PyTorch-CPU restatement of the reference's VAEDecoder
(python/src/diffusionkit/mlx/vae.py:20-25 upsample_nearest, :28-57 Attention,
:60-101 ResnetBlock2D, :104-149 EncoderDecoderBlock2D, :336-401 VAEDecoder).
Parity: wiring PINNED against the reference's own MLX vae.py run on the MLX stand-in (decoder and encoder, 5e-6) and against
its PyTorch VAEDecoder (torch/vae.py, 2e-6); MLX's arithmetic unpinned (see oracle/mmdit.py header).  Tensors are NHWC float32; weights use
the MLX layouts (Conv2d [O,kh,kw,I], Linear [O,I]).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .mmdit import Prec, linear

Tensor = torch.Tensor


def conv2d_nhwc(x: Tensor, w: Tensor, b: Optional[Tensor], P: Prec, padding: int = 1) -> Tensor:
    """nn.Conv2d on NHWC input with weight [O,kh,kw,I] (vae.py:73,79,134,349,384)."""
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, padding=padding)
    return P.r(y.permute(0, 2, 3, 1))


def group_norm_nhwc(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, P: Prec) -> Tensor:
    """nn.GroupNorm(pytorch_compatible=True): statistics over (H, W, C/groups) per
    (batch, group), fp32, then affine; one rounding (vae.py:34,72,78,381)."""
    y = F.group_norm(x.permute(0, 3, 1, 2), groups, gamma, beta, eps)
    return P.r(y.permute(0, 2, 3, 1))


def silu(x: Tensor, P: Prec) -> Tensor:
    return P.r(x * torch.sigmoid(x))


def upsample_nearest(x: Tensor, scale: int = 2) -> Tensor:
    """vae.py:20-25"""
    return x.repeat_interleave(scale, dim=1).repeat_interleave(scale, dim=2)


class OracleVAEDecoder:
    def __init__(self, cfg, weights: Dict[str, Tensor], prec: Optional[Prec] = None):
        self.cfg = cfg
        self.w = weights
        self.P = prec or Prec()

    def _gn(self, x, name):
        c = self.cfg
        return group_norm_nhwc(x, self.w[name + ".weight"], self.w[name + ".bias"],
                               c.resnet_groups, c.group_norm_eps, self.P)

    def _conv(self, x, name):
        return conv2d_nhwc(x, self.w[name + ".weight"], self.w[name + ".bias"], self.P)

    def _resnet(self, x, p):
        """ResnetBlock2D.__call__ (vae.py:86-101)"""
        P = self.P
        y = self._conv(silu(self._gn(x, p + ".norm1"), P), p + ".conv1")
        y = self._conv(silu(self._gn(y, p + ".norm2"), P), p + ".conv2")
        if (p + ".conv_shortcut.weight") in self.w:  # 1x1 shortcut stored as Linear (vae.py:84)
            x = linear(x, self.w[p + ".conv_shortcut.weight"], self.w[p + ".conv_shortcut.bias"], P)
        return P.r(y + x)

    def _attention(self, x, p):
        """Single-head attention over H*W tokens (vae.py:37-57)."""
        P = self.P
        B, H, W, C = x.shape
        y = self._gn(x, p + ".group_norm")
        q = linear(y, self.w[p + ".query_proj.weight"], self.w[p + ".query_proj.bias"], P).reshape(B, H * W, C)
        k = linear(y, self.w[p + ".key_proj.weight"], self.w[p + ".key_proj.bias"], P).reshape(B, H * W, C)
        v = linear(y, self.w[p + ".value_proj.weight"], self.w[p + ".value_proj.bias"], P).reshape(B, H * W, C)
        scale = 1.0 / math.sqrt(C)
        s = P.r(P.r(q * scale) @ k.transpose(1, 2))
        a = P.r(torch.softmax(s, dim=-1))
        y = P.r(a @ v).reshape(B, H, W, C)
        y = linear(y, self.w[p + ".out_proj.weight"], self.w[p + ".out_proj.bias"], P)
        return P.r(x + y)

    def __call__(self, x: Tensor, taps: Optional[dict] = None) -> Tensor:
        """VAEDecoder.__call__ (vae.py:386-401). x: [B,h,w,16] -> [B,8h,8w,3]."""
        c, P = self.cfg, self.P
        x = self._conv(P.r(x), "conv_in")
        x = self._resnet(x, "mid_blocks.0")
        x = self._attention(x, "mid_blocks.1")
        x = self._resnet(x, "mid_blocks.2")
        if taps is not None:
            taps["mid"] = x.clone()
        n = len(c.block_out_channels)
        # up_blocks is built with insert(0, ...) and run reversed (vae.py:379,393):
        # execution order is list index n-1 ... 0; index 0 has no upsample.
        for j in reversed(range(n)):
            for r in range(c.layers_per_block):
                x = self._resnet(x, f"up_blocks.{j}.resnets.{r}")
            if (f"up_blocks.{j}.upsample.weight") in self.w:
                x = self._conv(upsample_nearest(x), f"up_blocks.{j}.upsample")
            if taps is not None:
                taps[f"up{j}"] = x.clone()
        x = silu(self._gn(x, "conv_norm_out"), P)
        return self._conv(x, "conv_out")


def conv2d_s2_pad_br_nhwc(x: Tensor, w: Tensor, b: Optional[Tensor], P: Prec) -> Tensor:
    """EncoderDecoderBlock2D downsample (vae.py:141-143): mx.pad(x, [(0,0),(0,1),(0,1),(0,0)]) then
    nn.Conv2d(k3, stride 2, padding 0)."""
    xp = F.pad(x.permute(0, 3, 1, 2), (0, 1, 0, 1))
    y = F.conv2d(xp, w.permute(0, 3, 1, 2), b, stride=2, padding=0)
    return P.r(y.permute(0, 2, 3, 1))


class OracleVAEEncoder(OracleVAEDecoder):
    """VAEEncoder (vae.py:404-467); shares the block restatements of the decoder oracle."""

    def __call__(self, x: Tensor, taps: Optional[dict] = None) -> Tensor:
        """x: [B,H,W,3] in [-1,1] -> hidden [B,H/8,W/8,32] (mean | logvar)."""
        c, P = self.cfg, self.P
        x = self._conv(P.r(x), "conv_in")
        n = len(c.block_out_channels)
        for i in range(n):
            for r in range(c.layers_per_block):
                x = self._resnet(x, f"down_blocks.{i}.resnets.{r}")
            if (f"down_blocks.{i}.downsample.weight") in self.w:
                x = conv2d_s2_pad_br_nhwc(x, self.w[f"down_blocks.{i}.downsample.weight"],
                                          self.w[f"down_blocks.{i}.downsample.bias"], P)
            if taps is not None:
                taps[f"down{i}"] = x.clone()
        x = self._resnet(x, "mid_blocks.0")
        x = self._attention(x, "mid_blocks.1")
        x = self._resnet(x, "mid_blocks.2")
        x = silu(self._gn(x, "conv_norm_out"), P)
        return self._conv(x, "conv_out")


def sample_latent(hidden: Tensor, noise: Tensor) -> Tensor:
    """encode_image_to_latents tail (mlx/__init__.py:588-594): split mean / logvar on the channel axis,
    clip logvar to [-30, 20], latent = mean + exp(0.5 * logvar) * noise (fp32: the reference's encoder
    output is fp32 because the image enters as fp32)."""
    mean, logvar = hidden.float().chunk(2, dim=-1)
    logvar = torch.clip(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise.float()


def decode_latents_to_image(decoder: OracleVAEDecoder, x_t: Tensor) -> Tensor:
    """mlx/__init__.py:581-584: clip(x/2 + 0.5, 0, 1)."""
    x = decoder(x_t)
    return torch.clip(x / 2 + 0.5, 0, 1)


def to_uint8(img01: Tensor) -> Tensor:
    """mlx/__init__.py:525-526: (x*255).astype(uint8) truncates (quirk Q12)."""
    return (img01 * 255).to(torch.uint8)
