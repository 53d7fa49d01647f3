"""Weights of the hot path: synthetic initialisation and packing into the device layout.

There is no network / checkpoint in this environment, so benchmarks and parity tests use
seeded random weights with the shapes of the reference's module tree
(Linear/Conv weights N(0, 0.02^2), biases N(0, 0.02^2),
norm gains 1 + N(0, 0.02^2)).  Tensor names follow the reference's MLX module tree after
``model_io`` remapping (python/src/diffusionkit/mlx/model_io.py:130-311, 314-408, 411-486),
so a real checkpoint loader ("next" row f1) only has to produce the same dict.

``pack_mmdit`` / ``pack_vae`` turn that dict into the fused, K-major bf16 tensors the HIP
engine binds by name:
  * q/k/v projections -> one [3h, h] matrix, bias [3h] with a zero k part (quirk Q9);
  * single blocks: q|k|v|fc1 -> linear1 [7h, h]; o_proj | fc2 -> linear2 [h, 5h] with ONE bias (quirk Q8);
  * all adaLN_modulation Linears -> one [rows*h, h] matrix in the engine's row order;
  * conv weights [O,3,3,I] -> [O, 9*I] (already K-major in the MLX layout), conv_in padded to I=64.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib
from .config import MMDiTConfig, VAEDecoderConfig

Tensor = torch.Tensor


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


class _Init:
    """Seeded initialiser; with ``shapes_only`` it records ``name -> shape`` instead of drawing tensors
    (the checkpoint loader validates against that table)."""

    def __init__(self, device, seed, dtype, shapes_only=False):
        self.shapes_only = shapes_only
        self.out: Dict[str, Tensor] = {}
        if not shapes_only:
            self.device, self.dtype = torch.device(device), dtype
            self.g = _gen(self.device, seed)

    def normal(self, name, shape, std=0.02, mean=0.0):
        if self.shapes_only:
            self.out[name] = tuple(shape)
            return
        t = torch.randn(*shape, generator=self.g, device=self.device, dtype=torch.float32) * std + mean
        self.out[name] = t.to(self.dtype)

    def linear(self, prefix, out_f, in_f, bias=True):
        self.normal(prefix + ".weight", (out_f, in_f))
        if bias:
            self.normal(prefix + ".bias", (out_f,))


def mmdit_weight_shapes(cfg: MMDiTConfig) -> Dict[str, tuple]:
    """name -> shape of every tensor of the reference MMDiT module tree for ``cfg``."""
    return synth_mmdit_weights(cfg, shapes_only=True)


def vae_weight_shapes(cfg: VAEDecoderConfig) -> Dict[str, tuple]:
    return synth_vae_weights(cfg, shapes_only=True)


def synth_mmdit_weights(cfg: MMDiTConfig, seed: int = 1234, device="cpu", dtype=torch.bfloat16, shapes_only=False) -> Dict[str, Tensor]:
    """Random weights keyed like the reference MMDiT module tree (mmdit.py:22-75)."""
    I = _Init(device, seed, dtype, shapes_only)
    h, D, r = cfg.hidden_size, cfg.head_dim, cfg.mlp_ratio
    p = cfg.patch_size
    if cfg.patchify_via_reshape:
        I.normal("x_embedder.proj.weight", (h, 1, 1, cfg.patch_dim))
    else:
        I.normal("x_embedder.proj.weight", (h, p, p, cfg.vae_latent_dim))
    I.normal("x_embedder.proj.bias", (h,))
    if cfg.rope_axes_dim is None:
        I.normal("x_pos_embedder.pos_embed.weight", (cfg.max_latent_resolution ** 2, h))
    for emb, d_in in (("y_embedder", cfg.pooled_text_embed_dim), ("t_embedder", cfg.frequency_embed_dim)):
        I.linear(f"{emb}.mlp.layers.0", h, d_in)
        I.linear(f"{emb}.mlp.layers.2", h, h)
    I.linear("context_embedder", h, cfg.token_level_text_embed_dim)
    if cfg.guidance_embed:  # MLPEmbedder(in_dim=frequency_embed_dim, hidden_dim=hidden_size), mmdit.py:31-36,945-955
        I.linear("guidance_in.mlp.layers.0", h, cfg.frequency_embed_dim)
        I.linear("guidance_in.mlp.layers.2", h, h)

    def block(prefix, n_mod, skip_post=False, parallel=False):
        I.linear(prefix + ".attn.q_proj", h, h)
        I.linear(prefix + ".attn.k_proj", h, h, bias=False)
        I.linear(prefix + ".attn.v_proj", h, h)
        if cfg.use_qk_norm:
            I.normal(prefix + ".qk_norm.q_norm.weight", (D,), mean=1.0)
            I.normal(prefix + ".qk_norm.k_norm.weight", (D,), mean=1.0)
        if not skip_post:
            I.linear(prefix + ".attn.o_proj", h, h)
            I.linear(prefix + ".mlp.fc1", r * h, h)
            I.linear(prefix + ".mlp.fc2", h, r * h, bias=not parallel)
        I.linear(prefix + ".adaLN_modulation.layers.1", n_mod * h, h)

    for i in range(cfg.depth_multimodal):
        skip_txt = (i == cfg.depth_multimodal - 1) and cfg.depth_unified < 1
        block(f"multimodal_transformer_blocks.{i}.image_transformer_block", 6)
        block(f"multimodal_transformer_blocks.{i}.text_transformer_block", 2 if skip_txt else 6, skip_post=skip_txt)
    for i in range(cfg.depth_unified):
        block(f"unified_transformer_blocks.{i}.transformer_block", 3, parallel=True)
    I.linear("final_layer.linear", cfg.patch_dim, h)
    I.linear("final_layer.adaLN_modulation.layers.1", 2 * h, h)
    return I.out


def synth_vae_weights(cfg: VAEDecoderConfig, seed: int = 4321, device="cpu", dtype=torch.bfloat16, shapes_only=False) -> Dict[str, Tensor]:
    """Random weights keyed like the reference VAEDecoder module tree (vae.py:336-384)."""
    I = _Init(device, seed, dtype, shapes_only)

    def conv(prefix, o, i):
        I.normal(prefix + ".weight", (o, 3, 3, i))
        I.normal(prefix + ".bias", (o,))

    def norm(prefix, c):
        I.normal(prefix + ".weight", (c,), mean=1.0)
        I.normal(prefix + ".bias", (c,))

    def resnet(prefix, cin, cout):
        norm(prefix + ".norm1", cin)
        conv(prefix + ".conv1", cout, cin)
        norm(prefix + ".norm2", cout)
        conv(prefix + ".conv2", cout, cout)
        if cin != cout:
            I.linear(prefix + ".conv_shortcut", cout, cin)

    boc = list(cfg.block_out_channels)
    cm = boc[-1]
    conv("conv_in", cm, cfg.in_channels)
    resnet("mid_blocks.0", cm, cm)
    norm("mid_blocks.1.group_norm", cm)
    for n in ("query_proj", "key_proj", "value_proj", "out_proj"):
        I.linear(f"mid_blocks.1.{n}", cm, cm)
    resnet("mid_blocks.2", cm, cm)
    cprev = cm
    for j in reversed(range(len(boc))):  # execution order: list index n-1 ... 0 (vae.py:379,393)
        cout = boc[j]
        for rr in range(cfg.layers_per_block):
            resnet(f"up_blocks.{j}.resnets.{rr}", cprev if rr == 0 else cout, cout)
        if j > 0:
            conv(f"up_blocks.{j}.upsample", cout, cout)
        cprev = cout
    norm("conv_norm_out", boc[0])
    conv("conv_out", cfg.out_channels, boc[0])
    return I.out


def synth_vae_encoder_weights(cfg, seed: int = 8765, device="cpu", dtype=torch.bfloat16, shapes_only=False) -> Dict[str, Tensor]:
    """Random weights keyed like the reference VAEEncoder module tree (vae.py:404-454)."""
    I = _Init(device, seed, dtype, shapes_only)

    def conv(prefix, o, i):
        I.normal(prefix + ".weight", (o, 3, 3, i))
        I.normal(prefix + ".bias", (o,))

    def norm(prefix, c):
        I.normal(prefix + ".weight", (c,), mean=1.0)
        I.normal(prefix + ".bias", (c,))

    def resnet(prefix, cin, cout):
        norm(prefix + ".norm1", cin)
        conv(prefix + ".conv1", cout, cin)
        norm(prefix + ".norm2", cout)
        conv(prefix + ".conv2", cout, cout)
        if cin != cout:
            I.linear(prefix + ".conv_shortcut", cout, cin)

    boc = list(cfg.block_out_channels)
    conv("conv_in", boc[0], cfg.in_channels)
    cprev = boc[0]
    for i, cout in enumerate(boc):
        for rr in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{rr}", cprev if rr == 0 else cout, cout)
        if i < len(boc) - 1:
            conv(f"down_blocks.{i}.downsample", cout, cout)
        cprev = cout
    cm = boc[-1]
    resnet("mid_blocks.0", cm, cm)
    norm("mid_blocks.1.group_norm", cm)
    for n in ("query_proj", "key_proj", "value_proj", "out_proj"):
        I.linear(f"mid_blocks.1.{n}", cm, cm)
    resnet("mid_blocks.2", cm, cm)
    norm("conv_norm_out", cm)
    conv("conv_out", cfg.out_channels, cm)
    return I.out


def vae_encoder_weight_shapes(cfg) -> Dict[str, tuple]:
    return synth_vae_encoder_weights(cfg, shapes_only=True)


# ---------------------------------------------------------------------------------------------
# packing
# ---------------------------------------------------------------------------------------------
def adaln_order(cfg: MMDiTConfig):
    """Module order of the packed adaLN matrix = engine modulation-table row order
    (dk_mmdit_mod_offset): per double block image(6) then text(6|2), singles(3), final(2)."""
    names = []
    for i in range(cfg.depth_multimodal):
        names.append(f"multimodal_transformer_blocks.{i}.image_transformer_block")
        names.append(f"multimodal_transformer_blocks.{i}.text_transformer_block")
    for i in range(cfg.depth_unified):
        names.append(f"unified_transformer_blocks.{i}.transformer_block")
    names.append("final_layer")
    return names


E4M3_MAX = 448.0


def quantize_weight_e4m3(w: Tensor):
    """[N, K] weight -> (e4m3 bytes as uint8 [N, K], fp32 scale [N]): one scale per output channel, scale = row amax / 448
    (an all-zero row gets scale 1), elements = round-to-nearest-even(w / scale) in OCP e4m3.  The arithmetic runs in fp32 on
    whatever device ``w`` lives on; ``dequantize_weight_e4m3`` gives back the values the fp8 GEMM multiplies with."""
    wf = w.to(torch.float32)
    amax = wf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
    q = (wf / scale[:, None]).clamp_(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequantize_weight_e4m3(q: Tensor, scale: Tensor, k: int = None) -> Tensor:
    """fp32 values of a quantised weight (pad columns of a pitched matrix cut off with ``k``)."""
    w = q.view(torch.float8_e4m3fn).to(torch.float32) * scale.to(torch.float32)[:, None]
    return w if k is None else w[:, :k]


BLOCK_LINEARS = (".attn.qkv", ".attn.o_proj", ".mlp.fc1", ".mlp.fc2", ".linear1", ".linear2")  # what weight_dtype = "fp8_e4m3" quantises


def pack_mmdit(cfg: MMDiTConfig, w: Dict[str, Tensor], device, consume: bool = False) -> Dict[str, Tensor]:
    """Reference-named weights -> engine tensors (bf16, contiguous, on ``device``).
    ``consume=True`` pops source tensors as they are packed (halves peak memory at full size).
    With ``cfg.weight_dtype == "fp8_e4m3"`` the Linear matrices of the transformer blocks leave as
    "<name>.weight_fp8" (uint8 e4m3, rows at dk_weight_pitch_fp8) + "<name>.wscale" (fp32) instead of "<name>.weight"."""
    dev = torch.device(device)
    bf = torch.bfloat16
    fp8 = cfg.weight_dtype == "fp8_e4m3"
    out: Dict[str, Tensor] = {}
    get = (lambda k: w.pop(k)) if consume else (lambda k: w[k])

    if fp8:
        from .config import validate_fp8_policy
        validate_fp8_policy(cfg)
    n_bf = int(getattr(cfg, "fp8_bf16_double_blocks", 0)) if fp8 else 0
    keep_bf16 = tuple(f"multimodal_transformer_blocks.{i}." for i in range(min(n_bf, cfg.depth_multimodal)))  # precision policy

    def is_fp8(name):
        return fp8 and not (keep_bf16 and name.startswith(keep_bf16))

    def put(name, t):
        if is_fp8(name) and name.endswith(".weight") and name[:-len(".weight")].endswith(BLOCK_LINEARS):
            base = name[:-len(".weight")]
            q, scale = quantize_weight_e4m3(t.to(device=dev, dtype=bf))  # the bf16 weight is what gets quantised
            k = q.shape[1]
            pitch = int(_lib.load().dk_weight_pitch_fp8(k))
            if pitch != k:
                padded = torch.zeros(q.shape[0], pitch, dtype=torch.uint8, device=dev)
                padded[:, :k] = q
                q = padded
            out[base + ".weight_fp8"] = q.contiguous()
            out[base + ".wscale"] = scale.contiguous()
            return
        out[name] = t.to(device=dev, dtype=bf).contiguous()

    def put_pitched(name, t):
        """Long-reduction weights ([h, 4h] fc2, [h, 5h] linear2): rows at the engine's pitch (dk_weight_pitch, include/dk_hip.h)."""
        if is_fp8(name):
            return put(name, t)  # the fp8 branch of put() applies dk_weight_pitch_fp8
        k = t.shape[1]
        pitch = int(_lib.load().dk_weight_pitch(k))
        if pitch != k:
            padded = torch.zeros(t.shape[0], pitch, dtype=bf, device=dev)
            padded[:, :k] = t.to(device=dev, dtype=bf)
            t = padded
        put(name, t)

    xw = get("x_embedder.proj.weight")
    put("x_embedder.proj.weight", xw.reshape(xw.shape[0], -1))
    put("x_embedder.proj.bias", get("x_embedder.proj.bias"))
    if "x_pos_embedder.pos_embed.weight" in w:
        put("x_pos_embedder.pos_embed.weight", get("x_pos_embedder.pos_embed.weight"))
    for k in ("context_embedder.weight", "context_embedder.bias",
              "y_embedder.mlp.layers.0.weight", "y_embedder.mlp.layers.0.bias",
              "y_embedder.mlp.layers.2.weight", "y_embedder.mlp.layers.2.bias",
              "t_embedder.mlp.layers.0.weight", "t_embedder.mlp.layers.0.bias",
              "t_embedder.mlp.layers.2.weight", "t_embedder.mlp.layers.2.bias",
              "final_layer.linear.weight", "final_layer.linear.bias"):
        put(k, get(k))
    if cfg.guidance_embed:
        for k in ("guidance_in.mlp.layers.0.weight", "guidance_in.mlp.layers.0.bias", "guidance_in.mlp.layers.2.weight",
                  "guidance_in.mlp.layers.2.bias"):
            put(k, get(k))

    put("adaLN.weight", torch.cat([get(n + ".adaLN_modulation.layers.1.weight").to(dev) for n in adaln_order(cfg)], dim=0))
    put("adaLN.bias", torch.cat([get(n + ".adaLN_modulation.layers.1.bias").to(dev) for n in adaln_order(cfg)], dim=0))

    def stream(prefix, single=False, skip_post=False):
        q, k, v = (get(f"{prefix}.attn.{n}_proj.weight").to(dev) for n in "qkv")
        qb, vb = get(f"{prefix}.attn.q_proj.bias").to(dev), get(f"{prefix}.attn.v_proj.bias").to(dev)
        if single:
            # single-stream blocks: q/k/v and fc1 read the same modulated activations (mmdit.py:693-751;
            # the BFL checkpoint stores them as one "linear1", model_io.py:224-252) -> one [7h, h] matrix
            put(prefix + ".linear1.weight", torch.cat([q, k, v, get(prefix + ".mlp.fc1.weight").to(dev)], dim=0))
            put(prefix + ".linear1.bias", torch.cat([qb, torch.zeros_like(qb), vb, get(prefix + ".mlp.fc1.bias").to(dev)], dim=0))
        else:
            put(prefix + ".attn.qkv.weight", torch.cat([q, k, v], dim=0))
            put(prefix + ".attn.qkv.bias", torch.cat([qb, torch.zeros_like(qb), vb], dim=0))
        if cfg.use_qk_norm:
            put(prefix + ".qk_norm.q_norm.weight", get(prefix + ".qk_norm.q_norm.weight"))
            put(prefix + ".qk_norm.k_norm.weight", get(prefix + ".qk_norm.k_norm.weight"))
        if skip_post:
            return
        if not single:
            put(prefix + ".mlp.fc1.weight", get(prefix + ".mlp.fc1.weight"))
            put(prefix + ".mlp.fc1.bias", get(prefix + ".mlp.fc1.bias"))
        if single:
            put_pitched(prefix + ".linear2.weight",
                        torch.cat([get(prefix + ".attn.o_proj.weight").to(dev), get(prefix + ".mlp.fc2.weight").to(dev)], dim=1))
            put(prefix + ".linear2.bias", get(prefix + ".attn.o_proj.bias"))
        else:
            put(f"{prefix}.attn.o_proj.weight", get(f"{prefix}.attn.o_proj.weight"))
            put(f"{prefix}.attn.o_proj.bias", get(f"{prefix}.attn.o_proj.bias"))
            put_pitched(f"{prefix}.mlp.fc2.weight", get(f"{prefix}.mlp.fc2.weight"))
            put(f"{prefix}.mlp.fc2.bias", get(f"{prefix}.mlp.fc2.bias"))

    for i in range(cfg.depth_multimodal):
        skip_txt = (i == cfg.depth_multimodal - 1) and cfg.depth_unified < 1
        stream(f"multimodal_transformer_blocks.{i}.image_transformer_block")
        stream(f"multimodal_transformer_blocks.{i}.text_transformer_block", skip_post=skip_txt)
    for i in range(cfg.depth_unified):
        stream(f"unified_transformer_blocks.{i}.transformer_block", single=True)
    return out


def pack_vae(cfg, w: Dict[str, Tensor], device) -> Dict[str, Tensor]:
    """Decoder or encoder half: conv weights [O,3,3,I] -> [O, 9*I] (conv_in zero-padded to I = 64)."""
    dev = torch.device(device)
    out: Dict[str, Tensor] = {}
    for k, t in w.items():
        t = t.to(dev)
        if k == "conv_in.weight":  # pad input channels to 64 for the implicit-GEMM K tiling
            o, kh, kw, i = t.shape
            tp = torch.zeros(o, kh, kw, 64, dtype=t.dtype, device=dev)
            tp[..., :i] = t
            t = tp
        if t.dim() == 4:
            t = t.reshape(t.shape[0], -1)
        out[k] = t.to(torch.bfloat16).contiguous()
    # channel-changing resnets (vae.py:86-89,98-99): [conv2 | conv_shortcut] along the reduction, for the fused stage that runs the
    # 1x1 shortcut as extra K-tiles of conv2 (csrc/conv_halo.hip); the separate tensors stay for the unfused path
    for k in list(out):
        if k.endswith(".conv_shortcut.weight"):
            stem = k[:-len(".conv_shortcut.weight")]
            out[stem + ".conv2_sc.weight"] = torch.cat([out[stem + ".conv2.weight"], out[k]], dim=1).contiguous()
    return out


def blob_pack(tensors: Dict[str, Tensor]):
    """Flatten a tensor dict into one contiguous byte blob + index (for a single RCCL broadcast).  Tensors keep their dtype
    (bf16 weights, uint8 e4m3 weights, fp32 scales): the blob is raw bytes, every tensor 256-byte aligned."""
    index, off = [], 0
    for k in sorted(tensors):
        t = tensors[k]
        n = t.numel() * t.element_size()
        index.append((k, tuple(t.shape), str(t.dtype).replace("torch.", ""), off, n))
        off += (n + 255) // 256 * 256
    dev = next(iter(tensors.values())).device
    blob = torch.zeros(off, dtype=torch.uint8, device=dev)
    for k, shape, dt, o, n in index:
        blob[o:o + n] = tensors[k].contiguous().reshape(-1).view(torch.uint8)
    return blob, index


def blob_unpack(blob: Tensor, index) -> Dict[str, Tensor]:
    return {k: blob[o:o + n].view(getattr(torch, dt)).view(*shape) for k, shape, dt, o, n in index}
