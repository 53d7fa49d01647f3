"""Checkpoint loader: BFL FLUX.1 / Stability SD3 safetensors -> the reference's module-tree names.

SURVEY.md §8(f) row f1.  The reference remaps checkpoint keys with chains of ``str.replace`` over the
whole state dict (python/src/diffusionkit/mlx/model_io.py:130-311 FLUX, :314-408 SD3, :411-486 VAE
decoder).  Here the same mapping is an explicit table of (pattern -> target) rules, applied once per
key, plus the four tensor transformations the reference performs:

  * fused ``qkv`` projections are split into q / k / v (:141-153, :361-374); ``k_proj.bias`` does not
    exist in the reference module (mmdit.py:820-821) and is dropped (:389-390; FLUX drops it in
    ``model.update``);
  * FLUX single blocks: ``linear1`` -> q, k, v, fc1 (:224-252); ``linear2`` -> o_proj | fc2 along the
    input axis, its bias becomes o_proj's (the fc2 copy is zeroed on every call, mmdit.py:741-742:
    quirk Q8, so only one bias is kept);
  * conv weights OIHW -> OHWI (:398-400, :455-484), 1x1 convs (VAE attention q/k/v/proj_out,
    nin_shortcut) -> Linear weights;
  * SD3 ``pos_embed`` [1, T, h] -> ``x_pos_embedder.pos_embed.weight`` [T, h] (:392-396).

The result is the dict ``diffusionkit_amd.weights.pack_mmdit`` / ``pack_vae`` consume, i.e. exactly
what ``synth_mmdit_weights`` / ``synth_vae_weights`` produce for the benchmarks.  FLUX.1-dev's
``guidance_in`` is ignored, as in the reference (quirk Q7).
"""
from __future__ import annotations

import re
from typing import Callable, Dict, List, Optional, Tuple

import torch

from .config import MMDiTConfig, VAEDecoderConfig

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


class CheckpointError(ValueError):
    pass


def load_safetensors(path: str) -> StateDict:
    from safetensors.torch import load_file
    return load_file(path, device="cpu")


# ---------------------------------------------------------------------------------------------
# FLUX.1 (Black Forest Labs key layout)
# ---------------------------------------------------------------------------------------------
_FLUX_STREAM = {"img": "image_transformer_block", "txt": "text_transformer_block"}
_FLUX_TOP = {
    "img_in": "x_embedder.proj", "txt_in": "context_embedder",
    "time_in.in_layer": "t_embedder.mlp.layers.0", "time_in.out_layer": "t_embedder.mlp.layers.2",
    "vector_in.in_layer": "y_embedder.mlp.layers.0", "vector_in.out_layer": "y_embedder.mlp.layers.2",
    "final_layer.linear": "final_layer.linear", "final_layer.adaLN_modulation.1": "final_layer.adaLN_modulation.layers.1",
}


def flux_checkpoint_to_reference(sd: StateDict, cfg: MMDiTConfig) -> StateDict:
    """BFL ``flux1-{schnell,dev}.safetensors`` keys -> reference MMDiT names."""
    h, r = cfg.hidden_size, cfg.mlp_ratio
    out: StateDict = {}
    used = set()

    def take(k):
        used.add(k)
        return sd[k]

    for k in sd:
        m = re.fullmatch(r"(.+)\.(weight|bias)", k)
        stem, leaf = (m.group(1), m.group(2)) if m else (k, "")
        if stem in _FLUX_TOP:
            out[f"{_FLUX_TOP[stem]}.{leaf}"] = take(k)
            continue
        if stem.startswith("guidance_in."):
            # FLUX.1-dev guidance embedding (MLPEmbedder in_layer / out_layer -> mlp.layers.0 / .2, as the reference's
            # flux_state_dict_adjustments renames every in_layer / out_layer, model_io.py:292-298).  The reference then runs dev on
            # the schnell preset (quirk Q7) and never reads them: kept only when the configuration asks for the embedding
            if cfg.guidance_embed and stem in ("guidance_in.in_layer", "guidance_in.out_layer"):
                out[f"guidance_in.mlp.layers.{0 if stem.endswith('in_layer') else 2}.{leaf}"] = take(k)
            else:
                used.add(k)
            continue
        m = re.fullmatch(r"double_blocks\.(\d+)\.(img|txt)_attn\.norm\.(query|key)_norm\.scale", k)
        if m:
            base = f"multimodal_transformer_blocks.{m.group(1)}.{_FLUX_STREAM[m.group(2)]}"
            out[f"{base}.qk_norm.{m.group(3)[0]}_norm.weight"] = take(k)
            continue
        m = re.fullmatch(r"single_blocks\.(\d+)\.norm\.(query|key)_norm\.scale", k)
        if m:
            out[f"unified_transformer_blocks.{m.group(1)}.transformer_block.qk_norm.{m.group(2)[0]}_norm.weight"] = take(k)
            continue
        m = re.fullmatch(r"double_blocks\.(\d+)\.(img|txt)_(.+)", stem)
        if m:
            base = f"multimodal_transformer_blocks.{m.group(1)}.{_FLUX_STREAM[m.group(2)]}"
            part = m.group(3)
            t = take(k)
            if part == "mod.lin":
                out[f"{base}.adaLN_modulation.layers.1.{leaf}"] = t
            elif part == "attn.qkv":
                q, kk, v = torch.chunk(t, 3, dim=0)
                out[f"{base}.attn.q_proj.{leaf}"] = q
                out[f"{base}.attn.v_proj.{leaf}"] = v
                if leaf == "weight":
                    out[f"{base}.attn.k_proj.weight"] = kk
            elif part == "attn.proj":
                out[f"{base}.attn.o_proj.{leaf}"] = t
            elif part == "mlp.0":
                out[f"{base}.mlp.fc1.{leaf}"] = t
            elif part == "mlp.2":
                out[f"{base}.mlp.fc2.{leaf}"] = t
            else:
                raise CheckpointError(f"unknown FLUX double-block tensor: {k}")
            continue
        m = re.fullmatch(r"single_blocks\.(\d+)\.(linear1|linear2|modulation\.lin)", stem)
        if m:
            base = f"unified_transformer_blocks.{m.group(1)}.transformer_block"
            t = take(k)
            if m.group(2) == "modulation.lin":
                out[f"{base}.adaLN_modulation.layers.1.{leaf}"] = t
            elif m.group(2) == "linear1":
                q, kk, v, fc1 = torch.split(t, [h, h, h, r * h], dim=0)
                out[f"{base}.attn.q_proj.{leaf}"] = q
                out[f"{base}.attn.v_proj.{leaf}"] = v
                if leaf == "weight":
                    out[f"{base}.attn.k_proj.weight"] = kk
                out[f"{base}.mlp.fc1.{leaf}"] = fc1
            else:  # linear2 over [attention | gelu(fc1)]
                if leaf == "weight":
                    o, fc2 = torch.split(t, [h, r * h], dim=1)
                    out[f"{base}.attn.o_proj.weight"] = o.contiguous()
                    out[f"{base}.mlp.fc2.weight"] = fc2.contiguous()
                else:
                    out[f"{base}.attn.o_proj.bias"] = t  # the fc2 copy is zeroed at run time (Q8)
            continue
        raise CheckpointError(f"unknown FLUX checkpoint key: {k}")
    w = out.get("x_embedder.proj.weight")
    if w is None:
        raise CheckpointError("img_in.weight missing")
    out["x_embedder.proj.weight"] = w.reshape(w.shape[0], 1, 1, w.shape[1])  # Linear 64->h as a 1x1 conv (:306-308)
    _check_mmdit_complete(out, cfg)
    return out


# ---------------------------------------------------------------------------------------------
# SD3 / SD3.5 (Stability key layout, prefix "model.diffusion_model.")
# ---------------------------------------------------------------------------------------------
_SD3_STREAM = {"x_block": "image_transformer_block", "context_block": "text_transformer_block"}


def sd3_checkpoint_to_reference(sd: StateDict, cfg: MMDiTConfig, prefix: str = "model.diffusion_model.") -> StateDict:
    """``sd3_medium.safetensors``-style keys -> reference MMDiT names (VAE / teacher tensors are skipped)."""
    out: StateDict = {}
    for k0, t in sd.items():
        if not k0.startswith(prefix):
            continue  # first_stage_model.*, text encoders, teacher_model.* ...
        k = k0[len(prefix):]
        if k == "pos_embed":
            out["x_pos_embedder.pos_embed.weight"] = t[0]
            continue
        m = re.fullmatch(r"(.+)\.(weight|bias)", k)
        if not m:
            raise CheckpointError(f"unknown SD3 checkpoint key: {k0}")
        stem, leaf = m.group(1), m.group(2)
        if stem == "x_embedder.proj":
            out[f"x_embedder.proj.{leaf}"] = t.permute(0, 2, 3, 1).contiguous() if leaf == "weight" else t
        elif stem == "context_embedder" or stem == "final_layer.linear":
            out[f"{stem}.{leaf}"] = t
        elif (m3 := re.fullmatch(r"([ty]_embedder)\.mlp\.([02])", stem)):
            out[f"{m3.group(1)}.mlp.layers.{m3.group(2)}.{leaf}"] = t
        elif stem == "final_layer.adaLN_modulation.1":
            out[f"final_layer.adaLN_modulation.layers.1.{leaf}"] = t
        else:
            m2 = re.fullmatch(r"joint_blocks\.(\d+)\.(x_block|context_block)\.(.+)", stem)
            if not m2:
                raise CheckpointError(f"unknown SD3 checkpoint key: {k0}")
            base = f"multimodal_transformer_blocks.{m2.group(1)}.{_SD3_STREAM[m2.group(2)]}"
            part = m2.group(3)
            if part == "attn.qkv":
                q, kk, v = torch.chunk(t, 3, dim=0)
                out[f"{base}.attn.q_proj.{leaf}"] = q
                out[f"{base}.attn.v_proj.{leaf}"] = v
                if leaf == "weight":
                    out[f"{base}.attn.k_proj.weight"] = kk
            elif part == "attn.proj":
                out[f"{base}.attn.o_proj.{leaf}"] = t
            elif part in ("attn.ln_q", "attn.ln_k"):
                out[f"{base}.qk_norm.{part[-1]}_norm.{leaf}"] = t
            elif part in ("mlp.fc1", "mlp.fc2"):
                out[f"{base}.{part}.{leaf}"] = t
            elif part == "adaLN_modulation.1":
                out[f"{base}.adaLN_modulation.layers.1.{leaf}"] = t
            else:
                raise CheckpointError(f"unknown SD3 block tensor: {k0}")
    _check_mmdit_complete(out, cfg)
    return out


def _check_mmdit_complete(out: StateDict, cfg: MMDiTConfig) -> None:
    """Every tensor pack_mmdit needs must be present with the right shape (fail loudly, name the key)."""
    from .weights import mmdit_weight_shapes
    want = mmdit_weight_shapes(cfg)
    missing = sorted(set(want) - set(out))
    if missing:
        raise CheckpointError(f"checkpoint lacks {len(missing)} tensors, first: {missing[:3]}")
    for k, shp in want.items():
        if tuple(out[k].shape) != shp:
            raise CheckpointError(f"{k}: shape {tuple(out[k].shape)} != expected {shp}")
    extra = sorted(set(out) - set(want))
    if extra:
        raise CheckpointError(f"checkpoint produced unexpected tensors: {extra[:3]}")


# ---------------------------------------------------------------------------------------------
# VAE decoder (CompVis layout; prefix "first_stage_model.decoder." in SD3 files, "decoder." in ae.safetensors)
# ---------------------------------------------------------------------------------------------
def vae_decoder_checkpoint_to_reference(sd: StateDict, cfg: VAEDecoderConfig, prefix: str = "decoder.") -> StateDict:
    out: StateDict = {}

    def conv(t):  # OIHW -> OHWI
        return t.permute(0, 2, 3, 1).contiguous()

    for k0, t in sd.items():
        if not k0.startswith(prefix):
            continue  # encoder.*, model.diffusion_model.*, text encoders ...
        k = k0[len(prefix):]
        m = re.fullmatch(r"(.+)\.(weight|bias)", k)
        if not m:
            continue
        stem, leaf = m.group(1), m.group(2)
        w = leaf == "weight"
        if stem in ("conv_in", "conv_out"):
            out[f"{stem}.{leaf}"] = conv(t) if w else t
        elif stem == "norm_out":
            out[f"conv_norm_out.{leaf}"] = t
        elif (m2 := re.fullmatch(r"mid\.block_([12])\.(norm1|conv1|norm2|conv2)", stem)):
            idx = 0 if m2.group(1) == "1" else 2
            out[f"mid_blocks.{idx}.{m2.group(2)}.{leaf}"] = conv(t) if (w and "conv" in m2.group(2)) else t
        elif (m2 := re.fullmatch(r"mid\.attn_1\.(norm|q|k|v|proj_out)", stem)):
            name = {"norm": "group_norm", "q": "query_proj", "k": "key_proj", "v": "value_proj", "proj_out": "out_proj"}[m2.group(1)]
            out[f"mid_blocks.1.{name}.{leaf}"] = t[:, :, 0, 0].contiguous() if (w and name != "group_norm") else t
        elif (m2 := re.fullmatch(r"up\.(\d+)\.block\.(\d+)\.(norm1|conv1|norm2|conv2|nin_shortcut)", stem)):
            base = f"up_blocks.{m2.group(1)}.resnets.{m2.group(2)}"
            part = m2.group(3)
            if part == "nin_shortcut":
                out[f"{base}.conv_shortcut.{leaf}"] = t[:, :, 0, 0].contiguous() if w else t
            else:
                out[f"{base}.{part}.{leaf}"] = conv(t) if (w and "conv" in part) else t
        elif (m2 := re.fullmatch(r"up\.(\d+)\.upsample\.conv", stem)):
            out[f"up_blocks.{m2.group(1)}.upsample.{leaf}"] = conv(t) if w else t
        else:
            raise CheckpointError(f"unknown VAE decoder key: {k0}")
    from .weights import vae_weight_shapes
    want = vae_weight_shapes(cfg)
    missing = sorted(set(want) - set(out))
    if missing:
        raise CheckpointError(f"VAE checkpoint lacks {len(missing)} tensors, first: {missing[:3]}")
    for k, shp in want.items():
        if tuple(out[k].shape) != shp:
            raise CheckpointError(f"{k}: shape {tuple(out[k].shape)} != expected {shp}")
    return {k: out[k] for k in want}


def vae_encoder_checkpoint_to_reference(sd: StateDict, cfg, prefix: str = "encoder.") -> StateDict:
    """Stability / BFL autoencoder ``encoder.*`` tensors -> reference VAEEncoder names and layouts
    (behaviour of model_io.py:489-563; written as an explicit table, not chained substring replaces)."""
    out: StateDict = {}

    def conv(t):  # OIHW -> OHWI
        return t.permute(0, 2, 3, 1).contiguous()

    for k0, t in sd.items():
        if not k0.startswith(prefix):
            continue
        k = k0[len(prefix):]
        m = re.fullmatch(r"(.+)\.(weight|bias)", k)
        if not m:
            continue
        stem, leaf = m.group(1), m.group(2)
        w = leaf == "weight"
        if stem in ("conv_in", "conv_out"):
            out[f"{stem}.{leaf}"] = conv(t) if w else t
        elif stem == "norm_out":
            out[f"conv_norm_out.{leaf}"] = t
        elif (m2 := re.fullmatch(r"mid\.block_([12])\.(norm1|conv1|norm2|conv2)", stem)):
            idx = 0 if m2.group(1) == "1" else 2
            out[f"mid_blocks.{idx}.{m2.group(2)}.{leaf}"] = conv(t) if (w and "conv" in m2.group(2)) else t
        elif (m2 := re.fullmatch(r"mid\.attn_1\.(norm|q|k|v|proj_out)", stem)):
            name = {"norm": "group_norm", "q": "query_proj", "k": "key_proj", "v": "value_proj", "proj_out": "out_proj"}[m2.group(1)]
            out[f"mid_blocks.1.{name}.{leaf}"] = t[:, :, 0, 0].contiguous() if (w and name != "group_norm") else t
        elif (m2 := re.fullmatch(r"down\.(\d+)\.block\.(\d+)\.(norm1|conv1|norm2|conv2|nin_shortcut)", stem)):
            base = f"down_blocks.{m2.group(1)}.resnets.{m2.group(2)}"
            part = m2.group(3)
            if part == "nin_shortcut":
                out[f"{base}.conv_shortcut.{leaf}"] = t[:, :, 0, 0].contiguous() if w else t
            else:
                out[f"{base}.{part}.{leaf}"] = conv(t) if (w and "conv" in part) else t
        elif (m2 := re.fullmatch(r"down\.(\d+)\.downsample\.conv", stem)):
            out[f"down_blocks.{m2.group(1)}.downsample.{leaf}"] = conv(t) if w else t
        else:
            raise CheckpointError(f"unknown VAE encoder key: {k0}")
    from .weights import vae_encoder_weight_shapes
    want = vae_encoder_weight_shapes(cfg)
    missing = sorted(set(want) - set(out))
    if missing:
        raise CheckpointError(f"VAE checkpoint lacks {len(missing)} encoder tensors, first: {missing[:3]}")
    for k, shp in want.items():
        if tuple(out[k].shape) != shp:
            raise CheckpointError(f"{k}: shape {tuple(out[k].shape)} != expected {shp}")
    return {k: out[k] for k in want}


# ---------------------------------------------------------------------------------------------
# text encoders: Hugging Face CLIPTextModel(WithProjection) / T5EncoderModel checkpoints
# ---------------------------------------------------------------------------------------------
def clip_checkpoint_to_reference(sd: StateDict, cfg) -> StateDict:
    """HF ``text_model.*`` CLIP tensors -> reference CLIPTextModel names (behaviour of model_io.py:611-636)."""
    out: StateDict = {}
    for k0, t in sd.items():
        k = k0[len("text_model."):] if k0.startswith("text_model.") else k0
        if k.endswith("position_ids"):
            continue  # registered buffer of older transformers versions
        if k.startswith("embeddings."):
            k = k[len("embeddings."):]
        if k.startswith("encoder."):
            k = k[len("encoder."):]
        m = re.fullmatch(r"layers\.(\d+)\.(self_attn\.(q|k|v|out)_proj|mlp\.fc([12])|layer_norm([12]))\.(weight|bias)", k)
        if m:
            i, leaf = m.group(1), m.group(6)
            if m.group(3):
                name = {"q": "query_proj", "k": "key_proj", "v": "value_proj", "out": "out_proj"}[m.group(3)]
                out[f"layers.{i}.attention.{name}.{leaf}"] = t
            elif m.group(4):
                out[f"layers.{i}.linear{m.group(4)}.{leaf}"] = t
            else:
                out[f"layers.{i}.layer_norm{m.group(5)}.{leaf}"] = t
        elif k in ("token_embedding.weight", "position_embedding.weight", "final_layer_norm.weight", "final_layer_norm.bias",
                   "text_projection.weight"):
            out[k] = t
        else:
            raise CheckpointError(f"unknown CLIP text-encoder key: {k0}")
    from .text import synth_clip_weights
    want = synth_clip_weights(cfg, shapes_only=True)
    _check_against(out, want, "CLIP")
    return {k: out[k] for k in want}


def t5_checkpoint_to_reference(sd: StateDict, cfg) -> StateDict:
    """HF T5 encoder tensors -> reference SD3T5Encoder names (behaviour of model_io.py:565-608)."""
    out: StateDict = {}
    attn = {"q": "query_proj", "k": "key_proj", "v": "value_proj", "o": "out_proj"}
    for k, t in sd.items():
        if k in ("shared.weight",) or k.startswith("decoder.") or k.startswith("lm_head."):
            continue
        if k == "encoder.embed_tokens.weight":
            out["wte.weight"] = t
        elif k == "encoder.final_layer_norm.weight":
            out["encoder.ln.weight"] = t
        elif k == "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight":
            out["encoder.relative_attention_bias.embeddings.weight"] = t
        elif (m := re.fullmatch(r"encoder\.block\.(\d+)\.layer\.0\.SelfAttention\.(q|k|v|o)\.weight", k)):
            out[f"encoder.layers.{m.group(1)}.attention.{attn[m.group(2)]}.weight"] = t
        elif (m := re.fullmatch(r"encoder\.block\.(\d+)\.layer\.(0|1)\.layer_norm\.weight", k)):
            out[f"encoder.layers.{m.group(1)}.ln{int(m.group(2)) + 1}.weight"] = t
        elif (m := re.fullmatch(r"encoder\.block\.(\d+)\.layer\.1\.DenseReluDense\.(wi_0|wi_1|wo)\.weight", k)):
            out[f"encoder.layers.{m.group(1)}.dense.{m.group(2)}.weight"] = t
        else:
            raise CheckpointError(f"unknown T5 encoder key: {k}")
    if "wte.weight" not in out and "shared.weight" in sd:
        out["wte.weight"] = sd["shared.weight"]
    from .text import synth_t5_weights
    want = synth_t5_weights(cfg, shapes_only=True)
    _check_against(out, want, "T5")
    return {k: out[k] for k in want}


def _check_against(out: StateDict, want, what: str) -> None:
    missing = sorted(set(want) - set(out))
    if missing:
        raise CheckpointError(f"{what} checkpoint lacks {len(missing)} tensors, first: {missing[:3]}")
    for k, shp in want.items():
        if tuple(out[k].shape) != shp:
            raise CheckpointError(f"{k}: shape {tuple(out[k].shape)} != expected {shp}")


# ---------------------------------------------------------------------------------------------
# MLX group-quantised checkpoints (the reference's ``*-4bit-quantized`` model versions)
# ---------------------------------------------------------------------------------------------
MLX_QUANT_GROUP = 64  # nn.quantize defaults (mlx/model_io.py:731-733,775 call it without arguments)
MLX_QUANT_BITS = 4


def dequantize_mlx(wq: Tensor, scales: Tensor, biases: Tensor, group_size: int = MLX_QUANT_GROUP, bits: int = MLX_QUANT_BITS) -> Tensor:
    """``mx.dequantize``: uint32 packs [out, in * bits / 32] + per-group scales / biases [out, in / group_size] -> bf16 [out, in];
    element j of a pack sits in bits [bits * j, bits * (j + 1)), w = scale * q + bias evaluated in fp32 (oracle/mlxquant.py restates
    the contract; MLX itself is absent here: the bit layout is unpinned)."""
    if bits not in (2, 4, 8):
        raise CheckpointError(f"MLX quantisation with {bits} bits is not supported (2, 4, 8)")
    per = 32 // bits
    if wq.dtype not in (torch.uint32, torch.int32, torch.int64):
        raise CheckpointError(f"quantised weight of dtype {wq.dtype}, expected uint32 packs")
    out_f, packs = wq.shape
    n_in = packs * per
    if n_in % group_size != 0 or tuple(scales.shape) != (out_f, n_in // group_size) or tuple(biases.shape) != tuple(scales.shape):
        raise CheckpointError(f"quantised weight {tuple(wq.shape)} does not match scales {tuple(scales.shape)} at group size {group_size}")
    w64 = wq.to(torch.int64) & 0xFFFFFFFF
    shifts = torch.arange(per, dtype=torch.int64) * bits
    q = ((w64[:, :, None] >> shifts) & ((1 << bits) - 1)).reshape(out_f, n_in).to(torch.float32)
    w = q.reshape(out_f, -1, group_size) * scales.float()[:, :, None] + biases.float()[:, :, None]
    return w.reshape(out_f, n_in).to(torch.bfloat16)


def mlx_quantized_checkpoint_to_reference(sd: StateDict, group_size: int = MLX_QUANT_GROUP, bits: int = MLX_QUANT_BITS) -> StateDict:
    """The reference's ``...-4bit-quantized`` checkpoints hold MLX module-tree names already ("4-bit ckpt already adjusted",
    mlx/model_io.py:772-775; SD3.5-large under the prefix ``model.diffusion_model.``, :728-734) with every ``nn.Linear`` stored as
    ``weight`` (uint32 packs) / ``scales`` / ``biases`` (+ the Linear's own ``bias``).  The reference keeps them quantised and
    multiplies through ``mx.quantized_matmul``; this engine dequantises once at load (bf16, or on to e4m3 when
    ``MMDiTConfig.weight_dtype`` asks for the fp8 MFMA path): same values as ``mx.dequantize`` of the stored triplets."""
    prefix = "model.diffusion_model."
    if any(k.startswith(prefix) for k in sd):
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    out: StateDict = {}
    for k, t in sd.items():
        if k.endswith(".scales") or k.endswith(".biases"):
            continue
        stem = k[: -len(".weight")] if k.endswith(".weight") else None
        if stem is not None and f"{stem}.scales" in sd:
            if f"{stem}.biases" not in sd:
                raise CheckpointError(f"{stem}: quantised weight without biases")
            out[k] = dequantize_mlx(t, sd[f"{stem}.scales"], sd[f"{stem}.biases"], group_size, bits)
        else:
            out[k] = t
    return out


# ---------------------------------------------------------------------------------------------
# entry points used by the pipelines (local_ckpt={"mmdit": path_or_dict, "vae_decoder": path_or_dict})
# ---------------------------------------------------------------------------------------------
def load_mmdit_checkpoint(src, cfg: MMDiTConfig) -> StateDict:
    """``src``: a dict already in reference names, a raw checkpoint dict, or a .safetensors path."""
    sd = load_safetensors(src) if isinstance(src, str) else dict(src)
    if any(k.endswith(".scales") for k in sd):  # an MLX nn.quantize checkpoint (the *-4bit-quantized model versions)
        sd = mlx_quantized_checkpoint_to_reference(sd)
        _check_mmdit_complete(sd, cfg)
        return sd
    if any(k.startswith("multimodal_transformer_blocks.") for k in sd):
        return sd  # already remapped
    if any(k.startswith("double_blocks.") for k in sd):
        return flux_checkpoint_to_reference(sd, cfg)
    for pre in ("model.diffusion_model.", ""):
        if any(k.startswith(pre + "joint_blocks.") for k in sd):
            return sd3_checkpoint_to_reference(sd, cfg, prefix=pre)
    raise CheckpointError("unrecognised MMDiT checkpoint layout (neither BFL FLUX nor Stability SD3 keys)")


def load_vae_decoder_checkpoint(src, cfg: VAEDecoderConfig) -> StateDict:
    sd = load_safetensors(src) if isinstance(src, str) else dict(src)
    if any(k.startswith("up_blocks.") for k in sd):
        return sd
    pre = "first_stage_model.decoder." if any(k.startswith("first_stage_model.decoder.") for k in sd) else "decoder."
    return vae_decoder_checkpoint_to_reference(sd, cfg, prefix=pre)


def load_vae_encoder_checkpoint(src, cfg) -> StateDict:
    sd = load_safetensors(src) if isinstance(src, str) else dict(src)
    if any(k.startswith("down_blocks.") for k in sd):
        return sd
    pre = "first_stage_model.encoder." if any(k.startswith("first_stage_model.encoder.") for k in sd) else "encoder."
    return vae_encoder_checkpoint_to_reference(sd, cfg, prefix=pre)


def load_clip_checkpoint(src, cfg) -> StateDict:
    """``src``: a .safetensors path or state dict, Hugging Face CLIPTextModel(WithProjection) layout or reference names."""
    sd = load_safetensors(src) if isinstance(src, str) else dict(src)
    if any(k.startswith("layers.") for k in sd):
        return sd
    return clip_checkpoint_to_reference(sd, cfg)


def load_t5_checkpoint(src, cfg) -> StateDict:
    sd = load_safetensors(src) if isinstance(src, str) else dict(src)
    if "wte.weight" in sd:
        return sd
    return t5_checkpoint_to_reference(sd, cfg)

