"""Generates tests/golden/reference_torch_{vae,mmdit}.npz by RUNNING THE REFERENCE'S OWN PyTorch modules in this container:
/root/reference/python/src/diffusionkit/torch/{vae,mmdit,model_io}.py (the SD3 VAE decoder and MMDiT the reference keeps for its
Core ML export, plus its checkpoint-key adjustments).  Run from the repo root:  python tests/golden/make_reference_torch_fixtures.py

The reference modules import three packages that are not installed here (argmaxtools, beartype, jaxtyping).  They are replaced by
the stand-ins below -- nothing of the reference is copied, its files are imported where they lie:

  beartype.typing.Tuple, jaxtyping.Float      annotations only
  argmaxtools.utils.get_logger                logging.getLogger
  argmaxtools._sdpa.Cat(embed_dim, n_heads).sdpa(query, key, value, causal, key_padding_mask)
                                              softmax(q^T k / sqrt(d)) v per head on [B, C, 1, S] tensors, channels head-major
  argmaxtools.nn.LayerNorm(C, eps, elementwise_affine=False)   layer norm over the channel axis of [B, C, 1, S]
  argmaxtools.nn.Attention(embed_dim, n_heads, attention_type) container of q_proj / k_proj (no bias) / v_proj / o_proj 1x1 convs
  argmaxtools.nn.FFN(embed_dim, expansion_factor, activation_fn)   fc2(act(fc1(x))), 1x1 convs
  (the reference's own load_state_dict(strict) of the checkpoint its key adjustments produce confirms those parameter names)

So the VAE decoder runs entirely on the reference's code and torch.nn except for the attention product, and the MMDiT runs the
reference's wiring (timestep embedding, adaLN chunk order, affine_transform, joint sequence, last-block text skip, positional
embedding crop, final layer, unpatchify, the Stability checkpoint key map) over standard layers.  The fixtures hold the seed, the
tensor names and shapes of the synthetic checkpoints, a checksum of the generated weights, the inputs and the reference outputs;
tests/test_reference_torch_golden.py regenerates the checkpoints from the seed, sends them through THIS repository's checkpoint
loaders and oracle (and, on a GPU, the HIP engines) and compares.  /root/reference is not needed to run the tests.
"""
import json
import logging
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests._util import seeded_checkpoint, checkpoint_checksum  # noqa: E402


def install_stand_ins():
    class _Sub:
        def __class_getitem__(cls, item):
            return cls

    bt, btt = types.ModuleType("beartype"), types.ModuleType("beartype.typing")
    import typing
    btt.Tuple = typing.Tuple
    bt.typing = btt
    jt = types.ModuleType("jaxtyping")
    jt.Float = _Sub
    ax, axu, axs, axn = (types.ModuleType(n) for n in ("argmaxtools", "argmaxtools.utils", "argmaxtools._sdpa", "argmaxtools.nn"))
    axu.get_logger = logging.getLogger

    class Cat:
        def __init__(self, embed_dim, n_heads):
            self.n_heads, self.d = n_heads, embed_dim // n_heads

        def sdpa(self, query, key, value, causal=False, key_padding_mask=None):
            assert not causal and key_padding_mask is None
            b, c, _, s = query.shape
            q, k, v = (t.reshape(b, self.n_heads, self.d, t.shape[-1]) for t in (query, key, value))
            a = torch.softmax(torch.einsum("bhdq,bhdk->bhqk", q, k) / math.sqrt(self.d), dim=-1)
            return torch.einsum("bhqk,bhdk->bhdq", a, v).reshape(b, c, 1, s)

    axs.Cat = Cat

    class LayerNorm(nn.Module):
        def __init__(self, num_channels, eps=1e-5, elementwise_affine=True):
            super().__init__()
            assert not elementwise_affine
            self.eps = eps

        def forward(self, x):
            mu = x.mean(dim=1, keepdim=True)
            var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
            return (x - mu) * torch.rsqrt(var + self.eps)

    class AttentionType:
        SelfAttention = "self"

    class Attention(nn.Module):
        def __init__(self, embed_dim, n_heads, attention_type):
            super().__init__()
            self.q_proj = nn.Conv2d(embed_dim, embed_dim, 1)
            self.k_proj = nn.Conv2d(embed_dim, embed_dim, 1, bias=False)
            self.v_proj = nn.Conv2d(embed_dim, embed_dim, 1)
            self.o_proj = nn.Conv2d(embed_dim, embed_dim, 1)

    class FFN(nn.Module):
        def __init__(self, embed_dim, expansion_factor, activation_fn):
            super().__init__()
            self.fc1 = nn.Conv2d(embed_dim, expansion_factor * embed_dim, 1)
            self.act = activation_fn
            self.fc2 = nn.Conv2d(expansion_factor * embed_dim, embed_dim, 1)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    axn.LayerNorm, axn.Attention, axn.AttentionType, axn.FFN = LayerNorm, Attention, AttentionType, FFN
    ax.utils, ax._sdpa, ax.nn = axu, axs, axn
    for m in (bt, btt, jt, ax, axu, axs, axn):
        sys.modules[m.__name__] = m


def compvis_vae_spec(model):
    """(name, shape) of the CompVis-layout checkpoint the reference's vae_decoder_state_dict_adjustments turns into this module's
    state dict: its renames undone (attn_1.{q,k,v}_proj -> q,k,v, out_proj -> proj_out), 'first_stage_model.decoder.' in front."""
    spec = []
    for k, v in model.state_dict().items():
        for a, b in ((".attn_1.q_proj", ".attn_1.q"), (".attn_1.k_proj", ".attn_1.k"), (".attn_1.v_proj", ".attn_1.v"),
                     (".attn_1.out_proj", ".attn_1.proj_out")):
            k = k.replace(a, b)
        spec.append(("first_stage_model.decoder." + k, tuple(v.shape)))
    return spec


def stability_mmdit_spec(cfg):
    """(name, shape) of a Stability-layout SD3 checkpoint ('model.diffusion_model.' prefix) for the reference's MMDiTConfig."""
    h, p, r = cfg.hidden_size, cfg.patch_size, cfg.mlp_ratio
    pre = "model.diffusion_model."
    spec = [("pos_embed", (1, cfg.max_latent_resolution ** 2, h)),
            ("x_embedder.proj.weight", (h, cfg.vae_latent_dim, p, p)), ("x_embedder.proj.bias", (h,)),
            ("context_embedder.weight", (h, cfg.token_level_text_embed_dim)), ("context_embedder.bias", (h,))]
    for emb, d_in in (("y_embedder", cfg.pooled_text_embed_dim), ("t_embedder", cfg.frequency_embed_dim)):
        spec += [(f"{emb}.mlp.0.weight", (h, d_in)), (f"{emb}.mlp.0.bias", (h,)), (f"{emb}.mlp.2.weight", (h, h)), (f"{emb}.mlp.2.bias", (h,))]
    for i in range(cfg.depth):
        for blk in ("x_block", "context_block"):
            b = f"joint_blocks.{i}.{blk}"
            last_ctx = blk == "context_block" and i == cfg.depth - 1
            spec += [(f"{b}.attn.qkv.weight", (3 * h, h)), (f"{b}.attn.qkv.bias", (3 * h,))]
            n_mod = 2 if last_ctx else 6
            spec += [(f"{b}.adaLN_modulation.1.weight", (n_mod * h, h)), (f"{b}.adaLN_modulation.1.bias", (n_mod * h,))]
            if not last_ctx:
                spec += [(f"{b}.attn.proj.weight", (h, h)), (f"{b}.attn.proj.bias", (h,)),
                         (f"{b}.mlp.fc1.weight", (r * h, h)), (f"{b}.mlp.fc1.bias", (r * h,)),
                         (f"{b}.mlp.fc2.weight", (h, r * h)), (f"{b}.mlp.fc2.bias", (h,))]
    spec += [("final_layer.linear.weight", (p * p * cfg.vae_latent_dim, h)), ("final_layer.linear.bias", (p * p * cfg.vae_latent_dim,)),
             ("final_layer.adaLN_modulation.1.weight", (2 * h, h)), ("final_layer.adaLN_modulation.1.bias", (2 * h,))]
    return [(pre + k, s) for k, s in spec]


def main():
    install_stand_ins()
    sys.path.insert(0, "/root/reference/python/src")
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    import diffusionkit.torch.mmdit as rmm
    import diffusionkit.torch.model_io as rio
    import diffusionkit.torch.vae as rvae

    # ---- VAE decoder: channels (64, 64, 128, 128), 3 resnets per level, latent 8 x 12 -> image 64 x 96 ----
    vcfg = rvae.VAEDecoderConfig(resolution=64, base_channels=64, channel_multipliers=[1, 1, 2, 2])
    vae = rvae.VAEDecoder(vcfg).eval()
    spec = compvis_vae_spec(vae)
    seed = 20240924
    ckpt = seeded_checkpoint(spec, seed)
    vae.load_state_dict(rio.vae_decoder_state_dict_adjustments(dict(ckpt)))
    z = torch.randn(2, 16, 8, 12, generator=torch.Generator().manual_seed(seed + 1))
    img = vae(z)
    np.savez_compressed(os.path.join(HERE, "reference_torch_vae.npz"), spec=json.dumps(spec), seed=seed, checksum=checkpoint_checksum(ckpt),
                        z=z.numpy(), image=img.numpy(), channels=np.array([64, 64, 128, 128]), group_norm_eps=1e-6)
    print("vae:", tuple(img.shape), "checksum", checkpoint_checksum(ckpt))

    # ---- SD3 MMDiT: depth 2 (hidden 128, 2 heads of 64), latent 8 x 12, 20 text tokens, batch 2 ----
    mcfg = rmm.MMDiTConfig(depth=2, max_latent_resolution=16)
    model = rmm.MMDiT(mcfg).eval()
    spec = stability_mmdit_spec(mcfg)
    ckpt = seeded_checkpoint(spec, seed + 2)
    # the reference's loader strips the first two name components and drops VAE tensors, then adjusts (model_io.py:60-72)
    stripped = {".".join(k.rsplit(".")[2:]): v for k, v in ckpt.items()}
    model.load_state_dict(rio.mmdit_state_dict_adjustments(stripped))  # strict: names and shapes of the stand-in layers agree
    g = torch.Generator().manual_seed(seed + 3)
    latent = torch.randn(2, 16, 8, 12, generator=g)
    text = torch.randn(2, mcfg.token_level_text_embed_dim, 1, 20, generator=g)
    pooled = torch.randn(2, mcfg.pooled_text_embed_dim, 1, 1, generator=g)
    t = torch.tensor([857.5, 857.5])
    (out,) = model(latent, text, pooled, t)
    np.savez_compressed(os.path.join(HERE, "reference_torch_mmdit.npz"), spec=json.dumps(spec), seed=seed + 2, checksum=checkpoint_checksum(ckpt),
                        latent=latent.numpy(), text=text.numpy(), pooled=pooled.numpy(), timestep=t.numpy(), out=out.numpy(),
                        depth=mcfg.depth, max_latent_resolution=mcfg.max_latent_resolution)
    print("mmdit:", tuple(out.shape), "checksum", checkpoint_checksum(ckpt))


if __name__ == "__main__":
    main()
