"""Replays outputs of THE REFERENCE'S OWN PyTorch modules (python/src/diffusionkit/torch/{vae,mmdit,model_io}.py, executed in the
build container by tests/golden/make_reference_torch_fixtures.py -- see its header for the three absent packages it stands in
for) against this repository: the synthetic CompVis / Stability checkpoints are regenerated from the fixture's seed, sent through
diffusionkit_amd.model_io (the f1 checkpoint loaders) and the oracle, and -- on an MI355X -- through the HIP engines.

What this pins: the VAE decoder end to end (every op but the attention product is the reference's torch code), the MMDiT's wiring
(timestep embedding, adaLN chunk order, affine_transform, joint attention, last-block text skip, positional-embedding crop, final
layer, unpatchify) and both checkpoint key maps.  What it cannot pin: the MLX-only parts (FLUX blocks, RoPE, QK-norm, the exact-erf
GELU the MLX path uses where the PyTorch module uses the tanh form, MLX's rounding points)."""
import json
import os

import numpy as np
import pytest
import torch

from diffusionkit_amd.config import SD3_2b, VAEDecoderConfig
from diffusionkit_amd.model_io import load_mmdit_checkpoint, load_vae_decoder_checkpoint
from oracle.mmdit import OracleMMDiT, Prec
from oracle.vae import OracleVAEDecoder
from tests._util import checkpoint_checksum, psnr, rel_l2, seeded_checkpoint

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    f = np.load(os.path.join(GOLD, name), allow_pickle=False)
    spec = [(k, tuple(s)) for k, s in json.loads(str(f["spec"]))]
    ckpt = seeded_checkpoint(spec, int(f["seed"]))
    assert abs(checkpoint_checksum(ckpt) - float(f["checksum"])) < 1e-6 * abs(float(f["checksum"])), "seeded checkpoint drifted"
    return f, ckpt


def _vae_case():
    f, ckpt = _load("reference_torch_vae.npz")
    cfg = VAEDecoderConfig(block_out_channels=tuple(int(c) for c in f["channels"]), layers_per_block=3,
                           group_norm_eps=float(f["group_norm_eps"]))
    named = load_vae_decoder_checkpoint(ckpt, cfg)  # CompVis keys ('first_stage_model.decoder.') -> reference names
    z = torch.from_numpy(f["z"]).permute(0, 2, 3, 1).contiguous()  # NCHW -> NHWC
    want = torch.from_numpy(f["image"]).permute(0, 2, 3, 1).contiguous()
    return cfg, named, z, want


def _mmdit_case():
    from dataclasses import replace
    f, ckpt = _load("reference_torch_mmdit.npz")
    depth = int(f["depth"])  # the PyTorch config ties width and heads to the depth: hidden = 64 * depth, heads = depth
    cfg = replace(SD3_2b, num_heads=depth, depth_multimodal=depth, hidden_size_override=64 * depth,
                  max_latent_resolution=int(f["max_latent_resolution"]))
    named = load_mmdit_checkpoint(ckpt, cfg)  # Stability keys ('model.diffusion_model.') -> reference names
    lat = torch.from_numpy(f["latent"]).permute(0, 2, 3, 1).contiguous()
    text = torch.from_numpy(f["text"])[:, :, 0, :].transpose(1, 2).contiguous()  # [B, C, 1, S] -> [B, S, C]
    pooled = torch.from_numpy(f["pooled"])[:, :, 0, 0].contiguous()
    want = torch.from_numpy(f["out"]).permute(0, 2, 3, 1).contiguous()
    return cfg, named, lat, text, pooled, float(f["timestep"][0]), want


def test_oracle_vae_decoder_matches_reference_torch_module():
    cfg, named, z, want = _vae_case()
    got = OracleVAEDecoder(cfg, {k: v.float() for k, v in named.items()}, Prec())(z)
    assert got.shape == want.shape
    assert rel_l2(want, got) < 5e-6, rel_l2(want, got)


def test_oracle_mmdit_matches_reference_torch_module():
    cfg, named, lat, text, pooled, t, want = _mmdit_case()
    m = OracleMMDiT(cfg, {k: v.float() for k, v in named.items()}, Prec(), gelu="tanh")
    m.cache_modulation_params(pooled, torch.tensor([t]))
    got = m(lat, text, t)
    assert got.shape == want.shape
    err = rel_l2(want, got)
    assert err < 5e-6, err
    # the exact-erf GELU of the MLX path is a measurably different function (the switch is not a no-op)
    m2 = OracleMMDiT(cfg, {k: v.float() for k, v in named.items()}, Prec())
    m2.cache_modulation_params(pooled, torch.tensor([t]))
    assert rel_l2(want, m2(lat, text, t)) > 5.0 * err


@pytest.mark.gpu
def test_hip_vae_decoder_matches_reference_torch_module():
    """The HIP VAE decoder (bf16) against the image the reference's PyTorch decoder produced from the same checkpoint."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from diffusionkit_amd.engine import VAEDecoderEngine
    from diffusionkit_amd.weights import pack_vae
    cfg, named, z, want = _vae_case()
    dev = torch.device("cuda", 0)
    eng = VAEDecoderEngine(cfg, pack_vae(cfg, named, dev))
    img, _, _ = eng.decode(z.to(dev))  # the engine returns the reference's (x + 1) / 2 image, clipped
    want01 = ((want + 1.0) / 2.0).clamp(0, 1)
    assert psnr(want01, img.float().cpu()) > 35.0


@pytest.mark.gpu
def test_hip_mmdit_matches_reference_torch_module():
    """The HIP MMDiT engine (bf16, exact-erf GELU) against the output of the reference's PyTorch MMDiT (fp32, tanh GELU: a 6e-5
    relative difference, far below the bf16 noise) on the same Stability-layout checkpoint."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from diffusionkit_amd.engine import MMDiTEngine
    from diffusionkit_amd.weights import pack_mmdit
    cfg, named, lat, text, pooled, t, want = _mmdit_case()
    dev = torch.device("cuda", 0)
    eng = MMDiTEngine(cfg, pack_mmdit(cfg, named, dev))
    B, Hl, Wl, _ = lat.shape
    eng.prepare(B, (Hl, Wl), text.shape[1], 1)
    eng.cache_modulation_params(pooled.to(dev), [t])
    tok = eng.forward_tokens(eng.patchify(lat.to(dev)), text.to(dev, torch.bfloat16), 0)
    got = OracleMMDiT(cfg, {}, Prec())._unpatch(tok.float().cpu(), Hl, Wl)  # a pure reshape (mmdit.py:975-988)
    # the bf16-emulating oracle on the same inputs is the yardstick for what bf16 costs at this size
    emu = OracleMMDiT(cfg, {k: v.float() for k, v in named.items()}, Prec(torch.bfloat16))
    emu.cache_modulation_params(pooled, torch.tensor([t]))
    e_emu, e_hip = rel_l2(want, emu(lat, text, t)), rel_l2(want, got)
    assert e_hip <= 2.0 * e_emu + 2e-3, (e_hip, e_emu)
    assert psnr(want, got) > 35.0
