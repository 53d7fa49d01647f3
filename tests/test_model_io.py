"""Checkpoint loader: BFL FLUX / Stability SD3 / CompVis-VAE key layouts -> the
reference's module-tree names.  No real checkpoints exist in this environment, so the tests build
checkpoints in the upstream layouts from seeded reference-named weights with an independent inverse
of the reference's remapping (python/src/diffusionkit/mlx/model_io.py:130-486: qkv / linear1 / linear2
fusion, OIHW convs, 1x1-conv attention projections, pos_embed buffer), write them with safetensors and
require the loader to give back the reference-named tensors bit for bit."""
import os

import pytest
import torch

from diffusionkit_amd import model_io as mio
from diffusionkit_amd.config import tiny_flux, tiny_sd3, tiny_vae, tiny_vae_encoder
from diffusionkit_amd.weights import synth_mmdit_weights, synth_vae_encoder_weights, synth_vae_weights


def to_bfl_flux(w, cfg):
    """reference names -> BFL flux1-*.safetensors layout (inverse of flux_state_dict_adjustments)."""
    h = cfg.hidden_size
    sd = {}
    top = {"x_embedder.proj": "img_in", "context_embedder": "txt_in", "t_embedder.mlp.layers.0": "time_in.in_layer",
           "t_embedder.mlp.layers.2": "time_in.out_layer", "y_embedder.mlp.layers.0": "vector_in.in_layer",
           "y_embedder.mlp.layers.2": "vector_in.out_layer", "final_layer.linear": "final_layer.linear",
           "final_layer.adaLN_modulation.layers.1": "final_layer.adaLN_modulation.1"}
    for ref, bfl in top.items():
        for leaf in ("weight", "bias"):
            t = w[f"{ref}.{leaf}"]
            sd[f"{bfl}.{leaf}"] = t.reshape(h, -1) if (ref == "x_embedder.proj" and leaf == "weight") else t
    zeros = lambda t: torch.zeros_like(t)
    for i in range(cfg.depth_multimodal):
        for s_ref, s_bfl in (("image_transformer_block", "img"), ("text_transformer_block", "txt")):
            b = f"multimodal_transformer_blocks.{i}.{s_ref}"
            o = f"double_blocks.{i}.{s_bfl}"
            sd[f"{o}_attn.qkv.weight"] = torch.cat([w[f"{b}.attn.{n}_proj.weight"] for n in "qkv"], 0)
            qb = w[f"{b}.attn.q_proj.bias"]
            sd[f"{o}_attn.qkv.bias"] = torch.cat([qb, zeros(qb) + 0.5, w[f"{b}.attn.v_proj.bias"]], 0)  # k bias must be dropped
            sd[f"{o}_attn.norm.query_norm.scale"] = w[f"{b}.qk_norm.q_norm.weight"]
            sd[f"{o}_attn.norm.key_norm.scale"] = w[f"{b}.qk_norm.k_norm.weight"]
            for leaf in ("weight", "bias"):
                sd[f"{o}_attn.proj.{leaf}"] = w[f"{b}.attn.o_proj.{leaf}"]
                sd[f"{o}_mlp.0.{leaf}"] = w[f"{b}.mlp.fc1.{leaf}"]
                sd[f"{o}_mlp.2.{leaf}"] = w[f"{b}.mlp.fc2.{leaf}"]
                sd[f"{o}_mod.lin.{leaf}"] = w[f"{b}.adaLN_modulation.layers.1.{leaf}"]
    for i in range(cfg.depth_unified):
        b = f"unified_transformer_blocks.{i}.transformer_block"
        o = f"single_blocks.{i}"
        sd[f"{o}.linear1.weight"] = torch.cat([w[f"{b}.attn.{n}_proj.weight"] for n in "qkv"] + [w[f"{b}.mlp.fc1.weight"]], 0)
        qb = w[f"{b}.attn.q_proj.bias"]
        sd[f"{o}.linear1.bias"] = torch.cat([qb, zeros(qb) - 0.25, w[f"{b}.attn.v_proj.bias"], w[f"{b}.mlp.fc1.bias"]], 0)
        sd[f"{o}.linear2.weight"] = torch.cat([w[f"{b}.attn.o_proj.weight"], w[f"{b}.mlp.fc2.weight"]], 1)
        sd[f"{o}.linear2.bias"] = w[f"{b}.attn.o_proj.bias"]
        sd[f"{o}.norm.query_norm.scale"] = w[f"{b}.qk_norm.q_norm.weight"]
        sd[f"{o}.norm.key_norm.scale"] = w[f"{b}.qk_norm.k_norm.weight"]
        for leaf in ("weight", "bias"):
            sd[f"{o}.modulation.lin.{leaf}"] = w[f"{b}.adaLN_modulation.layers.1.{leaf}"]
    return {k: v.contiguous() for k, v in sd.items()}


def to_sai_sd3(w, cfg, prefix="model.diffusion_model."):
    """reference names -> sd3_medium.safetensors layout (inverse of mmdit_state_dict_adjustments)."""
    sd = {}
    sd["x_embedder.proj.weight"] = w["x_embedder.proj.weight"].permute(0, 3, 1, 2)  # OHWI -> OIHW
    sd["x_embedder.proj.bias"] = w["x_embedder.proj.bias"]
    sd["pos_embed"] = w["x_pos_embedder.pos_embed.weight"][None]
    for e in ("t_embedder", "y_embedder"):
        for l in (0, 2):
            for leaf in ("weight", "bias"):
                sd[f"{e}.mlp.{l}.{leaf}"] = w[f"{e}.mlp.layers.{l}.{leaf}"]
    for leaf in ("weight", "bias"):
        sd[f"context_embedder.{leaf}"] = w[f"context_embedder.{leaf}"]
        sd[f"final_layer.linear.{leaf}"] = w[f"final_layer.linear.{leaf}"]
        sd[f"final_layer.adaLN_modulation.1.{leaf}"] = w[f"final_layer.adaLN_modulation.layers.1.{leaf}"]
    for i in range(cfg.depth_multimodal):
        for s_ref, s_sai in (("image_transformer_block", "x_block"), ("text_transformer_block", "context_block")):
            b = f"multimodal_transformer_blocks.{i}.{s_ref}"
            o = f"joint_blocks.{i}.{s_sai}"
            sd[f"{o}.attn.qkv.weight"] = torch.cat([w[f"{b}.attn.{n}_proj.weight"] for n in "qkv"], 0)
            qb = w[f"{b}.attn.q_proj.bias"]
            sd[f"{o}.attn.qkv.bias"] = torch.cat([qb, torch.ones_like(qb), w[f"{b}.attn.v_proj.bias"]], 0)
            for leaf in ("weight", "bias"):
                sd[f"{o}.adaLN_modulation.1.{leaf}"] = w[f"{b}.adaLN_modulation.layers.1.{leaf}"]
                if f"{b}.attn.o_proj.{leaf}" in w:  # the last text block is pre-only
                    sd[f"{o}.attn.proj.{leaf}"] = w[f"{b}.attn.o_proj.{leaf}"]
                    sd[f"{o}.mlp.fc1.{leaf}"] = w[f"{b}.mlp.fc1.{leaf}"]
                    sd[f"{o}.mlp.fc2.{leaf}"] = w[f"{b}.mlp.fc2.{leaf}"]
    out = {prefix + k: v.contiguous() for k, v in sd.items()}
    out["first_stage_model.encoder.conv_in.weight"] = torch.zeros(2, 2)  # unrelated tensors in the same file are skipped
    out["text_encoders.clip_l.dummy.weight"] = torch.zeros(2, 2)
    return out


def to_compvis_vae(w, cfg, prefix="decoder."):
    """reference names -> CompVis decoder layout (inverse of vae_decoder_state_dict_adjustments)."""
    sd = {}
    oihw = lambda t: t.permute(0, 3, 1, 2)
    for k, t in w.items():
        stem, leaf = k.rsplit(".", 1)
        wt = leaf == "weight"
        if stem in ("conv_in", "conv_out"):
            sd[k] = oihw(t) if wt else t
        elif stem == "conv_norm_out":
            sd[f"norm_out.{leaf}"] = t
        elif stem.startswith("mid_blocks.1."):
            part = stem.split(".")[-1]
            name = {"group_norm": "norm", "query_proj": "q", "key_proj": "k", "value_proj": "v", "out_proj": "proj_out"}[part]
            sd[f"mid.attn_1.{name}.{leaf}"] = t[:, :, None, None] if (wt and part != "group_norm") else t
        elif stem.startswith("mid_blocks."):
            idx, part = stem.split(".")[1], stem.split(".")[2]
            sd[f"mid.block_{1 if idx == '0' else 2}.{part}.{leaf}"] = oihw(t) if (wt and "conv" in part) else t
        elif ".resnets." in stem:
            _, j, _, r, part = stem.split(".")
            if part == "conv_shortcut":
                sd[f"up.{j}.block.{r}.nin_shortcut.{leaf}"] = t[:, :, None, None] if wt else t
            else:
                sd[f"up.{j}.block.{r}.{part}.{leaf}"] = oihw(t) if (wt and "conv" in part) else t
        elif stem.endswith(".upsample"):
            j = stem.split(".")[1]
            sd[f"up.{j}.upsample.conv.{leaf}"] = oihw(t) if wt else t
        else:
            raise AssertionError(k)
    out = {prefix + k: v.contiguous() for k, v in sd.items()}
    out["encoder.conv_in.weight"] = torch.zeros(2, 2)
    return out


def same(a, b):
    assert set(a) == set(b), (sorted(set(a) - set(b))[:3], sorted(set(b) - set(a))[:3])
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_flux_checkpoint_round_trip(tmp_path):
    from safetensors.torch import save_file
    cfg = tiny_flux()
    w = synth_mmdit_weights(cfg, seed=5)
    ck = to_bfl_flux(w, cfg)
    ck["guidance_in.in_layer.weight"] = torch.zeros(4, 4)  # FLUX.1-dev extra: ignored like the reference does (Q7)
    path = os.path.join(tmp_path, "flux1-tiny.safetensors")
    save_file(ck, path)
    got = mio.load_mmdit_checkpoint(path, cfg)
    same(got, w)
    assert not any("k_proj.bias" in k for k in got)
    # x_embedder is a 1x1 conv over the 64 patch features (reference expands dims at load)
    assert got["x_embedder.proj.weight"].shape == (cfg.hidden_size, 1, 1, cfg.patch_dim)


def test_flux_dev_checkpoint_keeps_guidance_embedding(tmp_path):
    """guidance_embed = True (FLUX_DEV preset): guidance_in.{in,out}_layer -> guidance_in.mlp.layers.{0,2}; the packer finds them
    (the loader used to drop every guidance_in.* key, so a real FLUX.1-dev checkpoint could not be packed)."""
    from dataclasses import replace
    from safetensors.torch import save_file
    from diffusionkit_amd.weights import mmdit_weight_shapes
    cfg = replace(tiny_flux(), guidance_embed=True)
    w = synth_mmdit_weights(cfg, seed=9)
    ck = to_bfl_flux(w, cfg)
    for ref, bfl in (("guidance_in.mlp.layers.0", "guidance_in.in_layer"), ("guidance_in.mlp.layers.2", "guidance_in.out_layer")):
        for leaf in ("weight", "bias"):
            ck[f"{bfl}.{leaf}"] = w[f"{ref}.{leaf}"].contiguous()
    path = os.path.join(tmp_path, "flux1-dev-tiny.safetensors")
    save_file(ck, path)
    got = mio.load_mmdit_checkpoint(path, cfg)
    same(got, w)
    assert set(got) == set(mmdit_weight_shapes(cfg))
    # the same file under the schnell preset (what the reference selects for dev, quirk Q7): the four tensors are ignored
    got_s = mio.load_mmdit_checkpoint(path, tiny_flux())
    assert not any(k.startswith("guidance_in.") for k in got_s)


def test_sd3_checkpoint_round_trip(tmp_path):
    from safetensors.torch import save_file
    cfg = tiny_sd3()
    w = synth_mmdit_weights(cfg, seed=6)
    path = os.path.join(tmp_path, "sd3-tiny.safetensors")
    save_file(to_sai_sd3(w, cfg), path)
    same(mio.load_mmdit_checkpoint(path, cfg), w)


def test_vae_checkpoint_round_trip(tmp_path):
    from safetensors.torch import save_file
    cfg = tiny_vae()
    w = synth_vae_weights(cfg, seed=7)
    for prefix in ("decoder.", "first_stage_model.decoder."):
        path = os.path.join(tmp_path, "vae.safetensors")
        save_file(to_compvis_vae(w, cfg, prefix), path)
        same(mio.load_vae_decoder_checkpoint(path, cfg), w)


def to_compvis_vae_encoder(w, prefix):
    """reference VAEEncoder names -> CompVis autoencoder ``encoder.*`` layout (inverse of
    vae_encoder_state_dict_adjustments, model_io.py:489-563)."""
    oihw = lambda t: t.permute(0, 3, 1, 2)
    sd = {}
    for k, t in w.items():
        stem, leaf = k.rsplit(".", 1)
        wt = leaf == "weight"
        if stem in ("conv_in", "conv_out"):
            sd[k] = oihw(t) if wt else t
        elif stem == "conv_norm_out":
            sd[f"norm_out.{leaf}"] = t
        elif stem.startswith("mid_blocks.1."):
            part = stem.split(".")[-1]
            name = {"group_norm": "norm", "query_proj": "q", "key_proj": "k", "value_proj": "v", "out_proj": "proj_out"}[part]
            sd[f"mid.attn_1.{name}.{leaf}"] = t[:, :, None, None] if (wt and part != "group_norm") else t
        elif stem.startswith("mid_blocks."):
            idx, part = stem.split(".")[1], stem.split(".")[2]
            sd[f"mid.block_{1 if idx == '0' else 2}.{part}.{leaf}"] = oihw(t) if (wt and "conv" in part) else t
        elif ".resnets." in stem:
            _, j, _, r, part = stem.split(".")
            if part == "conv_shortcut":
                sd[f"down.{j}.block.{r}.nin_shortcut.{leaf}"] = t[:, :, None, None] if wt else t
            else:
                sd[f"down.{j}.block.{r}.{part}.{leaf}"] = oihw(t) if (wt and "conv" in part) else t
        elif stem.endswith(".downsample"):
            sd[f"down.{stem.split('.')[1]}.downsample.conv.{leaf}"] = oihw(t) if wt else t
        else:
            raise AssertionError(k)
    out = {prefix + k: v.contiguous() for k, v in sd.items()}
    out["decoder.conv_in.weight"] = torch.zeros(2, 2)  # the other half of the autoencoder file: ignored
    return out


def test_vae_encoder_checkpoint_round_trip(tmp_path):
    """f4: the img2img encoder half (model_io.py:489-563)."""
    from safetensors.torch import save_file
    cfg = tiny_vae_encoder()
    w = synth_vae_encoder_weights(cfg, seed=8)
    for prefix in ("encoder.", "first_stage_model.encoder."):
        path = os.path.join(tmp_path, "vae_enc.safetensors")
        save_file(to_compvis_vae_encoder(w, prefix), path)
        same(mio.load_vae_encoder_checkpoint(path, cfg), w)
    bad = to_compvis_vae_encoder(w, "encoder.")
    del bad["encoder.down.0.downsample.conv.bias"]
    with pytest.raises(mio.CheckpointError):
        mio.load_vae_encoder_checkpoint(bad, cfg)


def test_loader_fails_loudly():
    cfg = tiny_flux()
    ck = to_bfl_flux(synth_mmdit_weights(cfg, seed=5), cfg)
    bad = dict(ck)
    bad.pop("single_blocks.0.linear2.weight")
    with pytest.raises(mio.CheckpointError, match="lacks"):
        mio.load_mmdit_checkpoint(bad, cfg)
    bad = dict(ck)
    bad["double_blocks.0.img_attn.rope.weight"] = torch.zeros(2)
    with pytest.raises(mio.CheckpointError, match="unknown"):
        mio.load_mmdit_checkpoint(bad, cfg)
    bad = dict(ck)
    bad["txt_in.weight"] = torch.zeros(3, 3)
    with pytest.raises(mio.CheckpointError, match="shape"):
        mio.load_mmdit_checkpoint(bad, cfg)
    with pytest.raises(mio.CheckpointError, match="unrecognised"):
        mio.load_mmdit_checkpoint({"foo.weight": torch.zeros(1)}, cfg)


def test_reference_named_dict_passes_through():
    cfg = tiny_flux()
    w = synth_mmdit_weights(cfg, seed=5)
    assert mio.load_mmdit_checkpoint(w, cfg) is not None
    same(mio.load_mmdit_checkpoint(w, cfg), w)


# ---- MLX group-quantised checkpoints: the reference's *-4bit-quantized model versions (mlx/model_io.py:728-734,772-775) --------
def to_mlx_quantized(w, prefix="", fast=True):
    """reference-named weights -> an nn.quantize checkpoint: every Linear weight (2-D, input dimension a multiple of the group) becomes
    weight (uint32 packs) / scales / biases, everything else (biases, norms, the patch conv, pos-emb) stays.  Returns the checkpoint
    and, per quantised key, the fp32 matrix mx.dequantize gives back (oracle/mlxquant.py, element loops)."""
    import numpy as np
    from oracle import mlxquant as mq
    ck, deq = {}, {}
    for k, t in w.items():
        if k.endswith(".weight") and t.dim() == 2 and t.shape[1] % 64 == 0 and "pos_embed" not in k and "norm" not in k:
            wq, sc, bi = mq.quantize(t.float().numpy())
            sc16, bi16 = torch.from_numpy(sc).to(torch.float16), torch.from_numpy(bi).to(torch.float16)  # the reference stores them in its 16-bit dtype
            stem = k[: -len(".weight")]
            ck[prefix + k] = torch.from_numpy(wq.astype(np.int64)).to(torch.uint32)
            ck[prefix + stem + ".scales"], ck[prefix + stem + ".biases"] = sc16, bi16
            deq[k] = torch.from_numpy(mq.dequantize(wq, sc16.float().numpy(), bi16.float().numpy()))
        else:
            ck[prefix + k] = t.contiguous()
    return ck, deq


@pytest.mark.parametrize("family", ["flux", "sd35"])
def test_mlx_4bit_quantized_checkpoint_loads(tmp_path, family):
    """a ...-4bit-quantized file (already in module-tree names; SD3.5 under model.diffusion_model.) comes back as bf16 weights equal
    to mx.dequantize of its triplets, every other tensor bit for bit"""
    from dataclasses import replace
    from safetensors.torch import save_file
    cfg = tiny_flux(depth_multimodal=1, depth_unified=1) if family == "flux" else replace(tiny_sd3(depth=1), use_qk_norm=True)
    w = synth_mmdit_weights(cfg, seed=11)
    ck, deq = to_mlx_quantized(w, "" if family == "flux" else "model.diffusion_model.")
    assert len(deq) >= 10
    path = os.path.join(tmp_path, f"{family}-4bit-quantized.safetensors")
    save_file(ck, path)
    got = mio.load_mmdit_checkpoint(path, cfg)
    assert set(got) == set(w)
    for k in w:
        if k in deq:
            assert got[k].dtype == torch.bfloat16
            assert torch.equal(got[k], deq[k].to(torch.bfloat16)), k
            # and the 4-bit grid is what it should be: within half a step of the original weight
            assert (got[k].float() - w[k].float()).abs().max() <= 0.5 * (w[k].float().max() - w[k].float().min()) / 15 + 1e-2
        else:
            assert torch.equal(got[k], w[k]), k


def test_mlx_dequantize_matches_oracle_loops_and_rejects_bad_shapes():
    import numpy as np
    from oracle import mlxquant as mq
    g = torch.Generator().manual_seed(3)
    for bits in (2, 4, 8):
        w = torch.randn(5, 128, generator=g).numpy()
        wq, sc, bi = mq.quantize(w, group_size=32, bits=bits)
        ref = torch.from_numpy(mq.dequantize(wq, sc, bi, 32, bits))
        got = mio.dequantize_mlx(torch.from_numpy(wq.astype(np.int64)).to(torch.uint32), torch.from_numpy(sc), torch.from_numpy(bi), 32, bits)
        assert torch.equal(got, ref.to(torch.bfloat16))
    # a pack whose top element has its high bit set (sign bit of the int32 view)
    wq = np.array([[0xF0000001]], np.uint32)
    ref = mq.dequantize(wq, np.ones((1, 1), np.float32), np.zeros((1, 1), np.float32), 8, 4)
    got = mio.dequantize_mlx(torch.from_numpy(wq.astype(np.int64)).to(torch.uint32), torch.ones(1, 1), torch.zeros(1, 1), 8, 4)
    assert got.float().tolist() == ref.tolist() == [[1.0, 0, 0, 0, 0, 0, 0, 15.0]]
    with pytest.raises(mio.CheckpointError, match="scales"):
        mio.dequantize_mlx(torch.zeros(4, 8, dtype=torch.uint32), torch.ones(4, 2), torch.zeros(4, 2))
    with pytest.raises(mio.CheckpointError, match="biases"):
        mio.mlx_quantized_checkpoint_to_reference({"a.weight": torch.zeros(4, 8, dtype=torch.uint32), "a.scales": torch.ones(4, 1)})
