"""Generates tests/golden/reference_mlx_*.npz by RUNNING THE REFERENCE'S OWN MLX MODEL CODE
(/root/reference/python/src/diffusionkit/mlx/{config,mmdit,sampler}.py, imported where they lie, unmodified) on top of
tests/golden/mlx_standin.py -- a PyTorch-backed stand-in for the few dozen MLX operations those files use (MLX itself is Apple-only
and absent here).  Run from the repo root:  python tests/golden/make_reference_mlx_fixtures.py

What executing the reference this way pins: the MMDiT's wiring for FLUX (double + single-stream blocks, RoPE tables and their
application, QK-norm, the [text, image] joint order, the fused-bias quirk of the single blocks, reshape patchify / unpack), for SD3
(learned positional embedding crop, conv patchify / unpatchify, [image, text] order, the skipped text stream of the last block,
batch 1 = fused LayerNorm-modulate, batch 2 = unfused) and for the SD3.5 shape class (QK-norm without RoPE), the modulation cache
(cache_modulation_params: keyed by timestep.item(), adaLN weights emptied afterwards), and the samplers' sigma / timestep
schedules.  What it does not pin: MLX's own arithmetic (everything runs in float32 here, configs are built with dtype float32).

Weights: seeded synthetic tensors under the reference's module-tree names (tests/_util.seeded_checkpoint over the shapes of
diffusionkit_amd.weights.mmdit_weight_shapes); the fixtures store seed, checksum, inputs and the reference outputs.
"""
import importlib
import json
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import mlx_standin  # noqa: E402
from tests._util import checkpoint_checksum, seeded_checkpoint  # noqa: E402

REF = "/root/reference/python/src/diffusionkit"


def import_reference():
    import huggingface_hub  # noqa: F401
    import transformers  # noqa: F401  (both BEFORE the stand-in exists: transformers probes for an installed mlx when imported)
    mx, _ = mlx_standin.install()
    import typing
    bt, btt = types.ModuleType("beartype"), types.ModuleType("beartype.typing")
    for n in ("Dict", "List", "Optional", "Tuple"):
        setattr(btt, n, getattr(typing, n))
    bt.typing = btt
    ax, axu, axt = (types.ModuleType(n) for n in ("argmaxtools", "argmaxtools.utils", "argmaxtools.test_utils"))
    axu.get_logger = logging.getLogger
    axt.AppleSiliconContextMixin = type("AppleSiliconContextMixin", (), {})  # base classes of a benchmarking helper
    axt.InferenceContextSpec = type("InferenceContextSpec", (), {})
    ax.utils, ax.test_utils = axu, axt
    for m in (bt, btt, ax, axu, axt):
        sys.modules[m.__name__] = m
    sys.path.insert(0, os.path.dirname(REF))
    ref = importlib.import_module("diffusionkit.mlx")  # the reference's real package: pipeline, step loop, loaders, encoders
    mods = {n: importlib.import_module("diffusionkit.mlx." + n) for n in ("config", "mmdit", "sampler", "vae", "clip", "t5", "tokenizer")}
    return mx, ref, mods


def adaln_items(model, mx):
    """[(name, array)] of every adaLN_modulation weight and bias: what load_mmdit(only_modulation_dict=True) hands back to
    CFGDenoiser.clear_cache (mlx/__init__.py:686-689) after cache_modulation_params has emptied them"""
    return [(k, mx.array(v.t.clone())) for k, v in mlx_standin.tree_flatten(model.parameters()) if "adaLN_modulation" in k]


def run_denoise(mx, ref, mods, tag, ours, ref_kwargs, flux, cfg_weight, num_steps, latent_size, S_t, seed, denoise=1.0, encoder=None):
    """DiffusionPipeline.denoise_latents (mlx/__init__.py:253-292) itself -- get_empty_latent / get_noise / get_sigmas /
    noise_scaling / sample_euler / CFGDenoiser / latent format -- on a pipeline object assembled without its loader."""
    from diffusionkit_amd.weights import mmdit_weight_shapes
    rc, rm, rs = mods["config"], mods["mmdit"], mods["sampler"]
    cfg = rc.MMDiTConfig(dtype=mx.float32, float16_dtype=mx.float32, low_memory_mode=False, **ref_kwargs)
    model = rm.MMDiT(cfg)
    spec = sorted((k, tuple(v)) for k, v in mmdit_weight_shapes(ours).items())
    ckpt = seeded_checkpoint(spec, seed)
    set_weights(model, ckpt, mx)
    pipe = object.__new__(ref.FluxPipeline if flux else ref.DiffusionPipeline)  # no __init__: that one downloads checkpoints
    pipe.mmdit = model
    pipe.sampler = rs.FluxSampler(shift=1.0) if flux else rs.ModelSamplingDiscreteFlow(shift=3.0)
    pipe.latent_format = ref.FluxLatentFormat() if flux else ref.SD3LatentFormat()
    pipe.activation_dtype = mx.float32
    saved = adaln_items(model, mx)
    pipe.load_mmdit = lambda only_modulation_dict=False: saved
    g = torch.Generator().manual_seed(seed + 1)
    rows = 2 if cfg_weight > 0 else 1
    cond = torch.randn(rows, S_t, ours.token_level_text_embed_dim, generator=g)
    pooled = torch.randn(rows, ours.pooled_text_embed_dim, generator=g)
    extra = {}
    image_path = None
    if encoder is not None:
        from PIL import Image
        rgb = (torch.rand(latent_size[0] * 8, latent_size[1] * 8, 3, generator=g) * 255).to(torch.uint8).numpy()
        image_path = os.path.join(HERE, f"_tmp_{tag}.png")
        Image.fromarray(rgb).save(image_path)
        pipe.encoder = encoder
        extra["image"] = rgb
    latent, _ = pipe.denoise_latents(mx.array(cond), mx.array(pooled), num_steps=num_steps, cfg_weight=cfg_weight,
                                     latent_size=latent_size, seed=seed, image_path=image_path, denoise=denoise)
    if image_path:
        os.remove(image_path)
    np.savez_compressed(os.path.join(HERE, f"reference_mlx_denoise_{tag}.npz"), spec=json.dumps(spec), seed=seed,
                        checksum=checkpoint_checksum(ckpt), cond=cond.numpy(), pooled=pooled.numpy(), latent=np.asarray(latent),
                        num_steps=num_steps, cfg_weight=cfg_weight, latent_size=np.asarray(latent_size), denoise=denoise, **extra)
    print(f"denoise {tag}: latent {np.asarray(latent).shape}, |latent| mean {np.abs(np.asarray(latent)).mean():.4f}")


def run_vae(mx, tag, model, spec, x, seed):
    ckpt = seeded_checkpoint(spec, seed)
    set_weights(model, ckpt, mx)
    assert len(mlx_standin.tree_flatten(model.parameters())) == len(spec), tag
    out = np.asarray(model(mx.array(x)))
    np.savez_compressed(os.path.join(HERE, f"reference_mlx_{tag}.npz"), spec=json.dumps(spec), seed=seed, checksum=checkpoint_checksum(ckpt),
                        x=x.numpy(), out=out)
    print(f"{tag}: out {out.shape}, |out| mean {np.abs(out).mean():.4f}")


def set_weights(module, ckpt, mx):
    """Assign tensors by the reference's dotted names ('a.b.3.layers.1.weight')."""
    for name, t in ckpt.items():
        parts = name.split(".")
        obj = module
        for p in parts[:-1]:
            obj = obj[int(p)] if p.isdigit() else getattr(obj, p)
        cur = getattr(obj, parts[-1])
        assert tuple(cur.shape) == tuple(t.shape), (name, cur.shape, t.shape)
        setattr(obj, parts[-1], mx.array(t))


def run_case(mx, rc, rm, ours, ref_kwargs, B, latent_hw, S_t, timesteps, step, seed, tag):
    from diffusionkit_amd.weights import mmdit_weight_shapes
    cfg = rc.MMDiTConfig(dtype=mx.float32, float16_dtype=mx.float32, low_memory_mode=False, **ref_kwargs)
    model = rm.MMDiT(cfg)
    spec = sorted((k, tuple(v)) for k, v in mmdit_weight_shapes(ours).items())
    ckpt = seeded_checkpoint(spec, seed)
    set_weights(model, ckpt, mx)
    # the reference module holds one kind of tensor the engine layout does not: mlp.fc2.bias of the single-stream blocks, which the
    # reference multiplies by zero on every call (mmdit.py:741-742).  They get NON-zero values here, so the fixture only matches
    # an implementation that drops them as well.
    extra = sorted({k for k, _ in mlx_standin.tree_flatten(model.parameters())} - {k for k, _ in spec})
    assert all(k.startswith("unified_transformer_blocks.") and k.endswith(".mlp.fc2.bias") for k in extra), extra
    set_weights(model, seeded_checkpoint([(k, (ours.hidden_size,)) for k in extra], seed + 7), mx)
    g = torch.Generator().manual_seed(seed + 1)
    lat = torch.randn(B, latent_hw[0], latent_hw[1], 16, generator=g)
    text = torch.randn(B, S_t, ours.token_level_text_embed_dim, generator=g)
    pooled = torch.randn(B, ours.pooled_text_embed_dim, generator=g)
    ts = mx.array(np.asarray(timesteps, dtype=np.float32))
    model.cache_modulation_params(mx.array(pooled), ts)
    out = model(mx.array(lat), mx.array(text[:, :, None, :]), mx.array(np.full((B,), timesteps[step], dtype=np.float32)))
    out = np.asarray(out)
    np.savez_compressed(os.path.join(HERE, f"reference_mlx_mmdit_{tag}.npz"), spec=json.dumps(spec), seed=seed,
                        checksum=checkpoint_checksum(ckpt), latent=lat.numpy(), text=text.numpy(), pooled=pooled.numpy(),
                        timesteps=np.asarray(timesteps, dtype=np.float32), step=step, out=out)
    print(f"{tag}: out {out.shape}, |out| mean {np.abs(out).mean():.4f}, checksum {checkpoint_checksum(ckpt):.6f}")


def main():
    torch.set_grad_enabled(False)
    mx, ref, mods = import_reference()
    rc, rm, rs = mods["config"], mods["mmdit"], mods["sampler"]
    from dataclasses import replace
    from diffusionkit_amd.config import tiny_flux, tiny_sd3

    flux = tiny_flux()
    flux_kw = dict(num_heads=flux.num_heads, depth_multimodal=flux.depth_multimodal, depth_unified=flux.depth_unified,
                   parallel_mlp_for_unified_blocks=True, hidden_size_override=flux.hidden_size, patchify_via_reshape=True,
                   pos_embed_type=rc.PositionalEncoding.PreSDPARope, rope_axes_dim=(16, 56, 56), use_qk_norm=True,
                   pooled_text_embed_dim=flux.pooled_text_embed_dim, token_level_text_embed_dim=flux.token_level_text_embed_dim)
    run_case(mx, rc, rm, flux, flux_kw, 1, (8, 12), 20, [1000.0, 752.0, 500.0], 1, 4101, "flux_b1")
    run_case(mx, rc, rm, flux, flux_kw, 2, (8, 8), 12, [1000.0, 250.0], 1, 4102, "flux_b2")

    sd3 = tiny_sd3(depth=2, heads=2, max_res=16)
    sd3_kw = dict(num_heads=2, depth_multimodal=2, hidden_size_override=128, max_latent_resolution=16,
                  pooled_text_embed_dim=sd3.pooled_text_embed_dim, token_level_text_embed_dim=sd3.token_level_text_embed_dim)
    run_case(mx, rc, rm, sd3, sd3_kw, 2, (8, 12), 20, [1000.0, 857.5], 1, 4103, "sd3_b2")
    run_case(mx, rc, rm, sd3, sd3_kw, 1, (12, 8), 9, [857.5], 0, 4104, "sd3_b1")

    sd35 = replace(tiny_sd3(depth=3, heads=2, max_res=16), use_qk_norm=True)
    sd35_kw = dict(num_heads=2, depth_multimodal=3, hidden_size_override=128, max_latent_resolution=16, use_qk_norm=True,
                   pooled_text_embed_dim=sd35.pooled_text_embed_dim, token_level_text_embed_dim=sd35.token_level_text_embed_dim)
    run_case(mx, rc, rm, sd35, sd35_kw, 2, (8, 8), 10, [500.0], 0, 4105, "sd35_b2")

    # ---- VAE decoder and encoder (vae.py:336-467), the tiny configurations of the test suite ----
    rv = mods["vae"]
    from diffusionkit_amd.config import tiny_vae, tiny_vae_encoder
    from diffusionkit_amd.weights import vae_encoder_weight_shapes, vae_weight_shapes
    dc, ec = tiny_vae(), tiny_vae_encoder()
    g = torch.Generator().manual_seed(4200)
    dec = rv.VAEDecoder(dc.in_channels, dc.out_channels, list(dc.block_out_channels), dc.layers_per_block, dc.resnet_groups)
    run_vae(mx, "vae_decoder", dec, sorted((k, tuple(v)) for k, v in vae_weight_shapes(dc).items()), torch.randn(2, 8, 12, 16, generator=g), 4201)
    enc = rv.VAEEncoder(ec.in_channels, ec.out_channels, list(ec.block_out_channels), ec.layers_per_block, ec.resnet_groups)
    run_vae(mx, "vae_encoder", enc, sorted((k, tuple(v)) for k, v in vae_encoder_weight_shapes(ec).items()),
            torch.rand(2, 64, 96, 3, generator=g) * 2.0 - 1.0, 4202)

    # ---- text encoders (clip.py:62-120, t5.py:316-325): the tiny configurations of the test suite, exact math ----
    from transformers import T5Config
    from diffusionkit_amd import text as tx
    for tag, tcfg in (("clip_quick", tx.tiny_clip("quick_gelu", None)), ("clip_gelu_proj", tx.tiny_clip("gelu", 64))):
        rcfg = rc.CLIPTextModelConfig(num_layers=tcfg.num_layers, model_dims=tcfg.model_dims, num_heads=tcfg.num_heads,
                                      max_length=tcfg.max_length, vocab_size=tcfg.vocab_size, projection_dim=tcfg.projection_dim,
                                      hidden_act=tcfg.hidden_act)
        model = mods["clip"].CLIPTextModel(rcfg)
        spec = sorted((k, tuple(v)) for k, v in tx.synth_clip_weights(tcfg, shapes_only=True).items())
        ckpt = seeded_checkpoint(spec, 4400)
        set_weights(model, ckpt, mx)
        assert len(mlx_standin.tree_flatten(model.parameters())) == len(spec), tag
        gt = torch.Generator().manual_seed(4401)
        tokens = torch.randint(1, tcfg.vocab_size - 1, (2, 16), generator=gt)
        tokens[0, 9], tokens[1, 13] = tcfg.vocab_size - 1, tcfg.vocab_size - 1  # EOS = the largest id (clip.py:94)
        out = model(mx.array(tokens.numpy().astype(np.int32)))
        np.savez_compressed(os.path.join(HERE, f"reference_mlx_{tag}.npz"), spec=json.dumps(spec), seed=4400, checksum=checkpoint_checksum(ckpt),
                            tokens=tokens.numpy(), pooled=np.asarray(out.pooled_output), last=np.asarray(out.last_hidden_state),
                            hidden_m2=np.asarray(out.hidden_states[-2]))
        print(f"{tag}: pooled {np.asarray(out.pooled_output).shape}")
    t5c = tx.tiny_t5()
    hf = T5Config(vocab_size=t5c.vocab_size, d_model=t5c.d_model, d_kv=t5c.d_kv, num_heads=t5c.num_heads, d_ff=t5c.d_ff,
                  num_layers=t5c.num_layers, relative_attention_num_buckets=t5c.relative_attention_num_buckets,
                  relative_attention_max_distance=t5c.relative_attention_max_distance, layer_norm_epsilon=t5c.layer_norm_epsilon,
                  feed_forward_proj="gated-gelu")
    t5 = mods["t5"].SD3T5Encoder(hf, low_memory_mode=False)
    spec = sorted((k, tuple(v)) for k, v in tx.synth_t5_weights(t5c, shapes_only=True).items())
    ckpt = seeded_checkpoint(spec, 4410)
    set_weights(t5, ckpt, mx)
    assert len(mlx_standin.tree_flatten(t5.parameters())) == len(spec), "t5"
    tokens = torch.randint(0, t5c.vocab_size, (2, 24), generator=torch.Generator().manual_seed(4411))
    out = np.asarray(t5(mx.array(tokens.numpy().astype(np.int32))))
    np.savez_compressed(os.path.join(HERE, "reference_mlx_t5.npz"), spec=json.dumps(spec), seed=4410, checksum=checkpoint_checksum(ckpt),
                        tokens=tokens.numpy(), out=out)
    print("t5:", out.shape)

    # ---- CLIP byte-pair tokenizer (tokenizer.py:14-118) on a hand-made vocabulary ----
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789'.,!?-<|>")
    merges = [("t", "h"), ("th", "e</w>"), ("c", "a"), ("ca", "t</w>"), ("a", "n"), ("an", "d</w>"), ("d", "o"), ("do", "g</w>"),
              ("'", "s</w>"), ("i", "n"), ("in", "g</w>"), ("e", "r"), ("o", "n</w>"), ("s", "t"), ("st", "a"), ("!", "!</w>")]
    toks = chars + [c + "</w>" for c in chars] + [a + b for a, b in merges]
    vocab = {t: i for i, t in enumerate(dict.fromkeys(toks))}
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    ranks = {m: i for i, m in enumerate(merges)}
    texts = ["the cat and the dog", "The  CAT's   dog!!", "a photo of 2 cats, sitting on the star-ing dog's nose?", "xyz",
             "doing nothing " * 40, "<|startoftext|>the cat<|endoftext|>", "it's 42."]
    out = {}
    for pad_eos in (False, True):
        tk = mods["tokenizer"].Tokenizer(ranks, vocab, pad_with_eos=pad_eos)
        out[str(pad_eos)] = {"ids": [tk.tokenize(t) for t in texts], "no_specials": tk.tokenize(texts[0], prepend_bos=False, append_eos=False),
                             "batch": tk.tokenize(texts[:2])}
    json.dump({"vocab": vocab, "merges": merges, "texts": texts, "out": out}, open(os.path.join(HERE, "reference_mlx_tokenizer.json"), "w"))
    print("tokenizer:", [len(i) for i in out["False"]["ids"]])

    # ---- checkpoint key maps (model_io.py:130-563): BFL-FLUX / Stability-SD3 / CompVis-VAE layouts through the reference's own
    # *_state_dict_adjustments; the adjusted dict must load STRICTLY into the reference's module tree, and its per-tensor digests
    # are what this repository's loaders (diffusionkit_amd/model_io.py) have to reproduce from the same source checkpoint ----
    from tests.test_model_io import to_bfl_flux, to_compvis_vae, to_compvis_vae_encoder, to_sai_sd3
    from diffusionkit_amd.weights import mmdit_weight_shapes
    rio = importlib.import_module("diffusionkit.mlx.model_io")

    def digest(t):
        v = torch.as_tensor(np.asarray(t), dtype=torch.float64).flatten()
        return [list(np.asarray(t).shape), float((v * (torch.arange(v.numel(), dtype=torch.float64) % 613 + 1)).sum())]

    def named(ours_cfg, seed, shapes):
        return seeded_checkpoint(sorted((k, tuple(v)) for k, v in shapes.items()), seed)

    maps = {}
    w = named(flux, 4600, mmdit_weight_shapes(flux))
    adj = rio.flux_state_dict_adjustments({k: mx.array(v) for k, v in to_bfl_flux(w, flux).items()}, hidden_size=flux.hidden_size,
                                          mlp_ratio=flux.mlp_ratio)
    m = rm.MMDiT(rc.MMDiTConfig(dtype=mx.float32, float16_dtype=mx.float32, low_memory_mode=False, **flux_kw))
    m.update(mlx_standin.tree_unflatten(mlx_standin.tree_flatten(adj)))  # load_flux (model_io.py:784): keys the module lacks are ignored
    have = {k for k, _ in mlx_standin.tree_flatten(m.parameters())}
    assert all(k.endswith("k_proj.bias") for k in set(adj) - have), sorted(set(adj) - have)[:6]  # the k bias of the BFL qkv (quirk Q9)
    assert all(k.endswith("mlp.fc2.bias") for k in have - set(adj)), sorted(have - set(adj))[:6]  # zeroed on every call (Q8)
    maps["flux"] = {"seed": 4600, "tensors": {k: digest(v) for k, v in adj.items() if k in have}}
    w = named(sd3, 4601, mmdit_weight_shapes(sd3))
    # called the way load_mmdit calls it (model_io.py:726-729): the lstrip-by-character-set of that prefix and its "al_layer" repair
    # (:316, :330-332) only work out for keys that carry "model.diffusion_model."
    src = {k: v for k, v in to_sai_sd3(w, sd3, prefix="model.diffusion_model.").items() if not k.startswith("text_encoders.")}
    adj = rio.mmdit_state_dict_adjustments({k: mx.array(v) for k, v in src.items()}, prefix="model.diffusion_model.")
    m = rm.MMDiT(rc.MMDiTConfig(dtype=mx.float32, float16_dtype=mx.float32, low_memory_mode=False, **sd3_kw))
    m.load_weights([(k, v) for k, v in adj.items()], strict=True)
    maps["sd3"] = {"seed": 4601, "tensors": {k: digest(v) for k, v in adj.items()}}
    w = named(dc, 4602, vae_weight_shapes(dc))
    adj = rio.vae_decoder_state_dict_adjustments({k: mx.array(v) for k, v in to_compvis_vae(w, dc, prefix="decoder.").items()})
    rv.VAEDecoder(dc.in_channels, dc.out_channels, list(dc.block_out_channels), dc.layers_per_block, dc.resnet_groups).load_weights(
        [(k, v) for k, v in adj.items()], strict=True)
    maps["vae_decoder"] = {"seed": 4602, "tensors": {k: digest(v) for k, v in adj.items()}}
    w = named(ec, 4603, vae_encoder_weight_shapes(ec))
    adj = rio.vae_encoder_state_dict_adjustments({k: mx.array(v) for k, v in to_compvis_vae_encoder(w, "encoder.").items()})
    rv.VAEEncoder(ec.in_channels, ec.out_channels, list(ec.block_out_channels), ec.layers_per_block, ec.resnet_groups).load_weights(
        [(k, v) for k, v in adj.items()], strict=True)
    maps["vae_encoder"] = {"seed": 4603, "tensors": {k: digest(v) for k, v in adj.items()}}
    json.dump(maps, open(os.path.join(HERE, "reference_mlx_keymaps.json"), "w"))
    print("keymaps:", {k: len(v["tensors"]) for k, v in maps.items()})

    # ---- encode_text (mlx/__init__.py:176-251 SD3, :642-671 FLUX) on pipelines assembled from the pieces above ----
    class WordT5Tokenizer:  # stands in for the sentencepiece T5 tokenizer (a download): one id per word, EOS = 1, like T5
        pad_with_eos, pad_to_max_length = False, True

        def __init__(self, max_length):
            self.max_length = max_length

        def tokenize(self, s):
            return [2 + (sum(map(ord, w)) % 300) for w in s.split()][: self.max_length - 1] + [1]

    def clip_model(tcfg, seed):
        rcfg = rc.CLIPTextModelConfig(num_layers=tcfg.num_layers, model_dims=tcfg.model_dims, num_heads=tcfg.num_heads,
                                      max_length=tcfg.max_length, vocab_size=tcfg.vocab_size, projection_dim=tcfg.projection_dim,
                                      hidden_act=tcfg.hidden_act)
        m = mods["clip"].CLIPTextModel(rcfg)
        set_weights(m, seeded_checkpoint(sorted((k, tuple(v)) for k, v in tx.synth_clip_weights(tcfg, shapes_only=True).items()), seed), mx)
        return m

    from dataclasses import replace as dc_replace
    cl, cg = dc_replace(tx.tiny_clip("quick_gelu", None), vocab_size=len(vocab)), dc_replace(tx.tiny_clip("gelu", 64), vocab_size=len(vocab))
    enc_out = {}
    pipe = object.__new__(ref.DiffusionPipeline)
    pipe.tokenizer_l = mods["tokenizer"].Tokenizer(ranks, vocab, pad_with_eos=True)
    pipe.tokenizer_g = mods["tokenizer"].Tokenizer(ranks, vocab, pad_with_eos=False)
    pipe.clip_l, pipe.clip_g, pipe.use_t5 = clip_model(cl, 4500), clip_model(cg, 4501), False
    for name, cfgw, neg in (("sd3_cfg5", 5.0, "the dog"), ("sd3_cfg1", 1.0, "the dog")):
        c, p_ = pipe.encode_text("the cat and the dog's star", cfgw, neg)
        enc_out[name + "_cond"], enc_out[name + "_pooled"] = np.asarray(c), np.asarray(p_)
    fp = object.__new__(ref.FluxPipeline)
    fp.model_version = "argmaxinc/mlx-FLUX.1-schnell"
    fp.tokenizer_l, fp.clip_l = pipe.tokenizer_l, pipe.clip_l
    fp.t5_tokenizer, fp.t5_encoder = WordT5Tokenizer(ref.T5_MAX_LENGTH[fp.model_version]), t5
    c, p_ = fp.encode_text("the cat and the dog's star", 0.0, "ignored")
    enc_out["flux_cond"], enc_out["flux_pooled"] = np.asarray(c), np.asarray(p_)
    np.savez_compressed(os.path.join(HERE, "reference_mlx_encode_text.npz"), **enc_out)
    print("encode_text:", {k: v.shape for k, v in enc_out.items()})

    # ---- generate_image itself (mlx/__init__.py:294-534): prompt -> tokens -> CLIP-L / CLIP-G -> conditioning -> step loop with
    # CFG -> VAE decoder -> clip(x / 2 + 0.5) -> uint8, on an SD3 pipeline assembled from the reference's own parts ----
    e2e = dc_replace(tiny_sd3(depth=2, heads=2, max_res=16), token_level_text_embed_dim=4096, pooled_text_embed_dim=128 + 64)
    e2e_kw = dict(num_heads=2, depth_multimodal=2, hidden_size_override=128, max_latent_resolution=16, pooled_text_embed_dim=192,
                  token_level_text_embed_dim=4096)
    mm_e2e = rm.MMDiT(rc.MMDiTConfig(dtype=mx.float32, float16_dtype=mx.float32, low_memory_mode=False, **e2e_kw))
    spec_e2e = sorted((k, tuple(v)) for k, v in mmdit_weight_shapes(e2e).items())
    set_weights(mm_e2e, seeded_checkpoint(spec_e2e, 4700), mx)
    gp = object.__new__(ref.DiffusionPipeline)
    gp.mmdit, gp.decoder, gp.encoder = mm_e2e, dec, enc
    gp.clip_l, gp.clip_g, gp.tokenizer_l, gp.tokenizer_g = pipe.clip_l, pipe.clip_g, pipe.tokenizer_l, pipe.tokenizer_g
    gp.use_t5, gp.t5, gp.use_clip_g, gp.low_memory_mode = False, None, True, False
    gp.activation_dtype = mx.float32
    gp.sampler, gp.latent_format = rs.ModelSamplingDiscreteFlow(shift=3.0), ref.SD3LatentFormat()
    saved_e2e = adaln_items(mm_e2e, mx)
    gp.load_mmdit = lambda only_modulation_dict=False: saved_e2e
    img, log = gp.generate_image("the cat and the dog's star", num_steps=3, cfg_weight=5.0, negative_text="the dog", latent_size=(8, 12),
                                 seed=11, verbose=False)
    np.savez_compressed(os.path.join(HERE, "reference_mlx_generate_image.npz"), image=np.asarray(img), seed_mmdit=4700,
                        n_iter=len(log["denoising"]["iter_time"]))
    print("generate_image:", np.asarray(img).shape, np.asarray(img).dtype, "mean", float(np.asarray(img).mean()))

    # ---- the step loop: denoise_latents end to end (SD3 with CFG, FLUX without, SD3 img2img through the encoder above) ----
    run_denoise(mx, ref, mods, "sd3_cfg", sd3, sd3_kw, False, 5.0, 3, (8, 12), 20, 4301)
    run_denoise(mx, ref, mods, "flux", flux, flux_kw, True, 0.0, 4, (8, 8), 12, 4302)
    run_denoise(mx, ref, mods, "sd3_img2img", sd3, sd3_kw, False, 5.0, 5, (8, 8), 20, 4303, denoise=0.6, encoder=enc)

    # ---- read_image (mlx/__init__.py:536-551): sizes cut to multiples of 64 through a LANCZOS resize, RGBA -> RGB, [-1, 1] ----
    from PIL import Image as PILImage
    rgba = (torch.rand(100, 150, 4, generator=torch.Generator().manual_seed(4900)) * 255).to(torch.uint8).numpy()
    tmp = os.path.join(HERE, "_tmp_read_image.png")
    PILImage.fromarray(rgba).save(tmp)
    ri = np.asarray(ref.DiffusionPipeline.read_image(object.__new__(ref.DiffusionPipeline), tmp))
    os.remove(tmp)
    np.savez_compressed(os.path.join(HERE, "reference_mlx_read_image.npz"), rgba=rgba, out=ri)
    print("read_image:", ri.shape)

    # ---- the reference's quality metrics (python/src/diffusionkit/utils.py:52-82): compute_psnr on float arrays, image_psnr on
    # PIL images (whose uint8 difference wraps modulo 256 before it is squared) ----
    from PIL import Image
    ru = importlib.import_module("diffusionkit.utils")
    gq = torch.Generator().manual_seed(4800)
    a = torch.rand(32, 48, 3, generator=gq)
    b = (a + 0.03 * torch.randn(32, 48, 3, generator=gq)).clamp(0, 1)
    a8, b8 = (a * 255).to(torch.uint8).numpy(), (b * 255).to(torch.uint8).numpy()
    np.savez_compressed(os.path.join(HERE, "reference_psnr.npz"), a=a.numpy(), b=b.numpy(), a8=a8, b8=b8,
                        compute_psnr=float(ru.compute_psnr(a.numpy(), b.numpy())),
                        image_psnr=float(ru.image_psnr(Image.fromarray(a8), Image.fromarray(b8))))
    print("psnr:", float(ru.compute_psnr(a.numpy(), b.numpy())), float(ru.image_psnr(Image.fromarray(a8), Image.fromarray(b8))))

    # ---- samplers: the schedules the step loop indexes (sampler.py:10-77) ----
    out = {}
    for name, cls in (("flow", rs.ModelSamplingDiscreteFlow), ("flux", rs.FluxSampler)):
        for shift in (1.0, 3.0):
            s = cls(shift=shift)
            out[f"{name}_shift{shift}_sigmas"] = np.asarray(s.sigmas, dtype=np.float64)
            out[f"{name}_shift{shift}_timestep_of_half"] = np.asarray(s.timestep(mx.array(0.5)), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "reference_mlx_sampler.npz"), **out)
    print("sampler:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
