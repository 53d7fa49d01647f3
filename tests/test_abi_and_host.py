"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol the header
declares, the ctypes table covers the header, weight packing matches the engine's table layout,
and the host classes refuse to run without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from diffusionkit_amd import _lib
from diffusionkit_amd.config import FLUX_SCHNELL, SD3_2b, tiny_flux, tiny_sd3, tiny_vae
from diffusionkit_amd.weights import (adaln_order, blob_pack, blob_unpack, pack_mmdit, pack_vae, synth_mmdit_weights,
                                      synth_vae_weights)


def header_symbols():
    src = open(_lib.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dk_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.dk_abi_version() == 5


def test_mod_table_layout_matches_config():
    lib = _lib.load()
    from diffusionkit_amd.engine import MMDiTEngine  # noqa: F401  (import only)
    for cfg in (tiny_flux(), tiny_sd3(), FLUX_SCHNELL, SD3_2b):
        c = _lib.dk_mmdit_config()
        c.num_heads, c.depth_multimodal, c.depth_unified = cfg.num_heads, cfg.depth_multimodal, cfg.depth_unified
        c.hidden_size, c.mlp_ratio, c.vae_latent_dim, c.patch_size = cfg.hidden_size, 4, 16, 2
        c.pooled_text_embed_dim, c.token_level_text_embed_dim, c.frequency_embed_dim = 64, 64, 256
        h = ctypes.c_void_p()
        assert lib.dk_mmdit_create(ctypes.byref(c), ctypes.byref(h)) == 0
        assert lib.dk_mmdit_mod_rows(h) == cfg.num_modulation_rows()
        # offsets follow adaln_order: img(6), txt(6|2) per double block, singles(3), final(2)
        off = 0
        for i in range(cfg.depth_multimodal):
            assert lib.dk_mmdit_mod_offset(h, 0, i) == off
            off += 6
            assert lib.dk_mmdit_mod_offset(h, 1, i) == off
            off += 2 if (i == cfg.depth_multimodal - 1 and cfg.depth_unified < 1) else 6
        for i in range(cfg.depth_unified):
            assert lib.dk_mmdit_mod_offset(h, 2, i) == off
            off += 3
        assert lib.dk_mmdit_mod_offset(h, 3, 0) == off
        lib.dk_mmdit_destroy(h)
    if FLUX_SCHNELL.num_modulation_rows() != 344:
        raise AssertionError("FLUX modulation rows must be 19*12 + 38*3 + 2 = 344 (SURVEY.md §8a a6)")


def test_create_rejects_bad_config():
    lib = _lib.load()
    c = _lib.dk_mmdit_config()
    c.num_heads, c.hidden_size, c.patch_size, c.vae_latent_dim = 3, 96, 2, 16
    h = ctypes.c_void_p()
    assert lib.dk_mmdit_create(ctypes.byref(c), ctypes.byref(h)) != 0
    assert b"head_dim" in lib.dk_last_error()


def test_pack_mmdit_shapes():
    for cfg in (tiny_flux(), tiny_sd3()):
        w = synth_mmdit_weights(cfg)
        n_src = len(w)
        p = pack_mmdit(cfg, w, "cpu")
        assert len(w) == n_src  # not consumed by default
        h = cfg.hidden_size
        assert p["adaLN.weight"].shape == (cfg.num_modulation_rows() * h, h)
        assert p["x_embedder.proj.weight"].shape == (h, 64)
        b0 = "multimodal_transformer_blocks.0.image_transformer_block"
        assert p[b0 + ".attn.qkv.weight"].shape == (3 * h, h)
        assert torch.all(p[b0 + ".attn.qkv.bias"][h:2 * h] == 0)  # k_proj has no bias (quirk Q9)
        assert torch.equal(p[b0 + ".attn.qkv.weight"][h:2 * h], w[b0 + ".attn.k_proj.weight"])
        if cfg.depth_unified:
            s0 = "unified_transformer_blocks.0.transformer_block"
            assert p[s0 + ".linear2.weight"].shape == (h, 5 * h)  # below the padded-pitch threshold (dk_weight_pitch)
            assert torch.equal(p[s0 + ".linear2.weight"][:, :h], w[s0 + ".attn.o_proj.weight"])
            assert torch.equal(p[s0 + ".linear2.bias"], w[s0 + ".attn.o_proj.bias"])  # one bias (quirk Q8)
            # fused linear1 = [q | k | v | fc1] over one read of the modulated activations
            assert p[s0 + ".linear1.weight"].shape == ((3 + cfg.mlp_ratio) * h, h)
            assert torch.equal(p[s0 + ".linear1.weight"][2 * h:3 * h], w[s0 + ".attn.v_proj.weight"])
            assert torch.equal(p[s0 + ".linear1.weight"][3 * h:], w[s0 + ".mlp.fc1.weight"])
            assert torch.all(p[s0 + ".linear1.bias"][h:2 * h] == 0)
            assert torch.equal(p[s0 + ".linear1.bias"][3 * h:], w[s0 + ".mlp.fc1.bias"])
        first = adaln_order(cfg)[0]
        assert torch.equal(p["adaLN.weight"][:6 * h], w[first + ".adaLN_modulation.layers.1.weight"])
        blob, index = blob_pack(p)
        q = blob_unpack(blob, index)
        assert all(torch.equal(q[k], p[k]) for k in p)


def test_pack_vae_pads_conv_in():
    vc = tiny_vae()
    p = pack_vae(vc, synth_vae_weights(vc), "cpu")
    assert p["conv_in.weight"].shape == (128, 9 * 64)
    w4 = p["conv_in.weight"].reshape(128, 3, 3, 64)
    assert torch.all(w4[..., 16:] == 0)
    assert p["up_blocks.1.upsample.weight"].shape == (64, 9 * 64)
    assert "up_blocks.0.upsample.weight" not in p


def test_no_cpu_fallback():
    from diffusionkit_amd.engine import MMDiTEngine
    cfg = tiny_flux()
    packed = pack_mmdit(cfg, synth_mmdit_weights(cfg), "cpu")
    with pytest.raises(_lib.DkHipError):
        MMDiTEngine(cfg, packed)  # CPU tensors are rejected: the product path is HIP only


def test_unknown_model_version_is_a_keyerror():
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    from diffusionkit_amd import pipeline
    with pytest.raises(KeyError):
        pipeline.DiffusionPipeline(model_version="not-a-model")


def test_package_does_not_import_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusionkit_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_weight_pitch_rule_and_pitched_packing():
    """dk_weight_pitch (include/dk_hip.h): dense below 8192 elements per row, +64 from there on; pack_mmdit lays the
    fc2 / linear2 weights out at that pitch with zero pad columns (no compute: runs without a GPU)."""
    from diffusionkit_amd import _lib
    lib = _lib.load()
    assert [lib.dk_weight_pitch(k) for k in (64, 6144, 8191, 8192, 12288, 15360)] == [64, 6144, 8191, 8256, 12352, 15424]
    cfg = tiny_flux()
    w = synth_mmdit_weights(cfg)
    h, r = cfg.hidden_size, cfg.mlp_ratio
    assert lib.dk_tune_set(b"pitch_min_k", 64) == 0
    try:
        p = pack_mmdit(cfg, w, "cpu")
    finally:
        assert lib.dk_tune_set(b"pitch_min_k", 8192) == 0
    b0 = "multimodal_transformer_blocks.0.image_transformer_block.mlp.fc2.weight"
    s0 = "unified_transformer_blocks.0.transformer_block"
    assert p[b0].shape == (h, r * h + 64) and p[s0 + ".linear2.weight"].shape == (h, (1 + r) * h + 64)
    assert torch.equal(p[b0][:, :r * h], w[b0]) and torch.all(p[b0][:, r * h:] == 0)
    assert torch.equal(p[s0 + ".linear2.weight"][:, h:(1 + r) * h], w[s0 + ".mlp.fc2.weight"])
    assert torch.all(p[s0 + ".linear2.weight"][:, (1 + r) * h:] == 0)


def test_tune_keys_and_workspace_sizes():
    """dk_tune_set knows every documented key (include/dk_hip.h) and rejects others with an error message; the
    workspace-size queries answer without a GPU."""
    from diffusionkit_amd import _lib
    lib = _lib.load()
    for key in (b"gemm", b"gemm_v4", b"gemm_skew", b"gemm_mf", b"gemm_split", b"gemm_fuse_k", b"gemm_fuse_q", b"attn", b"attn_fuse_q", b"attn_split", b"conv_halo",
                b"conv_v4"):
        assert lib.dk_tune_set(key, -1 if key not in (b"gemm_fuse_k", b"attn_fuse_q", b"conv_v4") else 1) == 0, key
    assert lib.dk_tune_set(b"gemm_sched", 0) == -1  # a knob of the removed kernel generations
    assert b"unknown tuning key" in lib.dk_last_error()
    assert lib.dk_gemm_workspace_bytes() == 256 * 256 * 256 * 4 + 4096
    assert lib.dk_tune_set(b"attn_balance", 1) == -1 and lib.dk_tune_set(b"vae_attn", 0) == -1  # round 5: moved to profiles/lab_kernels
    # the partial results of attention5.hip's key-split workgroups: <= 255 blocks x 4 key ranges x (256 rows x 256 B + 256 x (offset, l))
    assert lib.dk_attention_workspace_bytes() == 1020 * (65536 + 2048)
    assert lib.dk_attention_set_workspace(None, 0) == 0


def test_bench_cpu_sample_keeps_the_workload_width():
    """bench.py's bounded CPU sample (`cpu_sample_config`) is the workload's model with 1/f of its blocks: same width and head size
    (SD3 derives its width from the depth, 64 x depth_multimodal -- the sample must not shrink it), block ratio kept; and
    `cpu_baseline` runs end to end on a small SD3-style workload (two conditioning rows)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from diffusionkit_amd.config import FLUX_DEV, SD3_8b
    for cfg, blocks, f in ((FLUX_SCHNELL, (1, 2), 19), (FLUX_DEV, (1, 2), 19), (SD3_2b, (2, 0), 12), (SD3_8b, (2, 0), 19)):
        scfg, ff = bench.cpu_sample_config(cfg)
        assert (scfg.depth_multimodal, scfg.depth_unified, ff) == (*blocks, f)
        assert scfg.hidden_size == cfg.hidden_size and scfg.head_dim == cfg.head_dim and scfg.num_heads == cfg.num_heads
    cfg = tiny_sd3(depth=4)
    out = bench.cpu_baseline({"cfg": cfg, "latent": (8, 8), "num_steps": 2, "S_t": 24, "rows": 2}, threads=2)
    assert out["value"] > 0 and out["kind"] == "port" and out["cores"] == 2
