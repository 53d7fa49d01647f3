"""Full-size / full-depth parity on the BASELINE.json configurations (VERDICT r1 "Next round" item 1): the HIP path against
fp32-oracle outputs computed once on the host and committed (tests/golden/fullsize_*.npz, generator
tests/golden/make_fullsize_fixtures.py -- same seeded weights and inputs, regenerated here).

Tolerances are ABSOLUTE, one per case, stated in DESIGN.md section 4 and here.  They are floors on PSNR (reference metric,
python/src/diffusionkit/utils.py:70-82) / ceilings on relative L2 of the HIP result against the fp32 oracle; what the
bf16-emulating oracle (a model of the reference's own MLX rounding points) reaches on the same case is stored in the fixture and
printed next to the measurement, as context only.
"""
import os
import sys

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import make_fullsize_fixtures as fx  # noqa: E402  (case definitions + seeded inputs shared with the generator)

from diffusionkit_amd.config import VAEDecoderConfig  # noqa: E402
from diffusionkit_amd.weights import pack_mmdit, pack_vae, synth_mmdit_weights, synth_vae_weights  # noqa: E402
from tests._util import BF, psnr, rel_l2  # noqa: E402

pytestmark = pytest.mark.gpu

# ---- stated tolerances ------------------------------------------------------------------------------
# Round 4: every gate sits at the value measured on MI355X (profiles/r03_fullsize_parity.log) minus 2 dB of
# PSNR / times 1.5 in relative L2 -- a regression of one bf16 rounding point per stored tensor fails.  The fp8 rows are gated at
# the same distance from THEIR measurement (the un-quantised oracle is 33-37 dB away by construction, section 4).
TOL = {
    # case: (min PSNR dB, max rel-L2) of the HIP result against the fp32 oracle
    "sd3_512_latent": (49.8, 1.65e-2),   # BASELINE configs[0]: 24 blocks x 4 steps, final latent
    "sd3_512_image": (47.5, None),      # ... decoded 512 x 512 image in [0, 1] (the reference's torch<->CoreML bar)
    "vae_1024_image": (47.6, None),     # full-size decode, image in [0, 1]
    "vae_1024_raw": (None, 2.6e-2),     # ... decoder output before the clip
    "sd3_1024_final": (50.5, 1.75e-2),   # SD3 bench shape, depth 2, model output
    "flux_1024_final": (48.3, 1.7e-2),  # FLUX depth 4 + 8 at S = 4352 (round 3: oracle with the reference's bf16 timestep embedding; emu 50.4 dB / 1.15e-2)
    "flux_1024_fp8_final": (31.6, 1.0e-1),  # ... with e4m3 weights / MX-fp8 activations against the fp32 oracle with the ORIGINAL weights
    # ---- round 3 ----
    "flux_dev_512_final": (52.3, 1.28e-2),      # configs[3]'s shape (S_t 512, S 4608), 1 + 1 blocks, bf16 weights (emu 54.3 dB / 8.7e-3)
    "flux_dev_512_fp8_final": (34.9, 9.0e-2),  # ... e4m3 weights / MX-fp8 activations against the fp32 oracle with the original weights
    "sd3_full_1024_x3": (71.2, 1.6e-3),         # configs[2] at full depth: 24 blocks, B 2, CFG 5, first 3 of 50 Euler steps
    "flux_full_latent": (52.4, 1.28e-2),      # BASELINE configs[1] end to end (57 blocks x 4 steps), round-3 fixture (the reference's bf16 timestep embedding in the oracle)
    "flux_full_fp8_latent": (35.3, 9.0e-2),  # the same image with e4m3 weights / MX-fp8 activations on every block Linear
    # ---- round 6 ----
    "flux_512_fp8_latent": (35.3, 9.0e-2),          # fp8 weights at 512 x 512 (the fp8 GEMM's K split inside the model): measured 37.34 dB / 6.03e-2 (1024 x 1024: 37.37 dB)
    "flux_512_fp8_policy_latent": (39.2, 5.8e-2),   # ... with the shipped precision policy: measured 41.26 dB / 3.84e-2
    "flux_768_latent": (52.0, 1.29e-2),             # 768 x 768 (two K ranges per tile, attention5 in one round): measured 54.02 dB / 8.60e-3
    "flux_512_latent": (52.0, 1.32e-2),      # FLUX.1-schnell end to end at the reference CLI's 512 x 512 default (the K-split launches inside the model): measured 54.06 dB / 8.79e-3 (emu 54.09 dB)
}


def load(name):
    path = os.path.join(GOLD, f"fullsize_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    return np.load(path)


def check(name, ref, got, f=None, emu_keys=("emu_psnr", "emu_rel_l2")):
    p, e = psnr(ref, got), rel_l2(ref, got)
    ctx = ""
    if f is not None:
        ctx = ", ".join(f"{k} {float(f[k]):.4g}" for k in emu_keys if k in f.files)
    print(f"[fullsize] {name}: PSNR {p:.2f} dB, rel-L2 {e:.4e}, max-abs {float((ref.double() - got.double().cpu()).abs().max()):.4g}   (bf16-emulating oracle{'; emu_fp16_*: the fp16-emulating one, SD3 reference dtype' if 'fp16' in ctx else ''}: {ctx})")
    min_p, max_e = TOL[name]
    if min_p is not None:
        assert p >= min_p, f"{name}: PSNR {p:.2f} dB < {min_p}"
    if max_e is not None:
        assert e <= max_e, f"{name}: rel-L2 {e:.3e} > {max_e}"


# ---- seeded weight sets, drawn ahead of the tests that use them -----------------------------------------------------------------
# The weights of a case come out of ONE sequential CPU generator stream (the fixture generator drew them the same way), 40-300 M
# values per second: 60 s for the 12 B parameters of FLUX.1-schnell, 10 s for SD3-medium.  Background threads draw the sets while
# earlier tests run on the GPU (torch.randn releases the GIL); a test takes its set from there, or draws it itself when no thread
# is running (a selection with -k).
import threading  # noqa: E402

_SYNTH_PLAN = []   # (key, cfg, seed, uses) in file order
_SYNTH = {}        # key -> [weights, uses left]
_SYNTH_COND = threading.Condition()
_SYNTH_THREAD = []


def _synth_key(cfg, seed):
    from dataclasses import replace
    return (repr(replace(cfg, weight_dtype="bf16")) if hasattr(cfg, "weight_dtype") else repr(cfg), int(seed))


def _synth_worker(plan):
    for key, cfg, seed, uses in plan:
        w = synth_mmdit_weights(cfg, seed=seed)
        with _SYNTH_COND:
            _SYNTH[key] = [w, uses]
            _SYNTH_COND.notify_all()


def start_synth_prefetch():
    """called by tests/conftest.py at the START of a GPU session that runs this whole file: two threads, the 12 B parameters of
    FLUX.1-schnell in one, the smaller sets in file order in the other -- they are ready when the earlier test files are through"""
    if _SYNTH_THREAD:
        return
    small = []
    for c, uses in ((fx.SD3_512, 1), (fx.SD3_1024, 1), (fx.FLUX_1024, 2), (fx.FLUX_DEV_512, 2), (fx.SD3_FULL_1024, 3)):
        small.append((_synth_key(c["cfg"], c["seed_w"]), c["cfg"], c["seed_w"], uses))
    if os.path.exists(os.path.join(GOLD, "fullsize_sd35_full.npz")):  # (8 B parameters, ~40 s: behind the smaller sets)
        small.append((_synth_key(fx.SD35_FULL["cfg"], fx.SD35_FULL["seed_w"]), fx.SD35_FULL["cfg"], fx.SD35_FULL["seed_w"], 1))
    big = [(_synth_key(fx.FLUX_FULL["cfg"], fx.FLUX_FULL["seed_w"]), fx.FLUX_FULL["cfg"], fx.FLUX_FULL["seed_w"], 1)]
    _SYNTH_PLAN.extend(big + small)
    for plan in (big, small):
        th = threading.Thread(target=_synth_worker, args=(plan,), daemon=True)
        _SYNTH_THREAD.append(th)
        th.start()


def synth_cached(cfg, seed, keep=False):
    """the weight set of (cfg, seed): a private (shallow) copy of the dict, so that pack_mmdit(consume=True) can empty it"""
    key = _synth_key(cfg, seed)
    if _SYNTH_THREAD and any(k == key for k, *_ in _SYNTH_PLAN):
        with _SYNTH_COND:
            while key not in _SYNTH:
                _SYNTH_COND.wait(timeout=600)
            ent = _SYNTH[key]
            ent[1] -= 1
            w = ent[0]
            if ent[1] <= 0 and not keep:
                del _SYNTH[key]
        return dict(w)
    return synth_mmdit_weights(cfg, seed=seed)


_FLUX_SYNTH = {}


def flux_full_synth():
    """the seeded 57-block FLUX.1-schnell weight set (bf16 on the host, 24 GB), drawn once per test session"""
    if "w" not in _FLUX_SYNTH:
        _FLUX_SYNTH["w"] = synth_cached(fx.FLUX_FULL["cfg"], fx.FLUX_FULL["seed_w"], keep=True)
    return _FLUX_SYNTH["w"]


_FLUX_PIPES = {}


def flux_full_pipe(dev, fp8=False):
    """the full-depth FLUX.1-schnell pipeline on the seeded weight set, packed once per session and weight dtype (24 GB bf16 /
    12 GB e4m3): shared by the end-to-end cases, the teacher-forced blocks and the teacher-forced Euler steps"""
    if fp8 not in _FLUX_PIPES:
        from dataclasses import replace
        from diffusionkit_amd.config import fp8_config
        from diffusionkit_amd.pipeline import FluxPipeline
        c = fx.FLUX_FULL
        # fp8: True = every block Linear in fp8; "quality" = the shipped precision policy (config.fp8_config: first 12 double blocks bf16)
        cfg = fp8_config(c["cfg"], "quality") if fp8 == "quality" else replace(c["cfg"], weight_dtype="fp8_e4m3") if fp8 else c["cfg"]
        packed = {"mmdit": pack_mmdit(cfg, flux_full_synth(), dev)}
        _FLUX_PIPES[fp8] = FluxPipeline(w16=True, a16=True, shift=c["shift"], device=dev, text_len=c["S_t"], packed_weights=packed, mmdit_config=cfg)
    return _FLUX_PIPES[fp8]


def forced_steps(pipe, c, dev):
    """Euler directions d_i of the teacher-forced steps of ``c`` (tests/golden/make_fullsize_fixtures.py: forced_inputs) through the
    pipeline's own sample_euler / CFGDenoiser, one two-entry schedule per step"""
    from diffusionkit_amd.pipeline import CFGDenoiser, sample_euler
    text, pooled, steps = fx.forced_inputs(c)
    extra = {"conditioning": text.to(dev, BF), "cfg_weight": c["cfg_weight"], "pooled_conditioning": pooled.to(dev, BF)}
    sig_pipe = np.asarray(pipe.get_sigmas(pipe.sampler, c["steps_of"]), dtype=np.float64)
    out = {}
    for i, x_i, sig2 in steps:
        assert np.allclose(sig_pipe[i: i + 2], sig2.numpy().astype(np.float64), rtol=0, atol=1e-7)  # the pipeline's own schedule entries
        x_next, it = sample_euler(CFGDenoiser(pipe), x_i.to(dev), sig2.numpy(), extra_args=dict(extra))
        assert len(it) == 1
        out[i] = fx.euler_direction(x_i, x_next.float().cpu(), sig2)
    return out


def test_sd3_medium_512_full_depth_pipeline(dev):
    """BASELINE configs[0]: SD3-medium, all 24 blocks, latent 64 x 64, 4 Euler steps, CFG off, + VAE decode -- through the
    public pipeline API (denoise_latents + decoder), vs the fp32 oracle's latent and image."""
    from diffusionkit_amd.pipeline import DiffusionPipeline
    f = load("sd3_512")
    c = fx.SD3_512
    packed = {"mmdit": pack_mmdit(c["cfg"], synth_cached(c["cfg"], c["seed_w"]), dev, consume=True),
              "vae_decoder": pack_vae(VAEDecoderConfig(), synth_vae_weights(VAEDecoderConfig(), seed=c["seed_vae"]), dev)}
    pipe = DiffusionPipeline(w16=True, a16=True, shift=c["shift"], device=dev, text_len=c["S_t"], packed_weights=packed)
    text, pooled = fx.sd3_512_inputs()
    lat, iter_time = pipe.denoise_latents(text.to(dev, BF), pooled.to(dev, BF), num_steps=c["steps"], cfg_weight=0.0, latent_size=c["latent"],
                                          seed=c["noise_seed"])
    assert len(iter_time) == c["steps"] and lat.shape == (1, 64, 64, 16)
    check("sd3_512_latent", torch.from_numpy(f["latent_fp32"]), lat.cpu(), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs", "emu_fp16_psnr", "emu_fp16_rel_l2"))
    # decode the ORACLE's latent (the image fixture was decoded from it), so that the image check isolates the VAE at 512 x 512
    img, u8, _ = pipe.decoder.decode(torch.from_numpy(f["latent_fp32"]).to(dev))
    check("sd3_512_image", torch.from_numpy(f["image_fp32_f16"].astype(np.float32)), img.cpu())
    # and the pipeline's own latent end to end: still an image of the same scene
    img2, _, _ = pipe.decoder.decode(lat)
    assert psnr(torch.from_numpy(f["image_fp32_f16"].astype(np.float32)), img2.cpu()) > 30.0


def test_vae_decode_1024_vs_oracle(dev):
    """one full-size decode (latent 128 x 128 -> 1024 x 1024): the T = 16384 mid-block attention and every conv at its bench size"""
    from diffusionkit_amd.engine import VAEDecoderEngine
    f = load("vae_1024")
    c = fx.VAE_1024
    eng = VAEDecoderEngine(c["cfg"], pack_vae(c["cfg"], synth_vae_weights(c["cfg"], seed=c["seed_vae"]), dev))
    z = fx.randn(1, c["latent"][0], c["latent"][1], 16, seed=c["z_seed"])
    img, u8, raw = eng.decode(z.to(dev), want_raw=True)
    ref_raw = torch.from_numpy(f["raw_fp32_f16"].astype(np.float32))
    check("vae_1024_raw", ref_raw, raw[..., :3].float().cpu(), f, ("emu_rel_l2",))
    check("vae_1024_image", torch.clip(ref_raw / 2 + 0.5, 0, 1), img.cpu(), f, ("emu_psnr_image",))


def _forward(c, dev):
    from diffusionkit_amd.engine import MMDiTEngine
    cfg = c["cfg"]
    eng = MMDiTEngine(cfg, pack_mmdit(cfg, synth_cached(cfg, c["seed_w"]), dev, consume=True))
    text, pooled, lat = fx.forward_inputs(c)
    eng.prepare(c["B"], c["latent"], c["S_t"], len(c["timesteps"]))
    eng.cache_modulation_params(pooled.to(dev), c["timesteps"])
    return eng.forward_tokens(eng.patchify(lat.to(dev)), text.to(dev, BF), c["step"]).float().cpu()


def test_sd3_1024_bench_shape_vs_oracle(dev):
    """SD3-medium at the shape bench.py --workload sd3-medium-1024 runs: CFG pair (B = 2), 589 ragged text tokens in two row
    segments, latent 128 x 128, depth 2"""
    f = load("sd3_1024")
    check("sd3_1024_final", torch.from_numpy(f["final_fp32"]), _forward(fx.SD3_1024, dev), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def test_flux_1024_depth_4_8_vs_oracle(dev):
    """FLUX.1-schnell geometry at S = 256 + 4096, 4 double + 8 single blocks"""
    f = load("flux_1024")
    check("flux_1024_final", torch.from_numpy(f["final_fp32"]), _forward(fx.FLUX_1024, dev), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def test_flux_1024_depth_4_8_fp8_weights_vs_oracle(dev):
    """The same 4 + 8 blocks with e4m3 weights and MX-fp8 activations on the block Linears (BASELINE configs[3]'s arithmetic) against
    the fp32 oracle with the ORIGINAL weights: the distance is the quantisation noise of the format plus the bf16 path's."""
    from dataclasses import replace
    f = load("flux_1024")
    c = dict(fx.FLUX_1024)
    c["cfg"] = replace(c["cfg"], weight_dtype="fp8_e4m3")
    check("flux_1024_fp8_final", torch.from_numpy(f["final_fp32"]), _forward(c, dev), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def test_flux_schnell_1024_full_depth_pipeline(dev):
    """BASELINE configs[1] end to end: FLUX.1-schnell, 19 + 38 blocks, latent 128 x 128, 4 Euler steps, vs the fp32 oracle's latent"""
    f = load("flux_full")
    c = fx.FLUX_FULL
    pipe = flux_full_pipe(dev)
    text, pooled = fx.flux_full_inputs()
    lat, _ = pipe.denoise_latents(text.to(dev, BF), pooled.to(dev, BF), num_steps=c["steps"], cfg_weight=0.0, latent_size=c["latent"],
                                  seed=c["noise_seed"])
    check("flux_full_latent", torch.from_numpy(f["latent_fp32"]), lat.cpu(), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def test_flux_schnell_512_full_depth_pipeline(dev):
    """Round 6: FLUX.1-schnell end to end at the resolution the reference's CLI defaults to (mlx/scripts/generate_images.py:15-30: 512 x 512 = latent
    64 x 64, S = 1280), 19 + 38 blocks, 4 Euler steps -- the configuration whose fc2 / linear2 launches (60 tiles) the round's small-launch rule cuts along K
    and hands to the 8-wave kernel -- against the fp32 oracle's latent"""
    f = load("flux_512")
    c = fx.FLUX_512
    pipe = flux_full_pipe(dev)
    text, pooled = fx.flux_full_inputs()
    lat, _ = pipe.denoise_latents(text.to(dev, BF), pooled.to(dev, BF), num_steps=c["steps"], cfg_weight=0.0, latent_size=c["latent"],
                                  seed=c["noise_seed"])
    assert lat.shape == (1, 64, 64, 16)
    check("flux_512_latent", torch.from_numpy(f["latent_fp32"]), lat.cpu(), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def _flux_res(dev, case, c, key, fp8=False):
    f = load(case)
    pipe = flux_full_pipe(dev, fp8=fp8)
    text, pooled = fx.flux_full_inputs()
    lat, _ = pipe.denoise_latents(text.to(dev, BF), pooled.to(dev, BF), num_steps=c["steps"], cfg_weight=0.0, latent_size=c["latent"],
                                  seed=c["noise_seed"])
    assert tuple(lat.shape) == (1, c["latent"][0], c["latent"][1], 16)
    check(key, torch.from_numpy(f["latent_fp32"]), lat.cpu(), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def test_flux_schnell_512_full_depth_pipeline_fp8_weights(dev):
    """... the same image with e4m3 weights / MX-fp8 activations on every block Linear: the fp8 GEMM's K split (gemm256f8.hip: F8Split, round 6) inside the
    model -- fc2 / linear2 are 60 tiles here, four K ranges each"""
    _flux_res(dev, "flux_512", fx.FLUX_512, "flux_512_fp8_latent", fp8=True)


def test_flux_schnell_512_full_depth_pipeline_fp8_policy(dev):
    """... and with the shipped precision policy (first 12 double-stream blocks bf16)"""
    _flux_res(dev, "flux_512", fx.FLUX_512, "flux_512_fp8_policy_latent", fp8="quality")


def test_flux_schnell_768_full_depth_pipeline(dev):
    """Round 6: FLUX.1-schnell end to end at 768 x 768 (latent 96 x 96, S = 2560): fc2 / linear2 are 120 tiles -- cut in TWO K ranges -- and the attention
    launch is 240 blocks of the one-wave-per-SIMD kernel, one round of the CUs"""
    _flux_res(dev, "flux_768", fx.FLUX_768, "flux_768_latent")


def test_flux_schnell_1024_full_depth_pipeline_fp8_weights(dev):
    """BASELINE configs[1]'s image with configs[3]'s arithmetic: FLUX.1-schnell, 19 + 38 blocks, 4 Euler steps, e4m3 weights and
    MX-fp8 activations on every block Linear, against the fp32 oracle's latent (original weights)"""
    f = load("flux_full")
    c = fx.FLUX_FULL
    pipe = flux_full_pipe(dev, fp8=True)
    text, pooled = fx.flux_full_inputs()
    lat, _ = pipe.denoise_latents(text.to(dev, BF), pooled.to(dev, BF), num_steps=c["steps"], cfg_weight=0.0, latent_size=c["latent"],
                                  seed=c["noise_seed"])
    check("flux_full_fp8_latent", torch.from_numpy(f["latent_fp32"]), lat.cpu(), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


# ---- round 3: the configurations no parity test reached ("Next round" item 1) ----------------------------------
def test_flux_dev_shape_st512_vs_oracle(dev):
    """BASELINE configs[3]'s shape: 512 text tokens, S = 4608 (18 row tiles of 256, the FLUX.1-dev attention grid), FLUX width,
    1 double + 1 single block, bf16 weights"""
    f = load("flux_dev_512")
    check("flux_dev_512_final", torch.from_numpy(f["final_fp32"]), _forward(fx.FLUX_DEV_512, dev), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def test_flux_dev_shape_st512_fp8_weights_vs_oracle(dev):
    """... with configs[3]'s arithmetic: e4m3 weights and MX-fp8 activations on the block Linears"""
    from dataclasses import replace
    f = load("flux_dev_512")
    c = dict(fx.FLUX_DEV_512)
    c["cfg"] = replace(c["cfg"], weight_dtype="fp8_e4m3")
    check("flux_dev_512_fp8_final", torch.from_numpy(f["final_fp32"]), _forward(c, dev), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def test_sd3_medium_1024_full_depth_cfg_first_steps(dev):
    """BASELINE configs[2] at full depth: SD3-medium, 24 blocks, CFG 5.0 (B = 2), 589 text tokens, latent 128 x 128 -- the first
    three Euler steps of the 50-step schedule through the pipeline's own sample_euler / CFGDenoiser, against the fp32 oracle"""
    from diffusionkit_amd.pipeline import CFGDenoiser, DiffusionPipeline, sample_euler
    f = load("sd3_full_1024")
    c = fx.SD3_FULL_1024
    packed = {"mmdit": pack_mmdit(c["cfg"], synth_cached(c["cfg"], c["seed_w"]), dev, consume=True),
              "vae_decoder": pack_vae(VAEDecoderConfig(), synth_vae_weights(VAEDecoderConfig(), seed=4321), dev)}
    pipe = DiffusionPipeline(w16=True, a16=True, shift=c["shift"], device=dev, text_len=c["S_t"], packed_weights=packed)
    text, pooled = fx.sd3_full_inputs()
    # x0 and the schedule as denoise_latents builds them (mlx/__init__.py:253-292), cut to the first n_steps + 1 sigmas
    sig = pipe.get_sigmas(pipe.sampler, c["steps_of"])
    x_T = pipe.get_empty_latent(*c["latent"])
    noise = pipe.get_noise(c["noise_seed"], x_T)
    x0 = pipe.sampler.noise_scaling(np.float32(sig[0]), noise, x_T, pipe.max_denoise(sig))
    x0_ref, sig_ref = fx.sd3_full_start()
    assert np.allclose(np.asarray(sig[: c["n_steps"] + 1], dtype=np.float64), sig_ref.numpy().astype(np.float64), rtol=0, atol=1e-7)
    assert np.array_equal(np.asarray(x0, dtype=np.float32), x0_ref.numpy())
    x = torch.from_numpy(np.ascontiguousarray(x0, dtype=np.float32)).to(dev)
    extra = {"conditioning": text.to(dev, BF), "cfg_weight": c["cfg_weight"], "pooled_conditioning": pooled.to(dev, BF)}
    x1, _ = sample_euler(CFGDenoiser(pipe), x, sig[:2], extra_args=dict(extra))
    got1 = x1.float().cpu()
    print(f"[fullsize] sd3_full_1024 after step 1: PSNR {psnr(torch.from_numpy(f['x_step1_fp32']), got1):.2f} dB, rel-L2 {rel_l2(torch.from_numpy(f['x_step1_fp32']), got1):.3e}")
    x3, it = sample_euler(CFGDenoiser(pipe), x, sig[: c["n_steps"] + 1], extra_args=dict(extra))
    assert len(it) == c["n_steps"]
    check("sd3_full_1024_x3", torch.from_numpy(f["x_step3_fp32"]), x3.float().cpu(), f, ("emu_psnr", "emu_rel_l2", "emu_max_abs"))


def test_flux_full_size_blocks_teacher_forced(dev):
    """The FLUX gate that bites: single blocks of the full-size 57-block model (first double, last double, one
    single block; their own weights and modulation rows) teacher-forced through dk_mmdit_run_blocks on a seeded N(0, 1) joint
    stream of 4352 rows, against the fp32 oracle's output rows.  One block = one bf16 rounding chain, so the bar is per block:
    the error of the HIP result is held against the error the bf16-emulating oracle (the reference's own rounding points) makes on
    the same block, and against an absolute ceiling."""
    f = load("flux_blocks")
    c = fx.FLUX_BLOCKS
    eng = flux_full_pipe(dev).mmdit
    x, pooled = fx.flux_blocks_inputs()
    eng.prepare(1, c["latent"], c["S_t"], len(c["timesteps"]))
    eng.cache_modulation_params(pooled.to(dev), c["timesteps"])
    rows = torch.from_numpy(f["rows"]).long()
    xd = x.to(dev, BF)
    for g in c["blocks"]:
        y = eng.run_blocks(xd, c["step"], g, 1).float().cpu()
        ref = torch.from_numpy(f[f"block{g}_rows_fp32"])
        e = rel_l2(ref, y[0, rows])
        e_delta = rel_l2(ref - x[0, rows], y[0, rows] - x[0, rows])
        emu, emu_delta = float(f[f"block{g}_emu_rel_l2"]), float(f[f"block{g}_emu_rel_l2_delta"])
        print(f"[fullsize] flux block {g}: rel-L2 of the output stream {e:.3e} (bf16-emulating oracle {emu:.3e}), of the block's "
              f"contribution {e_delta:.3e} (emu {emu_delta:.3e}; contribution / output = {float(f[f'block{g}_delta_over_out']):.3f})")
        assert e <= 4.5e-3, f"block {g}: rel-L2 {e:.3e} > 4.5e-3"  # measured 2.8e-3 ... 3.2e-3
        assert e_delta <= 1.5 * emu_delta + 2e-3, f"block {g}: contribution error {e_delta:.3e} vs emulation {emu_delta:.3e}"
    # two consecutive blocks through the range entry = the two single calls chained (the range loop itself)
    y01 = eng.run_blocks(xd, c["step"], 18, 2)
    y0 = eng.run_blocks(xd, c["step"], 18, 1)
    assert torch.equal(eng.run_blocks(y0, c["step"], 19, 1), y01)


# ---- round 4: full depth beyond the schedule's first entries (VERDICT r3 "Next round" item 2) ------------------------------------
FORCED_TOL = {
    # case: (min PSNR dB, max rel-L2) of every step's Euler direction d_i against the fp32 oracle's
    "flux_dev_full": (44.4, 2.7e-2),      # measured 46.43-47.45 dB / 1.62e-2-1.78e-2 on the MODEL OUTPUT of one 57-block forward
    "flux_dev_full_fp8": (29.6, 1.5e-1),  # measured 31.62-32.53 dB / 8.6e-2-1.02e-1 (e4m3 weights + MX-fp8 activations against the un-quantised oracle)
    # round 5: the fp8 precision policy (first 12 double blocks bf16): the bar of SURVEY.md section 8c (iii), not measured - 2 dB -- measured 35.14-36.36 dB / 6.5e-2
    "flux_dev_full_fp8_policy": (35.0, 7.5e-2),
    "sd35_full": (44.5, 2.4e-2),       # round 6: SD3.5-large at full depth (38 blocks, width 2432, CFG 5): measured 46.52-47.66 dB / 1.60e-2
    "sd3_full_late": (46.0, 2.5e-2),   # measured 47.98-49.55 dB / 1.58e-2-1.69e-2 (CFG 5 amplifies; bf16-emulating oracle 47.6-49.2 dB / 1.64e-2-1.75e-2)
}


def check_forced(name, f, got, tol_key):
    worst_p, worst_e = 1e9, 0.0
    for i in sorted(got):
        ref = torch.from_numpy(f[f"d{i}_fp32_f16"].astype(np.float32))
        p, e = psnr(ref, got[i].float()), rel_l2(ref, got[i].float())
        emu = f", bf16-emulating oracle {float(f[f'd{i}_emu_psnr']):.2f} dB / {float(f[f'd{i}_emu_rel_l2']):.3e}" if f"d{i}_emu_psnr" in f.files else ""
        if f"d{i}_emu_fp16_psnr" in f.files:  # (round 6: the reference's own dtype for SD3, mlx/config.py:79 -- context, not a gate)
            emu += f", fp16-emulating oracle {float(f[f'd{i}_emu_fp16_psnr']):.2f} dB / {float(f[f'd{i}_emu_fp16_rel_l2']):.3e}"
        print(f"[fullsize] {name} step {i + 1} of 50: Euler direction PSNR {p:.2f} dB, rel-L2 {e:.4e} (|d| rms {float(f[f'd{i}_rms']):.3f}{emu})")
        worst_p, worst_e = min(worst_p, p), max(worst_e, e)
    min_p, max_e = FORCED_TOL[tol_key]
    assert worst_p >= min_p, f"{name}: PSNR {worst_p:.2f} dB < {min_p}"
    assert worst_e <= max_e, f"{name}: rel-L2 {worst_e:.3e} > {max_e}"


def test_flux_dev_full_depth_st512_forced_steps(dev):
    """BASELINE configs[3] at FULL depth (19 + 38 blocks, S_t = 512, S = 4608): Euler steps 1, 2, 49 and 50 of the 50-step schedule,
    each teacher-forced from a seeded latent at its sigma, bf16 weights -- the Euler direction of every step against the fp32 oracle"""
    f = load("flux_dev_full")
    check_forced("flux_dev_full", f, forced_steps(flux_full_pipe(dev), fx.FLUX_DEV_FULL, dev), "flux_dev_full")


def test_flux_dev_full_depth_st512_forced_steps_fp8_weights(dev):
    """... with configs[3]'s arithmetic (e4m3 weights, MX-fp8 activations on every block Linear) against the same fp32 oracle
    (original weights): the distance is the format's quantisation noise over 57 blocks, reported and gated at its measured level"""
    f = load("flux_dev_full")
    check_forced("flux_dev_full fp8", f, forced_steps(flux_full_pipe(dev, fp8=True), fx.FLUX_DEV_FULL, dev), "flux_dev_full_fp8")


def test_flux_dev_full_depth_st512_forced_steps_fp8_policy(dev):
    """... and the fp8 path as configs[3] ships it (config.fp8_config(cfg, "quality"), round 5): the first 12 of the 19 double-stream
    blocks keep bf16 Linears, the other 7 and the 38 single blocks run e4m3 weights / MX-fp8 activations -- every step at >= 35 dB"""
    f = load("flux_dev_full")
    check_forced("flux_dev_full fp8 policy", f, forced_steps(flux_full_pipe(dev, fp8="quality"), fx.FLUX_DEV_FULL, dev), "flux_dev_full_fp8_policy")


def test_sd3_medium_1024_full_depth_cfg_late_steps(dev):
    """BASELINE configs[2] at full depth (24 blocks, B = 2, CFG 5.0, 589 text tokens): steps 1, 25, 49 and 50 of the 50-step
    schedule, teacher-forced -- the late steps run the fp16-rounded small timesteps (66.9, 8.93) no other full-depth case reaches"""
    from diffusionkit_amd.pipeline import DiffusionPipeline
    f = load("sd3_full_late")
    c = fx.SD3_FULL_LATE
    packed = {"mmdit": pack_mmdit(c["cfg"], synth_cached(c["cfg"], c["seed_w"]), dev, consume=True)}
    pipe = DiffusionPipeline(w16=True, a16=True, shift=c["shift"], device=dev, text_len=c["S_t"], packed_weights=packed)
    check_forced("sd3_full_late", f, forced_steps(pipe, c, dev), "sd3_full_late")


def test_sd35_large_1024_full_depth_cfg_forced_steps(dev):
    """Round 6 (VERDICT r5 missing 9): the reference's third model family (mlx/config.py:72-74) at FULL depth -- SD3.5-large, 38 blocks of width 2432
    (N % 256 = 128: the half column tile of gemm256v3.hip on every Linear), 38 heads of 64, QK-norm, B = 2 (CFG 5.0), 589 text tokens, latent 128 x 128:
    the first and the last step of the 50-step schedule, teacher-forced, Euler direction against the fp32 oracle"""
    from diffusionkit_amd.pipeline import DiffusionPipeline
    f = load("sd35_full")
    c = fx.SD35_FULL
    packed = {"mmdit": pack_mmdit(c["cfg"], synth_cached(c["cfg"], c["seed_w"]), dev, consume=True)}
    pipe = DiffusionPipeline(w16=True, a16=True, shift=c["shift"], device=dev, text_len=c["S_t"], packed_weights=packed, mmdit_config=c["cfg"],
                             model_version="argmaxinc/mlx-stable-diffusion-3.5-large")
    check_forced("sd35_full", f, forced_steps(pipe, c, dev), "sd35_full")
    del pipe, packed
    torch.cuda.empty_cache()


# ---- round 5: closed-loop trajectories of the 50-step configurations (VERDICT r4 "Next round" item 2) ---------------------------------
CLOSED_TOL = {
    # case: (min PSNR dB, max rel-L2) of the FINAL latent against the fp32 oracle's, every gate = measured - 2 dB / x 1.5
    # (profiles/r05_fullsize_parity.log); the intermediate latents are printed (how the distance grows along the trajectory)
    "sd3_full_50": (49.5, 1.53e-2),     # measured 51.52 dB / 1.018e-2 after 50 steps (78.6 dB after step 1, 65.3 after 10, 56.5 after 30: rel-L2 grows ~ linearly)
    "sd3_full_50_image": (44.4, None),  # measured 46.41 dB with the reference's own image metric (its gate for a trained model: 20 dB, tests/mlx/test_diffusion_pipeline.py:20)
    "flux_dev_10": (54.4, 1.01e-2),             # measured 56.49 dB / 6.73e-3 (bf16 weights, 10 closed-loop steps, sigma 1 -> 0)
    "flux_dev_10_fp8": (36.8, 7.7e-2),           # measured 38.83 dB / 5.14e-2 (every block Linear in fp8: ten ~32 dB steps)
    "flux_dev_10_fp8_policy": (40.9, 4.8e-2),    # measured 42.93 dB / 3.21e-2 (the shipped precision policy: first 12 double blocks bf16)
    # round 6: the same at configs[3]'s STATED length, all 50 steps closed loop (profiles/r06_fullsize_parity.log); gates = measured - 2 dB / x 1.5.  Fifty small
    # steps end CLOSER to the oracle than ten large ones: the latent integrates the directions with weights dt that sum to one, each taken nearer its neighbour
    "flux_dev_50": (56.1, 8.2e-3),              # measured 58.11 dB / 5.46e-3 (bf16 weights)
    "flux_dev_50_fp8": (37.4, 7.1e-2),           # measured 39.44 dB / 4.68e-2 (every block Linear in fp8: fifty ~32 dB steps)
    "flux_dev_50_fp8_policy": (41.4, 4.5e-2),    # measured 43.44 dB / 2.95e-2 (configs[3] as shipped: first 12 double blocks bf16, every step >= 35 dB)
}


def check_closed(name, ref, got, key):
    p, e = psnr(ref, got), rel_l2(ref, got)
    print(f"[fullsize] {name}: final latent PSNR {p:.2f} dB, rel-L2 {e:.4e}, max-abs {float((ref.double() - got.double()).abs().max()):.4g}")
    min_p, max_e = CLOSED_TOL[key]
    assert p >= min_p, f"{name}: PSNR {p:.2f} dB < {min_p}"
    assert max_e is None or e <= max_e, f"{name}: rel-L2 {e:.3e} > {max_e}"


def test_sd3_medium_1024_closed_loop_50_steps(dev):
    """BASELINE configs[2] END TO END, closed loop: SD3-medium, 24 blocks, CFG 5.0 (B = 2), 589 text tokens, latent 128 x 128, all 50 Euler
    steps of the 50-step schedule through the pipeline's own sample_euler / CFGDenoiser (mlx/__init__.py:761-788), every step from the
    latent the previous one left -- the latent after steps 1 / 3 / 10 / 20 / 30 / 40 / 50 against the fp32 oracle's trajectory, and the
    decoded 1024 x 1024 image (uint8, the reference's image PSNR metric utils.py:52-67) against the oracle's decode of ITS final latent"""
    from diffusionkit_amd.pipeline import CFGDenoiser, DiffusionPipeline, sample_euler
    from oracle.pipeline import image_psnr
    f = load("sd3_full_50")
    c = fx.SD3_FULL_50
    packed = {"mmdit": pack_mmdit(c["cfg"], synth_cached(c["cfg"], c["seed_w"]), dev, consume=True),
              "vae_decoder": pack_vae(VAEDecoderConfig(), synth_vae_weights(VAEDecoderConfig(), seed=c["seed_vae"]), dev)}
    pipe = DiffusionPipeline(w16=True, a16=True, shift=c["shift"], device=dev, text_len=c["S_t"], packed_weights=packed)
    text, pooled = fx.sd3_full_inputs()
    sig = pipe.get_sigmas(pipe.sampler, c["steps_of"])
    x0_ref, sig_ref = fx.sd3_full_start(c)
    # (the pipeline's and the oracle's schedules agree to 1.1e-7: the two evaluate sampler.py:31-35 in different orders of fp32 operations)
    assert np.allclose(np.asarray(sig, dtype=np.float64), sig_ref.numpy().astype(np.float64), rtol=0, atol=2e-7)
    extra = {"conditioning": text.to(dev, BF), "cfg_weight": c["cfg_weight"], "pooled_conditioning": pooled.to(dev, BF)}
    x = x0_ref.to(dev)
    prev = 0
    for k in c["keep"]:  # the loop in pieces: same modulation table rows (step indices of the full schedule), same arithmetic
        x, _ = sample_euler(CFGDenoiser(pipe), x, sig[prev: k + 1], extra_args=dict(extra))  # (timesteps = the schedule's own entries)
        prev = k
        ref = torch.from_numpy(f["x_step50_fp32"] if k == 50 else f[f"x_step{k}_f16"].astype(np.float32))
        got = x.float().cpu()
        emu = f", bf16-emulating oracle {float(f[f'emu_psnr_step{k}']):.2f} dB" if f"emu_psnr_step{k}" in f.files else ""
        print(f"[fullsize] sd3_full_50 after step {k:2d}: PSNR {psnr(ref, got):.2f} dB, rel-L2 {rel_l2(ref, got):.3e} (latent rms {float(ref.pow(2).mean().sqrt()):.3f}{emu})")
    check_closed("sd3_full_50", torch.from_numpy(f["x_step50_fp32"]), x.float().cpu(), "sd3_full_50")
    img, u8, _ = pipe.decoder.decode(pipe.latent_format.process_out(x))
    p_img = image_psnr(f["image_u8"], u8[0].cpu().numpy())
    print(f"[fullsize] sd3_full_50 decoded image (uint8, the reference's image_psnr): {p_img:.2f} dB")
    assert p_img >= CLOSED_TOL["sd3_full_50_image"][0]


def _flux_dev_10(dev, fp8, key, case="flux_dev_10"):
    f = load(case)
    c = fx.FLUX_DEV_10 if case == "flux_dev_10" else fx.FLUX_DEV_50
    pipe = flux_full_pipe(dev, fp8=fp8)
    text, pooled = fx.flux_dev_10_inputs()
    lat, it = pipe.denoise_latents(text.to(dev, BF), pooled.to(dev, BF), num_steps=c["steps"], cfg_weight=0.0, latent_size=c["latent"], seed=c["noise_seed"])
    assert len(it) == c["steps"]
    check_closed(f"{case} ({key})", torch.from_numpy(f["latent_fp32"]), lat.float().cpu(), key)


def test_flux_dev_shape_closed_loop_10_steps(dev):
    """BASELINE configs[3]'s shape closed loop: 19 + 38 blocks, S_t = 512, a COMPLETE 10-step schedule (sigma 1 -> 0) through denoise_latents,
    bf16 weights -- final latent against the fp32 oracle's"""
    _flux_dev_10(dev, False, "flux_dev_10")


def test_flux_dev_shape_closed_loop_10_steps_fp8(dev):
    """... every block Linear in fp8 (e4m3 weights, MX-fp8 activations): what ten 32 dB steps do to the image"""
    _flux_dev_10(dev, True, "flux_dev_10_fp8")


def test_flux_dev_shape_closed_loop_10_steps_fp8_policy(dev):
    """... and the shipped precision policy (first 12 double-stream blocks bf16)"""
    _flux_dev_10(dev, "quality", "flux_dev_10_fp8_policy")


def test_flux_dev_shape_closed_loop_50_steps(dev):
    """Round 6: BASELINE configs[3] at its STATED length, closed loop -- 19 + 38 blocks, S_t = 512, all 50 Euler steps of the 50-step
    schedule through denoise_latents, every step from the latent the previous one left; bf16 weights, final latent against the fp32 oracle's
    (tests/golden/fullsize_flux_dev_50.npz: 2.8 h of the CPU oracle)"""
    _flux_dev_10(dev, False, "flux_dev_50", "flux_dev_50")


def test_flux_dev_shape_closed_loop_50_steps_fp8_policy(dev):
    """... configs[3] as shipped: e4m3 weights / MX-fp8 activations with the precision policy (first 12 double-stream blocks bf16): what fifty ~35 dB
    steps accumulate to"""
    _flux_dev_10(dev, "quality", "flux_dev_50_fp8_policy", "flux_dev_50")


def test_flux_dev_shape_closed_loop_50_steps_fp8(dev):
    """... and every block Linear in fp8 (fifty ~32 dB steps)"""
    _flux_dev_10(dev, True, "flux_dev_50_fp8", "flux_dev_50")


def test_eight_seeds_one_step_loop_equal_single_runs(dev):
    """BASELINE configs[4]'s per-GPU shape at production width: 8 images in ONE batched step loop (M = 8 x 4352 = 34 816 rows per
    Linear; FLUX width, 1 double + 1 single block, 2 Euler steps) must equal the eight single-seed runs"""
    from dataclasses import replace
    from diffusionkit_amd.pipeline import FluxPipeline
    cfg = replace(fx.FLUX_FULL["cfg"], depth_multimodal=1, depth_unified=1)
    packed = {"mmdit": pack_mmdit(cfg, synth_mmdit_weights(cfg, seed=1234), dev, consume=True)}
    pipe = FluxPipeline(w16=True, a16=True, device=dev, text_len=256, packed_weights=packed, mmdit_config=cfg)
    text, pooled = fx.flux_full_inputs()
    text, pooled = text.to(dev, BF), pooled.to(dev, BF)
    seeds = list(range(40, 48))
    lat8, _ = pipe.denoise_latents(text, pooled, num_steps=2, cfg_weight=0.0, latent_size=(128, 128), seed=seeds)
    assert lat8.shape == (8, 128, 128, 16)
    worst = 0.0
    for i, sd in enumerate(seeds):
        lat1, _ = pipe.denoise_latents(text, pooled, num_steps=2, cfg_weight=0.0, latent_size=(128, 128), seed=sd)
        worst = max(worst, rel_l2(lat1[0].cpu(), lat8[i].cpu()))
    print(f"[fullsize] 8 seeds in one step loop vs single runs: worst rel-L2 {worst:.3e}")
    assert worst <= 4e-3  # measured 2.5e-3 (the batched GEMMs pick other tile heights)


def test_vae_decode_batch8_equals_single_decodes(dev):
    """... and its decode: 8 latents of 128 x 128 in ONE VAE decode (activation buffers of 8 x 1024 x 1024 x 128 bf16 = 2.1 GB,
    past 2^31 bytes) against eight single decodes.  With the K split of the under-filled stages off, both run the same kernels in
    the same summation order: bit for bit.  Under the default launch rules the batch changes which stages split K: equal to
    rounding."""
    from diffusionkit_amd import _lib
    from diffusionkit_amd.engine import VAEDecoderEngine
    vcfg = VAEDecoderConfig()
    eng = VAEDecoderEngine(vcfg, pack_vae(vcfg, synth_vae_weights(vcfg, seed=4321), dev))
    z = torch.cat([fx.randn(1, 128, 128, 16, seed=200 + i) for i in range(8)], 0).to(dev)
    lib = _lib.load()
    try:
        _lib.check(lib.dk_tune_set(b"gemm_split", 0), "tune")
        img8, u88, raw8 = eng.decode(z, want_raw=True)
        for i in range(8):
            img1, u81, raw1 = eng.decode(z[i:i + 1], want_raw=True)
            assert torch.equal(raw1[0], raw8[i]), f"image {i}: decoder output differs between the batched and the single decode"
            assert torch.equal(u81[0], u88[i]) and torch.equal(img1[0], img8[i])
    finally:
        _lib.check(lib.dk_tune_set(b"gemm_split", -1), "tune")
    img8d, u88d, raw8d = eng.decode(z, want_raw=True)
    e = rel_l2(raw8[..., :3].float().cpu(), raw8d[..., :3].float().cpu())
    frac = float((u88d != u88).float().mean())
    print(f"[fullsize] 8-image decode, default launch rules vs split-free: raw rel-L2 {e:.3e}, uint8 pixels that differ {frac:.2e}")
    assert e <= 3e-3 and int((u88d.int() - u88.int()).abs().max()) <= 1
