"""Full-size / full-depth oracle fixtures for the BASELINE.json configurations ("Next round" item 1).

The CPU oracle needs minutes to hours at these sizes, so it is run ONCE (here, 8 host cores) and its fp32 outputs are committed;
tests/test_gpu_fullsize.py regenerates the same seeded weights and inputs on the GPU box and compares the HIP path with them.
Every case also records what the bf16-emulating oracle (the model of the reference's MLX rounding points) reaches against the
fp32 oracle, as information: the tolerances the tests hold are absolute and stated per case in DESIGN.md section 4.

  sd3_512     BASELINE configs[0]: SD3-medium, all 24 blocks, latent 64 x 64 (512 x 512), 4 Euler steps, CFG off,
              77 + 512 text tokens, + VAE decode at the production channel plan -> final latent, decoded image
  vae_1024    one full-size decode, latent 128 x 128 -> 1024 x 1024 (mid attention T = 16384, the split-K P.V path)
  sd3_1024    SD3-medium at the bench shape: B = 2 (CFG pair), S_t = 589, latent 128 x 128, depth 2 -> model output
  flux_1024   FLUX.1-schnell geometry at S = 256 + 4096, depth 4 + 8 -> model output
  flux_full   BASELINE configs[1] end to end: FLUX.1-schnell, all 19 + 38 blocks, latent 128 x 128 (1024 x 1024), 4 Euler
              steps, 256 text tokens -> final latent (fp32 oracle only: ~15 min on 8 cores; pass "emu" as well for the
              bf16-emulating run, ~1 h)

Round 3 (VERDICT r2 "Next round" item 1: the configurations no parity test reached, and a FLUX gate that bites):
  flux_dev_512   BASELINE configs[3]'s shape: FLUX width, 1 double + 1 single block, S_t = 512 (S = 4608: 18 row tiles, the
                 attention grid of FLUX.1-dev) -> model output; replayed with bf16 and with fp8 weights
  sd3_full_1024  BASELINE configs[2] at full depth: SD3-medium, all 24 blocks, B = 2 (CFG 5.0), 589 text tokens, latent 128 x 128,
                 the first 3 Euler steps of the 50-step schedule -> latent after step 3 (and after step 1)
  flux_blocks    teacher-forced single blocks of the full-size FLUX.1-schnell model (first double, last double, one single
                 block of the seeded 57-block weight set) on a seeded N(0, 1) joint stream of S = 4352 rows -> selected rows of
                 the block's output stream; a wrong fragment in one block cannot hide behind the conditioning of the stack

Round 4 (VERDICT r3 "Next round" item 2: full depth beyond the schedule's first entries; gates that bite):
  flux_dev_full  BASELINE configs[3] at FULL depth: 19 + 38 blocks at FLUX width, S_t = 512 (S = 4608), the 50-step schedule -- Euler steps
                 1, 2, 49 and 50, each TEACHER-FORCED from a seeded latent x_i = sigma_i * noise + (1 - sigma_i) * x_clean (the
                 reference's noise_scaling form, sampler.py:41-42), so that the oracle runs 4 forwards instead of 50 -> the Euler
                 direction d_i = (x_i - denoised) / sigma_i of every step (what a step adds to the latent, not the latent it is added
                 to); replayed with bf16 and with fp8 weights
  sd3_full_late  BASELINE configs[2] at full depth (24 blocks, B = 2, CFG 5.0, 589 text tokens): steps 1, 25, 49 and 50 of the 50-step
                 schedule the same way (late steps: small sigma, fp16-rounded timesteps 8.93 / 66.9)

Round 5 (VERDICT r4 "Next round" item 2: CLOSED-LOOP runs of the 50-step configurations; error accumulation over a whole trajectory):
  sd3_full_50    BASELINE configs[2] end to end: SD3-medium, 24 blocks, B = 2 (CFG 5.0), 589 text tokens, latent 128 x 128, ALL 50 Euler
                 steps of the 50-step schedule, each from the latent the previous one left (fp32 oracle, ~1 h on 6 cores) -> the latent
                 after steps 1, 3, 10, 20, 30, 40 (fp16) and 50 (fp32), and the decoded 1024 x 1024 image (fp32 oracle decode of the
                 bf16-rounded final latent, uint8)
  flux_dev_10    BASELINE configs[3]'s shape closed loop: 19 + 38 blocks at FLUX width, S_t = 512, a complete 10-step schedule
                 (sigma 1 -> 0) -> the latent after steps 1, 2, 5 (fp16) and 10 (fp32); replayed with bf16 and with fp8 weights

Round 6: flux_768 -- the same at 768 x 768 (latent 96 x 96, S = 2560: the two-range K split and the one-round attention5 launch), fp32 oracle
Round 6: flux_512 -- FLUX.1-schnell, all 57 blocks, 4 Euler steps at 512 x 512 (latent 64 x 64, S = 1280: the reference CLI's default resolution) -> final
                 latent, fp32 oracle + the bf16-emulating one
Round 6 (VERDICT r5 missing 9): sd35_full -- SD3.5-large (38 blocks, width 2432, QK-norm) at full depth, B = 2, CFG 5.0, steps 1 and 50 of the
                 50-step schedule teacher-forced (the Euler direction of each, fp32 oracle)
Round 6 (VERDICT r5 "Next round" item 4: configs[3] closed loop at its STATED length):
  flux_dev_50    the same model, conditioning and start noise over the complete 50-step schedule (fp32 oracle, ~3 h on 5 cores) -> the
                 latent after steps 1, 2, 5, 10, 20, 30, 40 (fp16) and 50 (fp32); replayed with bf16, the fp8 policy and every block fp8

The fp32 oracle of the FLUX cases and of every round-3 case is the reference's function with fp32 ACTIVATIONS: its timestep
embedding is still evaluated in config.dtype (mmdit.py:379-389, quirk Q2; bf16 for FLUX, fp16 for SD3) -- ``ref_model`` below.
Round 2's FLUX fixtures used an oracle with an exact embedding and measured, at 27-32 dB, the distance between two different
modulation tables rather than rounding noise (oracle/mmdit.py, ``embed_prec``); flux_1024 and flux_full were regenerated in
round 3 with the quirk in place.  The SD3 fixtures of round 2 (sd3_512, sd3_1024: fp16 embedding, 52 dB either way) are unchanged.

Run from the repo root (each case separately, they take minutes):
    python tests/golden/make_fullsize_fixtures.py sd3_512 vae_1024 sd3_1024 flux_1024
"""
import os
import sys
import time
from dataclasses import replace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from diffusionkit_amd.config import FLUX_SCHNELL, SD3_2b, SD3_8b, VAEDecoderConfig  # noqa: E402
from diffusionkit_amd.weights import synth_mmdit_weights, synth_vae_weights  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from oracle.mmdit import OracleMMDiT, Prec, embed_dtype  # noqa: E402
from oracle.vae import OracleVAEDecoder  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
BF = torch.bfloat16


def randn(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(BF).float()


def rel_l2(ref, got):
    ref, got = ref.double().reshape(-1), got.double().reshape(-1)
    return float(torch.linalg.norm(ref - got) / (torch.linalg.norm(ref) + 1e-30))


def psnr(ref, got):
    ref, got = ref.double().numpy().ravel(), got.double().numpy().ravel()
    return float(20 * np.log10((np.abs(ref).max() + 1e-5) / (np.sqrt(np.mean((ref - got) ** 2)) + 1e-10)))


# ---- case definitions (shared with tests/test_gpu_fullsize.py) -------------------------------------------------
SD3_512 = dict(cfg=SD3_2b, seed_w=1234, seed_vae=4321, latent=(64, 64), S_t=77 + 512, steps=4, shift=3.0, noise_seed=0)
VAE_1024 = dict(cfg=VAEDecoderConfig(), seed_vae=4321, latent=(128, 128), z_seed=13)
SD3_1024 = dict(cfg=replace(SD3_2b, depth_multimodal=2, hidden_size_override=1536), seed_w=1234, B=2, latent=(128, 128), S_t=589,
                timesteps=[1000.0, 857.5], step=1)
FLUX_FULL = dict(cfg=FLUX_SCHNELL, seed_w=1234, latent=(128, 128), S_t=256, steps=4, shift=1.0, noise_seed=0)
FLUX_1024 = dict(cfg=replace(FLUX_SCHNELL, depth_multimodal=4, depth_unified=8), seed_w=1234, B=1, latent=(128, 128), S_t=256,
                 timesteps=[1000.0, 752.0], step=1)


# ---- round 3 cases ------------------------------------------------------------------------------------------------------
FLUX_DEV_512 = dict(cfg=replace(FLUX_SCHNELL, depth_multimodal=1, depth_unified=1), seed_w=1234, B=1, latent=(128, 128), S_t=512,
                    timesteps=[1000.0, 752.0], step=1)
SD3_FULL_1024 = dict(cfg=SD3_2b, seed_w=1234, latent=(128, 128), S_t=589, steps_of=50, n_steps=3, shift=3.0, cfg_weight=5.0, noise_seed=0)
# global block indices of the full model (doubles 0..18, singles 19..56): first double, last double, a single block
FLUX_BLOCKS = dict(cfg=FLUX_SCHNELL, seed_w=1234, latent=(128, 128), S_t=256, timesteps=[1000.0, 752.0], step=1, blocks=(0, 18, 39),
                   x_seed=31, rows=tuple(list(range(0, 16)) + list(range(240, 272)) + list(range(2296, 2312)) + list(range(4336, 4352))))


# one double + one single block at width (every launch shape of the bench, bf16 and fp8): the oracle outputs the two "width block
# pair" GPU tests compare against -- the fp32 and bf16-emulating oracles, and both again on the fake-quantised weights / activations
# of the fp8 path (oracle/fp8.py); every 4th image token is kept
FLUX_PAIR = dict(cfg=replace(FLUX_SCHNELL, depth_multimodal=1, depth_unified=1), seed_w=1234, B=1, latent=(128, 128), S_t=256,
                 timesteps=[1000.0, 752.0], step=1, row_stride=4)


# ---- round 4 cases: teacher-forced Euler steps of the 50-step schedules at full depth -------------------------------------------
FLUX_DEV_FULL = dict(cfg=FLUX_SCHNELL, seed_w=1234, latent=(128, 128), S_t=512, steps_of=50, shift=1.0, cfg_weight=0.0, flux=True,
                     step_ids=(0, 1, 48, 49), noise_seed=0, clean_seed=91, text_seed=73, rows=1)
SD3_FULL_LATE = dict(cfg=SD3_2b, seed_w=1234, latent=(128, 128), S_t=589, steps_of=50, shift=3.0, cfg_weight=5.0, flux=False,
                     step_ids=(0, 24, 48, 49), noise_seed=0, clean_seed=92, text_seed=81, rows=2)


# ---- round 5 cases: closed-loop trajectories ---------------------------------------------------------------------------------------
SD3_FULL_50 = dict(SD3_FULL_1024, n_steps=50, keep=(1, 3, 10, 20, 30, 40, 50), seed_vae=4321)
FLUX_DEV_10 = dict(cfg=FLUX_SCHNELL, seed_w=1234, latent=(128, 128), S_t=512, steps=10, shift=1.0, noise_seed=0, keep=(1, 2, 5, 10),
                   text_seed=73)


# ---- round 6 case: the reference's third model family at FULL depth (VERDICT r5 missing 9; mlx/config.py:72-74): SD3.5-large, 38 blocks of width 2432
# (9.5 column tiles of 256: the GEMM's half-tile path), 38 heads of 64, QK-norm, B = 2 (CFG 5.0), 589 text tokens, latent 128 x 128 -- the first and the
# last Euler step of the 50-step schedule, teacher-forced like sd3_full_late
SD35_FULL = dict(cfg=SD3_8b, seed_w=1234, latent=(128, 128), S_t=589, steps_of=50, shift=3.0, cfg_weight=5.0, flux=False,
                 step_ids=(0, 49), noise_seed=0, clean_seed=93, text_seed=83, rows=2)


# ---- round 6 case: FLUX.1-schnell end to end at the resolution the reference's CLI defaults to (generate_images.py:15-30): latent 64 x 64, S = 1280 --
# the launches the round's small-launch rules changed (fc2 / linear2 cut along K inside the model)
FLUX_512 = dict(FLUX_FULL, latent=(64, 64))
FLUX_768 = dict(FLUX_FULL, latent=(96, 96))   # S = 256 + 2304 = 2560: 120-tile launches cut in TWO K ranges, attention5 in one round of 240 blocks


# ---- round 6 case: configs[3]'s shape closed loop at its STATED length ---------------------------------------------
FLUX_DEV_50 = dict(FLUX_DEV_10, steps=50, keep=(1, 2, 5, 10, 20, 30, 40, 50))


def forced_inputs(c):
    """conditioning rows and, per step id i, (x_i, sigma_i, sigma_{i+1}) of a teacher-forced step: x_i is what noise_scaling
    (sampler.py:41-42) makes of a seeded clean latent at sigma_i"""
    cfg = c["cfg"]
    text = randn(c["rows"], c["S_t"], cfg.token_level_text_embed_dim, seed=c["text_seed"])
    pooled = randn(c["rows"], cfg.pooled_text_embed_dim, seed=c["text_seed"] + 1)
    sig = op.get_sigmas(c["shift"], c["flux"], c["steps_of"])
    noise = op.get_noise(c["noise_seed"], *c["latent"])
    clean = 0.8 * randn(1, c["latent"][0], c["latent"][1], 16, seed=c["clean_seed"])
    steps = []
    for i in c["step_ids"]:
        s0 = sig[i]
        steps.append((i, (s0 * noise + (1.0 - s0) * clean).to(torch.float32), sig[i: i + 2].clone()))
    return text, pooled, steps


def euler_direction(x_i, x_next, sig2):
    """d_i = (x_i - denoised) / sigma_i, recovered from one Euler step x_next = x_i + d_i * (sigma_{i+1} - sigma_i)
    (mlx/__init__.py:756,778-781)"""
    return (x_next.double() - x_i.double()) / (float(sig2[1]) - float(sig2[0]))


def make_forced(c, name, with_emu):
    cfg = c["cfg"]
    named = synth_mmdit_weights(cfg, seed=c["seed_w"])
    big = cfg.is_flux or cfg.depth_multimodal > 24  # (12 B / 8 B parameters: fp32 copies are handed out tensor by tensor)
    w = LazyFloat(named) if big else {k: v.float() for k, v in named.items()}
    text, pooled, steps = forced_inputs(c)
    t_act = None if c["flux"] else Prec(torch.float16)
    out = {"step_ids": np.asarray(c["step_ids"])}
    for pname, P in (("fp32", Prec()),) + ((("emu", Prec(BF)),) if with_emu else ()):
        m = ref_model(cfg, w, P)
        for i, x_i, sig2 in steps:
            t0 = time.time()
            x_next = op.sample_euler(m, x_i, sig2, text, pooled, c["cfg_weight"], Prec(BF), t_act=t_act)
            d = euler_direction(x_i, x_next, sig2)
            print(f"{name} {pname} step {i + 1}: {time.time() - t0:.0f} s, |d| rms {float(d.pow(2).mean().sqrt()):.3f}", flush=True)
            if pname == "fp32":
                out[f"d{i}_fp32_f16"] = d.numpy().astype(np.float16)
                out[f"d{i}_rms"] = np.float64(d.pow(2).mean().sqrt())
                keep = d
                out.setdefault("_keep", {})[i] = keep
            else:
                ref = out["_keep"][i]
                out[f"d{i}_emu_rel_l2"] = np.float64(rel_l2(ref, d))
                out[f"d{i}_emu_psnr"] = np.float64(psnr(ref, d))
    out.pop("_keep", None)
    return out


def ref_model(cfg, w, P):
    """the oracle with the reference's config-dtype timestep embedding (quirk Q2) whatever the activation precision ``P``"""
    return OracleMMDiT(cfg, w, P, embed_prec=Prec(embed_dtype(cfg)))


def sd3_full_inputs():
    c = SD3_FULL_1024
    text = randn(2, c["S_t"], c["cfg"].token_level_text_embed_dim, seed=81)   # rows: [prompt, negative] (encode_text's layout)
    pooled = randn(2, c["cfg"].pooled_text_embed_dim, seed=82)
    return text, pooled


def sd3_full_start(c=None):
    """x0 and the truncated schedule exactly as DiffusionPipeline.denoise_latents builds them (mlx/__init__.py:253-292)"""
    c = c or SD3_FULL_1024
    sig = op.get_sigmas(c["shift"], False, c["steps_of"])
    noise = op.get_noise(c["noise_seed"], *c["latent"])
    x0 = sig[0] * noise + (1.0 - sig[0]) * op.get_empty_latent(*c["latent"])
    return x0, sig[: c["n_steps"] + 1]


def flux_blocks_inputs():
    c = FLUX_BLOCKS
    cfg = c["cfg"]
    S = c["S_t"] + (c["latent"][0] // 2) * (c["latent"][1] // 2)
    x = randn(1, S, cfg.hidden_size, seed=c["x_seed"])  # the joint stream [text rows, image rows], bf16-representable
    pooled = randn(1, cfg.pooled_text_embed_dim, seed=c["x_seed"] + 1)
    return x, pooled


def sd3_512_inputs():
    c = SD3_512
    text = randn(1, c["S_t"], c["cfg"].token_level_text_embed_dim, seed=7)
    pooled = randn(1, c["cfg"].pooled_text_embed_dim, seed=8)
    return text, pooled


def forward_inputs(c):
    cfg = c["cfg"]
    text = randn(c["B"], c["S_t"], cfg.token_level_text_embed_dim, seed=3)
    pooled = randn(c["B"], cfg.pooled_text_embed_dim, seed=4)
    lat = randn(c["B"], c["latent"][0], c["latent"][1], 16, seed=5)
    return text, pooled, lat


def make_sd3_512():
    c = SD3_512
    cfg = c["cfg"]
    w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=c["seed_w"]).items()}
    vw = {k: v.float() for k, v in synth_vae_weights(VAEDecoderConfig(), seed=c["seed_vae"]).items()}
    text, pooled = sd3_512_inputs()
    out = {}
    lat = {}
    for pname, P in (("fp32", Prec()), ("emu", Prec(BF))):
        t0 = time.time()
        trace = []
        lat[pname] = op.denoise_latents(OracleMMDiT(cfg, w, P), text, pooled, c["steps"], 0.0, c["latent"], c["noise_seed"], c["shift"],
                                        False, Prec(BF), trace=trace, t_act=Prec(torch.float16))
        print(f"sd3_512 {pname}: {time.time() - t0:.0f} s", flush=True)
        if pname == "fp32":
            out["latent_fp32"] = lat[pname].numpy()
            out["trace_step0_fp32"] = trace[0].numpy()  # latent after the first Euler step (sd3 latent space before process_out)
    out["emu_rel_l2"] = np.float64(rel_l2(lat["fp32"], lat["emu"]))
    out["emu_psnr"] = np.float64(psnr(lat["fp32"], lat["emu"]))
    out["emu_max_abs"] = np.float64((lat["fp32"] - lat["emu"]).abs().max())
    # decode of the fp32 latent (rounded to bf16 like the engine's input) by the fp32 oracle
    z = lat["fp32"].to(BF).float()
    t0 = time.time()
    raw = OracleVAEDecoder(VAEDecoderConfig(), vw, Prec())(z)
    print(f"sd3_512 decode: {time.time() - t0:.0f} s", flush=True)
    img = torch.clip(raw / 2 + 0.5, 0, 1)
    out["image_fp32_f16"] = img.numpy().astype(np.float16)
    return out


class LazyFloat(dict):
    """bf16 weight dict that hands out fp32 copies on access (the full FLUX model does not fit the host twice)."""

    def __getitem__(self, k):
        return dict.__getitem__(self, k).float()

    def get(self, k, default=None):
        return self[k] if k in self else default


def flux_full_inputs():
    c = FLUX_FULL
    text = randn(1, c["S_t"], c["cfg"].token_level_text_embed_dim, seed=71)
    pooled = randn(1, c["cfg"].pooled_text_embed_dim, seed=72)
    return text, pooled


def make_flux_full(with_emu=False, c=None, name="flux_full"):
    c = c or FLUX_FULL
    cfg = c["cfg"]
    w = LazyFloat(synth_mmdit_weights(cfg, seed=c["seed_w"]))
    text, pooled = flux_full_inputs()
    out, lat = {}, {}
    for pname, P in (("fp32", Prec()),) + ((("emu", Prec(BF)),) if with_emu else ()):
        t0 = time.time()
        trace = []
        lat[pname] = op.denoise_latents(ref_model(cfg, w, P), text, pooled, c["steps"], 0.0, c["latent"], c["noise_seed"], c["shift"],
                                        True, Prec(BF), trace=trace)
        print(f"{name} {pname}: {time.time() - t0:.0f} s", flush=True)
        if pname == "fp32":
            out["latent_fp32"] = lat[pname].numpy()
            out["trace_step0_fp32"] = trace[0].numpy()
    if with_emu:
        out["emu_rel_l2"] = np.float64(rel_l2(lat["fp32"], lat["emu"]))
        out["emu_psnr"] = np.float64(psnr(lat["fp32"], lat["emu"]))
        out["emu_max_abs"] = np.float64((lat["fp32"] - lat["emu"]).abs().max())
    return out


def make_vae_1024():
    c = VAE_1024
    vw = {k: v.float() for k, v in synth_vae_weights(c["cfg"], seed=c["seed_vae"]).items()}
    z = randn(1, c["latent"][0], c["latent"][1], 16, seed=c["z_seed"])
    out = {}
    res = {}
    for pname, P in (("fp32", Prec()), ("emu", Prec(BF))):
        t0 = time.time()
        res[pname] = OracleVAEDecoder(c["cfg"], vw, P)(z)
        print(f"vae_1024 {pname}: {time.time() - t0:.0f} s", flush=True)
    out["raw_fp32_f16"] = res["fp32"].numpy().astype(np.float16)  # decoder output before /2 + 0.5 and clip
    out["emu_rel_l2"] = np.float64(rel_l2(res["fp32"], res["emu"]))
    out["emu_psnr_image"] = np.float64(psnr(torch.clip(res["fp32"] / 2 + 0.5, 0, 1), torch.clip(res["emu"] / 2 + 0.5, 0, 1)))
    return out


def make_forward(c, name):
    cfg = c["cfg"]
    w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=c["seed_w"]).items()}
    text, pooled, lat = forward_inputs(c)
    ts = c["timesteps"]
    res = {}
    for pname, P in (("fp32", Prec()), ("emu", Prec(BF))):
        t0 = time.time()
        m = ref_model(cfg, w, P) if c.get("q2", cfg.is_flux) else OracleMMDiT(cfg, w, P)
        m.cache_modulation_params(pooled, torch.tensor(ts))
        taps = {}
        m(lat, text, ts[c["step"]], taps=taps)
        res[pname] = taps["final"]
        print(f"{name} {pname}: {time.time() - t0:.0f} s", flush=True)
    return {"final_fp32": res["fp32"].numpy(), "emu_rel_l2": np.float64(rel_l2(res["fp32"], res["emu"])),
            "emu_psnr": np.float64(psnr(res["fp32"], res["emu"])), "emu_max_abs": np.float64((res["fp32"] - res["emu"]).abs().max())}


def make_sd3_full_1024():
    c = SD3_FULL_1024
    cfg = c["cfg"]
    w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=c["seed_w"]).items()}
    text, pooled = sd3_full_inputs()
    x0, sig = sd3_full_start()
    out, last = {}, {}
    for pname, P in (("fp32", Prec()), ("emu", Prec(BF))):
        t0 = time.time()
        trace = []
        last[pname] = op.sample_euler(ref_model(cfg, w, P), x0, sig, text, pooled, c["cfg_weight"], Prec(BF), trace, t_act=Prec(torch.float16))
        print(f"sd3_full_1024 {pname}: {time.time() - t0:.0f} s", flush=True)
        if pname == "fp32":
            out["x_step3_fp32"] = last[pname].numpy()
            out["x_step1_fp32"] = trace[0].numpy()
    out["emu_rel_l2"] = np.float64(rel_l2(last["fp32"], last["emu"]))
    out["emu_psnr"] = np.float64(psnr(last["fp32"], last["emu"]))
    out["emu_max_abs"] = np.float64((last["fp32"] - last["emu"]).abs().max())
    return out


def make_flux_blocks():
    c = FLUX_BLOCKS
    cfg = c["cfg"]
    w = LazyFloat(synth_mmdit_weights(cfg, seed=c["seed_w"]))
    x, pooled = flux_blocks_inputs()
    rows = torch.tensor(c["rows"])
    S_t = c["S_t"]
    from oracle.mmdit import rope_table
    rope = rope_table(cfg, S_t, c["latent"][0] // 2, c["latent"][1] // 2)
    tkey = float(c["timesteps"][c["step"]])
    out = {"rows": rows.numpy()}
    res = {}
    for pname, P in (("fp32", Prec()), ("emu", Prec(BF))):
        m = ref_model(cfg, w, P)
        m.cache_modulation_params(pooled, torch.tensor(c["timesteps"]))
        for g in c["blocks"]:
            t0 = time.time()
            if g < cfg.depth_multimodal:
                img, txt = m._double_block(g, x[:, S_t:], x[:, :S_t], tkey, rope)
                y = torch.cat([txt, img], dim=1)
            else:
                y = m._single_block(g - cfg.depth_multimodal, x, tkey, rope)
            res[(pname, g)] = y
            print(f"flux_blocks {pname} block {g}: {time.time() - t0:.0f} s", flush=True)
    for g in c["blocks"]:
        f, e = res[("fp32", g)], res[("emu", g)]
        out[f"block{g}_rows_fp32"] = f[0, rows].numpy()
        out[f"block{g}_emu_rel_l2"] = np.float64(rel_l2(f, e))
        out[f"block{g}_emu_rel_l2_delta"] = np.float64(rel_l2(f - x, e - x))  # of what the block ADDS to the stream
        out[f"block{g}_delta_over_out"] = np.float64(float(torch.linalg.norm(f - x) / torch.linalg.norm(f)))
    return out


def make_flux_pair():
    from oracle import fp8 as o8
    c = FLUX_PAIR
    cfg = c["cfg"]
    named = synth_mmdit_weights(cfg, seed=c["seed_w"])
    plain = {k: v.float() for k, v in named.items()}
    fq = o8.fake_quant_block_weights(replace(cfg, weight_dtype="fp8_e4m3"), named)
    text, pooled, lat = forward_inputs(c)
    ts = c["timesteps"]
    out = {}
    for name, w, P, aq in (("fp32", plain, Prec(), None), ("emu", plain, Prec(BF), None), ("fq_fp32", fq, Prec(), o8.mx8_fake_quant),
                           ("fq_emu", fq, Prec(BF), o8.mx8_fake_quant)):
        t0 = time.time()
        m = OracleMMDiT(replace(cfg, weight_dtype="fp8_e4m3") if aq is not None else cfg, w, P, act_quant=aq, embed_prec=Prec(embed_dtype(cfg)))
        m.cache_modulation_params(pooled, torch.tensor(ts))
        taps = {}
        m(lat, text, ts[c["step"]], taps=taps)
        out["final_" + name] = taps["final"][:, ::c["row_stride"]].contiguous().numpy()
        print(f"flux_pair {name}: {time.time() - t0:.0f} s", flush=True)
    return out


def make_sd3_full_50():
    """configs[2] closed loop, all 50 steps (fp32 oracle; same weights, conditioning, x0 and schedule as sd3_full_1024)"""
    c = SD3_FULL_50
    cfg = c["cfg"]
    w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=c["seed_w"]).items()}
    text, pooled = sd3_full_inputs()
    x0, sig = sd3_full_start(c)
    assert len(sig) == 51 and float(sig[-1]) == 0.0
    t0 = time.time()
    trace = ProgressTrace("sd3_full_50", t0)
    last = op.sample_euler(ref_model(cfg, w, Prec()), x0, sig, text, pooled, c["cfg_weight"], Prec(BF), trace, t_act=Prec(torch.float16))
    out = {"keep": np.asarray(c["keep"])}
    for k in c["keep"][:-1]:
        out[f"x_step{k}_f16"] = trace[k - 1].numpy().astype(np.float16)
    out["x_step50_fp32"] = last.numpy()
    # the image: process_out + decode as DiffusionPipeline.decode_latents_to_image (mlx/__init__.py:576-584), fp32 oracle decoder on the
    # bf16-rounded latent (the engine's input dtype), uint8 as generate_image returns it (:525-526)
    del w
    vw = {k: v.float() for k, v in synth_vae_weights(VAEDecoderConfig(), seed=c["seed_vae"]).items()}
    z = op.process_out(last, "sd3").to(BF).float()
    raw = OracleVAEDecoder(VAEDecoderConfig(), vw, Prec())(z)
    img = torch.clip(raw / 2 + 0.5, 0, 1)
    out["image_u8"] = (img[0] * 255).numpy().astype(np.uint8)
    print(f"sd3_full_50 decode done, {time.time() - t0:.0f} s", flush=True)
    return out


def make_sd3_full_50_emu():
    """context for the closed-loop row (round 5): the bf16-EMULATING oracle (the reference's MLX rounding points) over the same 50 steps,
    measured against the fp32 trajectory of fullsize_sd3_full_50.npz -- added to that file as emu_psnr / emu_rel_l2 (final latent) and
    emu_psnr_step<k> for the kept steps (~2 h on 8 cores)"""
    c = SD3_FULL_50
    cfg = c["cfg"]
    path = os.path.join(HERE, "fullsize_sd3_full_50.npz")
    old = dict(np.load(path))
    w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=c["seed_w"]).items()}
    text, pooled = sd3_full_inputs()
    x0, sig = sd3_full_start(c)
    trace = ProgressTrace("sd3_full_50 emu", time.time())
    last = op.sample_euler(ref_model(cfg, w, Prec(BF)), x0, sig, text, pooled, c["cfg_weight"], Prec(BF), trace, t_act=Prec(torch.float16))
    for k in c["keep"]:
        ref = torch.from_numpy(old["x_step50_fp32"] if k == 50 else old[f"x_step{k}_f16"].astype(np.float32))
        old[f"emu_psnr_step{k}"] = np.float64(psnr(ref, trace[k - 1]))
    ref = torch.from_numpy(old["x_step50_fp32"])
    old["emu_psnr"] = np.float64(psnr(ref, last))
    old["emu_rel_l2"] = np.float64(rel_l2(ref, last))
    print({k: float(v) for k, v in old.items() if k.startswith("emu")}, flush=True)
    return old


def make_sd3_fp16_context():
    """Round 6: SD3's reference dtype is float16 (mlx/config.py:79), the build runs bf16 and the fixtures' "emu" rows model bf16
    rounding points.  This adds what the SAME oracle reaches with every rounding point in float16 -- the arithmetic the reference actually runs --
    against the fp32 trajectory, as context rows beside the bf16 emulation: fullsize_sd3_512.npz gets emu_fp16_{rel_l2,psnr,max_abs} (configs[0]: 24
    blocks, 4 steps, final latent), fullsize_sd3_full_late.npz d<i>_emu_fp16_{rel_l2,psnr} (configs[2] full depth, CFG 5, steps 1 / 25 / 49 / 50
    teacher-forced).  Existing arrays are written back unchanged."""
    F16 = torch.float16
    # ---- configs[0]
    c = SD3_512
    cfg = c["cfg"]
    path = os.path.join(HERE, "fullsize_sd3_512.npz")
    old = dict(np.load(path))
    w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=c["seed_w"]).items()}
    text, pooled = sd3_512_inputs()
    t0 = time.time()
    lat = op.denoise_latents(OracleMMDiT(cfg, w, Prec(F16)), text, pooled, c["steps"], 0.0, c["latent"], c["noise_seed"], c["shift"], False, Prec(F16),
                             t_act=Prec(F16))
    ref = torch.from_numpy(old["latent_fp32"])
    old["emu_fp16_rel_l2"] = np.float64(rel_l2(ref, lat))
    old["emu_fp16_psnr"] = np.float64(psnr(ref, lat))
    old["emu_fp16_max_abs"] = np.float64((ref - lat).abs().max())
    print(f"sd3_512 fp16 emulation: {time.time() - t0:.0f} s, rel-L2 {old['emu_fp16_rel_l2']:.3e}, {old['emu_fp16_psnr']:.2f} dB "
          f"(bf16 emulation: {float(old['emu_rel_l2']):.3e}, {float(old['emu_psnr']):.2f} dB)", flush=True)
    np.savez_compressed(path, **old)
    # ---- configs[2], teacher-forced steps
    c = SD3_FULL_LATE
    cfg = c["cfg"]
    path = os.path.join(HERE, "fullsize_sd3_full_late.npz")
    old = dict(np.load(path))
    w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=c["seed_w"]).items()}
    text, pooled, steps = forced_inputs(c)
    m = ref_model(cfg, w, Prec(F16))
    for i, x_i, sig2 in steps:
        t0 = time.time()
        x_next = op.sample_euler(m, x_i, sig2, text, pooled, c["cfg_weight"], Prec(F16), t_act=Prec(F16))
        d = euler_direction(x_i, x_next, sig2)
        ref = torch.from_numpy(old[f"d{i}_fp32_f16"].astype(np.float32)).double()
        old[f"d{i}_emu_fp16_rel_l2"] = np.float64(rel_l2(ref, d))
        old[f"d{i}_emu_fp16_psnr"] = np.float64(psnr(ref, d))
        print(f"sd3_full_late fp16 emulation step {i + 1}: {time.time() - t0:.0f} s, rel-L2 {old[f'd{i}_emu_fp16_rel_l2']:.3e}, {old[f'd{i}_emu_fp16_psnr']:.2f} dB "
              f"(bf16 emulation: {float(old[f'd{i}_emu_rel_l2']):.3e}, {float(old[f'd{i}_emu_psnr']):.2f} dB)", flush=True)
    np.savez_compressed(path, **old)
    return None


class ProgressTrace(list):
    def __init__(self, name, t0):
        super().__init__()
        self.name, self.t0 = name, t0

    def append(self, x):
        super().append(x)
        print(f"{self.name} step {len(self)}: {time.time() - self.t0:.0f} s, rms {float(x.pow(2).mean().sqrt()):.4f}", flush=True)


def flux_dev_10_inputs():
    c = FLUX_DEV_10
    text = randn(1, c["S_t"], c["cfg"].token_level_text_embed_dim, seed=c["text_seed"])
    pooled = randn(1, c["cfg"].pooled_text_embed_dim, seed=c["text_seed"] + 1)
    return text, pooled


def make_flux_dev_10(c=None, name="flux_dev_10"):
    """configs[3]'s shape closed loop: a complete 10-step (round 6: 50-step, FLUX_DEV_50) schedule through denoise_latents (fp32 oracle)"""
    c = c or FLUX_DEV_10
    cfg = c["cfg"]
    w = LazyFloat(synth_mmdit_weights(cfg, seed=c["seed_w"]))
    text, pooled = flux_dev_10_inputs()
    t0 = time.time()
    trace = ProgressTrace(name, t0)
    lat = op.denoise_latents(ref_model(cfg, w, Prec()), text, pooled, c["steps"], 0.0, c["latent"], c["noise_seed"], c["shift"], True, Prec(BF),
                             trace=trace)
    out = {"keep": np.asarray(c["keep"]), "latent_fp32": lat.numpy()}
    for k in c["keep"][:-1]:
        out[f"x_step{k}_f16"] = trace[k - 1].numpy().astype(np.float16)   # sample_euler's latent (before process_out)
    return out


CASES = {"sd3_fp16_context": make_sd3_fp16_context, "flux_dev_50": lambda: make_flux_dev_10(FLUX_DEV_50, "flux_dev_50"), "sd3_full_50": make_sd3_full_50, "sd3_full_50_emu": make_sd3_full_50_emu, "flux_dev_10": make_flux_dev_10, "flux_pair": make_flux_pair, "sd3_512": make_sd3_512, "vae_1024": make_vae_1024, "sd3_1024": lambda: make_forward(SD3_1024, "sd3_1024"),
         "flux_1024": lambda: make_forward(FLUX_1024, "flux_1024"), "flux_full": make_flux_full,
         "flux_full_emu": lambda: make_flux_full(True),
         "flux_dev_512": lambda: make_forward(FLUX_DEV_512, "flux_dev_512"), "sd3_full_1024": make_sd3_full_1024,
         "flux_blocks": make_flux_blocks,
         "flux_dev_full": lambda: make_forced(FLUX_DEV_FULL, "flux_dev_full", True),
         "sd3_full_late": lambda: make_forced(SD3_FULL_LATE, "sd3_full_late", True),
         "sd35_full": lambda: make_forced(SD35_FULL, "sd35_full", False),  # (fp32 oracle only: the bf16-emulating pass on 8 B parameters ran this 62 GB host out of memory)
         "flux_512": lambda: make_flux_full(True, FLUX_512, "flux_512"),
         "flux_768": lambda: make_flux_full(False, FLUX_768, "flux_768")}

if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("DK_FIXTURE_THREADS", os.cpu_count() or 8)))
    for name in sys.argv[1:] or list(CASES):
        t0 = time.time()
        res = CASES[name]()
        if res is None:  # (the case updated its fixture files itself)
            continue
        path = os.path.join(HERE, f"fullsize_{name.replace('_emu', '')}.npz")
        np.savez_compressed(path, **res)
        print(name, {k: (v.shape if getattr(v, "ndim", 0) else float(v)) for k, v in res.items()}, f"{time.time() - t0:.0f} s",
              os.path.getsize(path) // 1024, "KiB", flush=True)
