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

from diffusionkit_amd.config import FLUX_SCHNELL, SD3_2b, VAEDecoderConfig  # noqa: E402
from diffusionkit_amd.weights import synth_mmdit_weights, synth_vae_weights  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from oracle.mmdit import OracleMMDiT, Prec  # noqa: E402
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


def make_flux_full(with_emu=False):
    c = FLUX_FULL
    cfg = c["cfg"]
    w = LazyFloat(synth_mmdit_weights(cfg, seed=c["seed_w"]))
    text, pooled = flux_full_inputs()
    out, lat = {}, {}
    for pname, P in (("fp32", Prec()),) + ((("emu", Prec(BF)),) if with_emu else ()):
        t0 = time.time()
        trace = []
        lat[pname] = op.denoise_latents(OracleMMDiT(cfg, w, P), text, pooled, c["steps"], 0.0, c["latent"], c["noise_seed"], c["shift"],
                                        True, Prec(BF), trace=trace)
        print(f"flux_full {pname}: {time.time() - t0:.0f} s", flush=True)
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
        m = OracleMMDiT(cfg, w, P)
        m.cache_modulation_params(pooled, torch.tensor(ts))
        taps = {}
        m(lat, text, ts[c["step"]], taps=taps)
        res[pname] = taps["final"]
        print(f"{name} {pname}: {time.time() - t0:.0f} s", flush=True)
    return {"final_fp32": res["fp32"].numpy(), "emu_rel_l2": np.float64(rel_l2(res["fp32"], res["emu"])),
            "emu_psnr": np.float64(psnr(res["fp32"], res["emu"])), "emu_max_abs": np.float64((res["fp32"] - res["emu"]).abs().max())}


CASES = {"sd3_512": make_sd3_512, "vae_1024": make_vae_1024, "sd3_1024": lambda: make_forward(SD3_1024, "sd3_1024"),
         "flux_1024": lambda: make_forward(FLUX_1024, "flux_1024"), "flux_full": make_flux_full,
         "flux_full_emu": lambda: make_flux_full(True)}

if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    for name in sys.argv[1:] or list(CASES):
        t0 = time.time()
        res = CASES[name]()
        path = os.path.join(HERE, f"fullsize_{name.replace('_emu', '')}.npz")
        np.savez_compressed(path, **res)
        print(name, {k: (v.shape if getattr(v, "ndim", 0) else float(v)) for k, v in res.items()}, f"{time.time() - t0:.0f} s",
              os.path.getsize(path) // 1024, "KiB", flush=True)
