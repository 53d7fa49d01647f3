"""Regenerates tests/golden/oracle_tiny.npz from the CPU oracle.

The reference (MLX) cannot be imported in this environment, so these are NOT reference outputs:
they are self-consistency goldens that pin the oracle restatement against regressions and give
the `-m "not gpu"` suite something concrete to check (PARITY UNPINNED, see oracle/mmdit.py).
Run from the repo root:  python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from diffusionkit_amd.config import tiny_flux, tiny_sd3, tiny_vae  # noqa: E402
from diffusionkit_amd.weights import synth_mmdit_weights, synth_vae_weights  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from oracle.mmdit import OracleMMDiT, Prec  # noqa: E402
from oracle.vae import OracleVAEDecoder, decode_latents_to_image  # noqa: E402


def case_inputs(cfg, batch, text_len=16):
    g = torch.Generator().manual_seed(7)
    text = torch.randn(batch, text_len, cfg.token_level_text_embed_dim, generator=g).to(torch.bfloat16).float()
    pooled = torch.randn(batch, cfg.pooled_text_embed_dim, generator=g).to(torch.bfloat16).float()
    return text, pooled


def run():
    torch.set_num_threads(4)
    out = {}
    for name, cfg, shift, cfgw in (("flux", tiny_flux(), 1.0, 0.0), ("sd3", tiny_sd3(), 3.0, 5.0)):
        w = {k: v.float() for k, v in synth_mmdit_weights(cfg, seed=1234).items()}
        text, pooled = case_inputs(cfg, 2 if cfgw > 0 else 1)
        for pname, P in (("fp32", Prec()), ("bf16", Prec(torch.bfloat16))):
            m = OracleMMDiT(cfg, w, P)
            lat = op.denoise_latents(m, text, pooled, 3, cfgw, (8, 8), 0, shift, cfg.is_flux, Prec(torch.bfloat16))
            out[f"{name}_{pname}_latent"] = lat.numpy()
    vc = tiny_vae()
    vw = {k: v.float() for k, v in synth_vae_weights(vc, seed=4321).items()}
    g = torch.Generator().manual_seed(11)
    z = torch.randn(1, 8, 8, 16, generator=g)
    for pname, P in (("fp32", Prec()), ("bf16", Prec(torch.bfloat16))):
        img = decode_latents_to_image(OracleVAEDecoder(vc, vw, P), z)
        out[f"vae_{pname}_image"] = img[:, ::4, ::4].numpy()  # subsampled to keep the fixture small
    return out


if __name__ == "__main__":
    res = run()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_tiny.npz")
    np.savez_compressed(path, **res)
    for k, v in res.items():
        print(k, v.shape, float(np.abs(v).mean()))
