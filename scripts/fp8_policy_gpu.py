"""fp8 precision policy, measured on the GPU at full depth: dB per step against ms per step for
``MMDiTConfig.fp8_bf16_double_blocks = n`` -- the first n double-stream blocks keep bf16 Linears, everything else runs e4m3 weights /
MX-fp8 activations.  Case = tests/golden/fullsize_flux_dev_full.npz (BASELINE configs[3] at full depth: 19 + 38 blocks, S_t = 512,
teacher-forced Euler steps 1 / 2 / 49 / 50 of the 50-step schedule against the fp32 oracle with the ORIGINAL weights).

    python scripts/fp8_policy_gpu.py [n ...]        (default n = 0 1 2 3 4 6 8 12 19; "bf16" = the bf16 model as the ceiling)
"""
import gc
import os
import sys
import time
from dataclasses import replace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_fullsize_fixtures as fx  # noqa: E402
from diffusionkit_amd.pipeline import FluxPipeline  # noqa: E402
from diffusionkit_amd.weights import pack_mmdit, synth_mmdit_weights  # noqa: E402
from tests import test_gpu_fullsize as tf  # noqa: E402
from tests._util import BF, psnr, rel_l2  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ns = sys.argv[1:] or ["0", "1", "2", "3", "4", "6", "8", "12", "19", "bf16"]
    c = fx.FLUX_DEV_FULL
    f = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_flux_dev_full.npz"))
    t0 = time.time()
    w = synth_mmdit_weights(c["cfg"], seed=c["seed_w"])
    print(f"weights drawn in {time.time() - t0:.0f} s", flush=True)
    text, pooled, _ = fx.forced_inputs(c)
    for n in ns:
        if n == "bf16":
            cfg = c["cfg"]
        else:
            cfg = replace(c["cfg"], weight_dtype="fp8_e4m3", fp8_bf16_double_blocks=int(n))
        pipe = FluxPipeline(w16=True, a16=True, shift=c["shift"], device=dev, text_len=c["S_t"], packed_weights={"mmdit": pack_mmdit(cfg, dict(w), dev)},
                            mmdit_config=cfg)
        got = tf.forced_steps(pipe, c, dev)
        ps, es = [], []
        for i in sorted(got):
            ref = torch.from_numpy(f[f"d{i}_fp32_f16"].astype(np.float32))
            ps.append(psnr(ref, got[i].float()))
            es.append(rel_l2(ref, got[i].float()))
        # time per step: 6 forwards of the prepared engine through the pipeline's own step loop
        lat, it = pipe.denoise_latents(text.to(dev, BF), pooled.to(dev, BF), num_steps=8, cfg_weight=0.0, latent_size=c["latent"], seed=0)
        torch.cuda.synchronize()
        t1 = time.time()
        lat, it = pipe.denoise_latents(text.to(dev, BF), pooled.to(dev, BF), num_steps=8, cfg_weight=0.0, latent_size=c["latent"], seed=0)
        torch.cuda.synchronize()
        ms = (time.time() - t1) / 8 * 1e3
        print(f"bf16 double blocks {n:>4}: Euler direction PSNR steps 1 / 2 / 49 / 50 = " + " / ".join(f"{p:.2f}" for p in ps) +
              f" dB (worst {min(ps):.2f}), rel-L2 worst {max(es):.3e}, {ms:.2f} ms per step (8-step loop incl. host)", flush=True)
        del pipe, got, lat
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
