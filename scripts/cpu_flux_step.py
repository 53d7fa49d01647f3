#!/usr/bin/env python
"""CPU baseline at full size: ONE FLUX.1-schnell denoising step of the oracle
restatement -- all 19 + 38 blocks, S = 256 + 4096 tokens, width 3072, fp32, PyTorch CPU -- timed on this box's host cores, next to
the 1/19 sample `bench.py`'s cpu_baseline takes of the same step (1 double + 2 single blocks), so that the x 19 the bench line
applies is a measured statement.  Test infrastructure: the oracle is the checker, never the product path.

    python scripts/cpu_flux_step.py [threads]      ->  profiles/r04_cpu_flux_step.log (copy of stdout)
Run it ALONE on the box (nothing else on the host cores or the GPU): round 4's first run shared the host with pytest's weight
threads and measured 250 s against 166 s extrapolated.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from diffusionkit_amd.config import FLUX_SCHNELL  # noqa: E402
from diffusionkit_amd.weights import synth_mmdit_weights  # noqa: E402
from oracle.mmdit import OracleMMDiT, Prec, embed_dtype  # noqa: E402


def aliased_full_weights(cfg, scfg, f):
    """the weight dict of the full model whose block i shares the tensors of block i % (sample depth) of the sample's weight set:
    the arithmetic (and its time) is that of the full model, the 12 B-parameter draw (100 s, 48 GB in fp32) is not needed -- a
    step's duration does not depend on the weight values"""
    import re
    w = {k: v.float() for k, v in synth_mmdit_weights(scfg, seed=1).items()}
    full = dict(w)
    for kind, depth, sdepth in (("multimodal_transformer_blocks", cfg.depth_multimodal, scfg.depth_multimodal),
                                ("unified_transformer_blocks", cfg.depth_unified, scfg.depth_unified)):
        for i in range(sdepth, depth):
            for k, v in w.items():
                m = re.match(rf"{kind}\.(\d+)\.(.*)", k)
                if m and int(m.group(1)) == i % sdepth:
                    full[f"{kind}.{i}.{m.group(2)}"] = v
    return full


def main():
    cores = os.cpu_count() or 1
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(cores, 64)
    torch.set_num_threads(threads)
    cfg, latent, S_t = FLUX_SCHNELL, (128, 128), 256
    S_i = (latent[0] // 2) * (latent[1] // 2)
    print(f"host: {cores} cores, torch threads {threads}; FLUX.1-schnell step: 19 + 38 blocks, {S_t} + {S_i} tokens, width {cfg.hidden_size}", flush=True)
    wl = {"cfg": cfg, "latent": latent, "num_steps": 4, "S_t": S_t, "rows": 1}
    t0 = time.perf_counter()
    cb = bench.cpu_baseline(wl, threads=threads)
    print(f"bench.py cpu_baseline (sample, {time.perf_counter() - t0:.0f} s incl. weight draw): {cb['value']:.5f} images/s, {cb['cpu_s_per_step']} s per step", flush=True)
    print("  " + cb["sample"], flush=True)
    scfg, f = bench.cpu_sample_config(cfg)
    w = aliased_full_weights(cfg, scfg, f)
    g = torch.Generator().manual_seed(0)
    text = torch.randn(1, S_t, cfg.token_level_text_embed_dim, generator=g)
    pooled = torch.randn(1, cfg.pooled_text_embed_dim, generator=g)
    lat = torch.randn(1, latent[0], latent[1], 16, generator=g)
    model = OracleMMDiT(cfg, w, Prec(), embed_prec=Prec(embed_dtype(cfg)))
    model.cache_modulation_params(pooled, torch.tensor([1000.0]))
    t0 = time.perf_counter()
    model(lat, text, 1000.0)
    t_step = time.perf_counter() - t0
    fl = bench.mmdit_step_flops(cfg, S_t, S_i, 1)
    print(f"ONE FULL STEP (57 blocks; block i runs on the tensors of sample block i mod 1 / 2: same arithmetic, no 12 B-parameter draw): {t_step:.1f} s = {fl / 1e12:.2f} TFLOP at {fl / t_step / 1e9:.0f} GFLOP/s", flush=True)
    print(f"sample x 19 = {cb['cpu_s_per_step']} s  ->  ratio full / extrapolated = {t_step / cb['cpu_s_per_step']:.3f}", flush=True)
    img_s = 4 * t_step + (cb["cpu_s_per_image"] - 4 * cb["cpu_s_per_step"])
    print(f"image (4 measured-rate steps + the sample's decode time): {img_s:.0f} s = {1.0 / img_s:.5f} images/s on {threads} threads of {cores} cores", flush=True)


if __name__ == "__main__":
    main()
