#!/usr/bin/env python
"""Headline benchmark: images/sec for FLUX.1-schnell 1024x1024, 4 Euler steps, bf16, batch 1 per GPU
(BASELINE.json configs[1]); data-parallel over N GPUs of one node (independent images, one RCCL
weight broadcast at load, no collectives in the step loop).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...                      (spawns the N ranks itself when not started by a launcher)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Other BASELINE.json lines: --workload sd3-medium-1024 (configs[2]), --workload flux-dev-1024 --fp8 (configs[3]),
--batch 8 on 8 GPUs (configs[4]: 64 images per step over the node; --batch auto picks it when N > 1).  The default single-GPU run
appends short legs of configs[2] and configs[3] to the same JSON line as "other_configs" (--no-other-configs skips them).

A "step" is one pass of the hot path over one synthetic input per rank: the denoising steps (MMDiT
forward + fused CFG/Euler update) followed by the VAE latent decode to a uint8 image.  Inputs
(conditioning, weights) are resident in HBM before the timed region; the per-image noise draw is the
reference's host numpy RNG (mlx/__init__.py:553-557) and its 1 MiB upload is inside the region.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_FP8_TFLOPS = 5000.0   # same table: dense fp8 (block-scaled K = 128 form)


def block_flops(S, h):
    return 24.0 * S * h * h + 4.0 * S * S * h


def mmdit_step_flops(cfg, S_t, S_i, B):
    """Algorithmic FLOPs (2*MAC) of one MMDiT forward, modulation excluded."""
    h, S = cfg.hidden_size, S_t + S_i
    f = (cfg.depth_multimodal + cfg.depth_unified) * block_flops(S, h)
    f += 2.0 * S_t * cfg.token_level_text_embed_dim * h + 2.0 * S_i * cfg.patch_dim * h * 2
    return B * f


def vae_flops(vcfg, h, w):
    """conv / linear / attention FLOPs of the decoder (vae.py:386-401)."""
    boc = list(vcfg.block_out_channels)
    cm = boc[-1]
    f = 2.0 * h * w * 9 * vcfg.in_channels * cm
    res = lambda H, W, ci, co: 2.0 * H * W * 9 * (ci * co + co * co) + (2.0 * H * W * ci * co if ci != co else 0)
    f += 2 * res(h, w, cm, cm)
    T = h * w
    f += 4 * 2.0 * T * cm * cm + 4.0 * T * T * cm
    H, W, cprev = h, w, cm
    for j in reversed(range(len(boc))):
        co = boc[j]
        for r in range(vcfg.layers_per_block):
            f += res(H, W, cprev if r == 0 else co, co)
        cprev = co
        if j > 0:
            H, W = 2 * H, 2 * W
            f += 2.0 * H * W * 9 * co * co
    f += 2.0 * H * W * 9 * boc[0] * vcfg.out_channels
    return f


def cpu_sample_config(cfg):
    """The bounded CPU sample of a workload: the same model at the same sequence length with 1/f of its blocks, the double : single
    ratio kept (FLUX: 1 double + 2 single of 19 + 38, f = 19; SD3-medium: 2 of 24, f = 12; SD3.5-large: 2 of 38, f = 19) -- every
    block of a kind does the same work, so one denoising step = f x the sample (embedders and final layer, < 0.2 % of the
    sample's FLOPs, are then counted f times: the CPU figure is that much too slow, not too fast)."""
    from dataclasses import replace
    from math import gcd
    dm, du = cfg.depth_multimodal, cfg.depth_unified
    f = gcd(dm, du) if du else (dm // 2 if dm % 2 == 0 else dm)
    # (SD3 derives its width from the depth -- 64 x depth_multimodal: the sample keeps the workload's width)
    return replace(cfg, depth_multimodal=dm // f, depth_unified=du // f, hidden_size_override=cfg.hidden_size), f


def cpu_baseline(workload, threads=None):
    """CPU 'port' baseline: the oracle restatement (fp32, PyTorch CPU) timed on this box's host cores ON THE BENCH WORKLOAD ITSELF,
    bounded to 10-30 s: one forward of 1/f of the model's blocks at the workload's full sequence length and batch rows
    (``cpu_sample_config``), and one VAE decode at a quarter of the pixels (latent 64 x 64, scaled to the workload's latent by the
    decoder's algorithmic FLOPs: its convolutions are linear in the pixel count).  images/s = 1 / (steps x f x t_sample + t_decode).
    A whole FLUX.1-schnell step timed the same way (57 blocks, ~2 min) is in profiles/r04_cpu_flux_step.log
    (scripts/cpu_flux_step.py) next to its own 1/19 sample."""
    import torch
    from diffusionkit_amd.config import VAEDecoderConfig
    from diffusionkit_amd.weights import synth_mmdit_weights, synth_vae_weights
    from oracle.mmdit import OracleMMDiT, Prec, embed_dtype
    from oracle.vae import OracleVAEDecoder
    cores = os.cpu_count() or 1
    threads = threads or min(cores, 64)  # torch's CPU GEMMs stop scaling (and the elementwise ops regress) beyond a few dozen threads
    torch.set_num_threads(threads)
    cfg, latent, num_steps, S_t, rows = workload["cfg"], workload["latent"], workload["num_steps"], workload["S_t"], workload["rows"]
    scfg, f = cpu_sample_config(cfg)
    vcfg = VAEDecoderConfig()
    w = {k: v.float() for k, v in synth_mmdit_weights(scfg, seed=1).items()}
    g = torch.Generator().manual_seed(0)
    text = torch.randn(rows, S_t, cfg.token_level_text_embed_dim, generator=g)
    pooled = torch.randn(rows, cfg.pooled_text_embed_dim, generator=g)
    lat = torch.randn(rows, latent[0], latent[1], 16, generator=g)
    model = OracleMMDiT(scfg, w, Prec(), embed_prec=Prec(embed_dtype(scfg)))
    model.cache_modulation_params(pooled, torch.tensor([1000.0]))
    t0 = time.perf_counter()
    model(lat, text, 1000.0)
    t_sample = time.perf_counter() - t0
    del model, w
    S_i = (latent[0] // cfg.patch_size) * (latent[1] // cfg.patch_size)
    sample_fl = mmdit_step_flops(scfg, S_t, S_i, rows)
    vw = {k: v.float() for k, v in synth_vae_weights(vcfg, seed=2).items()}
    small = (min(latent[0], 64), min(latent[1], 64))
    z = torch.randn(1, small[0], small[1], 16, generator=g)
    t0 = time.perf_counter()
    OracleVAEDecoder(vcfg, vw, Prec())(z)
    t_dec_small = time.perf_counter() - t0
    dec_scale = vae_flops(vcfg, *latent) / vae_flops(vcfg, *small)
    t_step, t_dec = f * t_sample, dec_scale * t_dec_small
    image_s = num_steps * t_step + t_dec
    return {
        "value": 1.0 / image_s, "unit": "images/s", "cores": threads, "host_cores": cores, "kind": "port",
        "sample": f"oracle (fp32, PyTorch CPU, {threads} threads of {cores} host cores) on this workload: one forward of "
                  f"{scfg.depth_multimodal} double + {scfg.depth_unified} single blocks (1/{f} of the model) at {rows} x ({S_t} + {S_i}) tokens, "
                  f"width {cfg.hidden_size} = {sample_fl / 1e12:.2f} TFLOP in {t_sample:.1f} s ({sample_fl / t_sample / 1e9:.0f} GFLOP/s) -> {t_step:.0f} s per step; "
                  f"VAE decode at latent {small[0]}x{small[1]} in {t_dec_small:.1f} s, x {dec_scale:.2f} by decoder FLOPs -> {t_dec:.0f} s; "
                  f"image = {num_steps} steps + decode = {image_s:.0f} s",
        "sample_seconds": round(t_sample + t_dec_small, 2),
        "cpu_s_per_step": round(t_step, 1), "cpu_s_per_image": round(image_s, 1),
    }


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL rendezvous on
    127.0.0.1), relay rank 0's JSON line, and -- if any rank fails -- the tail of that rank's stderr."""
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        errs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    for r, (rc, f) in enumerate(zip(rcs, errs)):
        f.seek(0)
        tail = f.read().decode(errors="replace")
        if rc != 0:
            sys.stderr.write(f"---- rank {r} exited with {rc}; stderr tail ----\n{tail[-4000:]}\n")
        elif r == 0 and tail:
            sys.stderr.write(tail[-2000:])
    sys.exit(max(abs(rc) for rc in rcs))


WORKLOAD_NAMES = {"flux-schnell-1024": "FLUX.1-schnell 1024x1024 4-step", "flux-dev-1024": "FLUX.1-dev 1024x1024 50-step",
                  "sd3-medium-1024": "SD3-medium 1024x1024 50-step CFG 5.0", "sd35-large-1024": "SD3.5-large 1024x1024 50-step CFG 5.0",
                  "flux-schnell-512": "FLUX.1-schnell 512x512 4-step (the reference CLI's default resolution)",
                  "sd3-medium-512": "SD3-medium 512x512 50-step CFG 5.0 (the reference CLI's default resolution)",
                  "tiny": "tiny"}


def run_workload(ctx, workload, fp8, B, steps, warmup, guidance_embed=False, overlap_decode=False, want_roofline=True, max_replay=4, fp8_policy="quality", res=None):
    """One bench leg: build the pipeline of `workload` on this rank, `warmup` untimed images, EXACTLY `steps` timed images
    (each = the denoising steps + the VAE decode) bracketed by barrier + synchronize on both sides, max over ranks; then the
    per-launch HIP-event replay for the roofline figures (rank 0).  Returns a dict of measurements (rank 0) or None."""
    from dataclasses import replace

    import numpy as np
    import torch
    import torch.distributed as dist
    from diffusionkit_amd import _lib
    from diffusionkit_amd import dist as dk
    from diffusionkit_amd.config import FLUX_DEV, FLUX_SCHNELL, SD3_2b, SD3_8b, VAEDecoderConfig, tiny_flux, tiny_vae
    from diffusionkit_amd.pipeline import DiffusionPipeline, FluxPipeline
    from diffusionkit_amd.weights import pack_mmdit, pack_vae, synth_mmdit_weights, synth_vae_weights

    rank, world, dev, lib = ctx["rank"], ctx["world"], ctx["dev"], ctx["lib"]
    if workload == "flux-schnell-1024":
        cfg, vcfg, cls, mv = FLUX_SCHNELL, VAEDecoderConfig(), FluxPipeline, "argmaxinc/mlx-FLUX.1-schnell"
        latent, num_steps, cfg_weight, shift, S_t, rows = (128, 128), 4, 0.0, 1.0, 256, 1
    elif workload == "flux-schnell-512":
        # the resolution the reference's CLI defaults to (mlx/scripts/generate_images.py:15-30): latent 64 x 64 = 1024 image tokens, S = 1280
        cfg, vcfg, cls, mv = FLUX_SCHNELL, VAEDecoderConfig(), FluxPipeline, "argmaxinc/mlx-FLUX.1-schnell"
        latent, num_steps, cfg_weight, shift, S_t, rows = (64, 64), 4, 0.0, 1.0, 256, 1
    elif workload == "sd3-medium-512":
        cfg, vcfg, cls, mv = SD3_2b, VAEDecoderConfig(), DiffusionPipeline, "argmaxinc/mlx-stable-diffusion-3-medium"
        latent, num_steps, cfg_weight, shift, S_t, rows = (64, 64), 50, 5.0, 3.0, 589, 2
    elif workload == "flux-dev-1024":
        # BASELINE configs[3] shape: 50 steps, 512 text tokens; like the reference (quirk Q7) FLUX.1-dev runs without its guidance
        # embedding unless --guidance-embed
        cfg, vcfg, cls, mv = (FLUX_DEV if guidance_embed else FLUX_SCHNELL), VAEDecoderConfig(), FluxPipeline, "argmaxinc/mlx-FLUX.1-dev"
        latent, num_steps, cfg_weight, shift, S_t, rows = (128, 128), 50, 0.0, 1.0, 512, 1
    elif workload == "sd3-medium-1024":
        cfg, vcfg, cls, mv = SD3_2b, VAEDecoderConfig(), DiffusionPipeline, "argmaxinc/mlx-stable-diffusion-3-medium"
        latent, num_steps, cfg_weight, shift, S_t, rows = (128, 128), 50, 5.0, 3.0, 589, 2
    elif workload == "sd35-large-1024":
        # the reference's third model family (mlx/config.py:72-74, mlx/__init__.py:39): 38 blocks, 38 heads of 64 = h 2432 (9.5 column
        # tiles of 256: the 256^2 GEMM's half-tile path), QK-norm; the CLI's SD3 settings (scripts/generate_images.py:15-40)
        cfg, vcfg, cls, mv = SD3_8b, VAEDecoderConfig(), DiffusionPipeline, "argmaxinc/mlx-stable-diffusion-3.5-large"
        latent, num_steps, cfg_weight, shift, S_t, rows = (128, 128), 50, 5.0, 3.0, 589, 2
    else:
        cfg, vcfg, cls, mv = tiny_flux(), tiny_vae(), FluxPipeline, "argmaxinc/mlx-FLUX.1-schnell"
        latent, num_steps, cfg_weight, shift, S_t, rows = (16, 16), 4, 0.0, 1.0, 64, 1
    if res:  # lab sweeps (scripts/res_sweep.sh): the same model at another square resolution, latent = pixels / 8
        latent = (res // 8, res // 8)
    if fp8:
        assert cfg.is_flux and workload != "tiny", "--fp8 is offered for the FLUX workloads (head_dim 128, token counts multiples of 128)"
        from diffusionkit_amd.config import fp8_config
        cfg = fp8_config(cfg, fp8_policy)  # "quality": the first 12 double blocks keep bf16 Linears (>= 35 dB per step); "speed": every block fp8
    assert B >= 1 and (B == 1 or rows == 1), "--batch > 1 is offered for the FLUX workloads (one conditioning row per image)"

    # ---- weights: rank 0 creates them, one RCCL broadcast of the packed blob over xGMI ----
    t0 = time.perf_counter()
    packed = None
    if rank == 0:
        mm = pack_mmdit(cfg, synth_mmdit_weights(cfg, seed=1234, device=dev), dev, consume=True)
        vv = pack_vae(vcfg, synth_vae_weights(vcfg, seed=1235, device=dev), dev)
        packed = {"mmdit/" + k: v for k, v in mm.items()}
        packed.update({"vae/" + k: v for k, v in vv.items()})
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    t0 = time.perf_counter()
    packed = dk.broadcast_weights(packed, dev, src=0)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    blob_bytes = sum(v.numel() * v.element_size() for v in packed.values())
    weights = {"mmdit": {k[6:]: v for k, v in packed.items() if k.startswith("mmdit/")},
               "vae_decoder": {k[4:]: v for k, v in packed.items() if k.startswith("vae/")}}
    pipe = cls(w16=True, a16=True, shift=shift, model_version=mv, mmdit_config=cfg, vae_config=vcfg, device=dev,
               text_len=S_t, packed_weights=weights)

    # synthetic conditioning of the reference's shapes, resident in HBM (text encoders are out of scope)
    g = torch.Generator().manual_seed(1 + rank)
    cond = torch.randn(rows, S_t, cfg.token_level_text_embed_dim, generator=g).to(dev, torch.bfloat16)
    pooled = torch.randn(rows, cfg.pooled_text_embed_dim, generator=g).to(dev, torch.bfloat16)

    denoise_ms, vae_ms = [], []
    pending = []

    def one_image(seed):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        lat, _ = pipe.denoise_latents(cond, pooled, num_steps=num_steps, cfg_weight=cfg_weight, latent_size=latent,
                                      seed=seed if B == 1 else [seed * B + b for b in range(B)])
        e1.record()
        if overlap_decode:
            pending.append(pipe.decode_async(lat))  # waited for in barrier(): every image is decoded inside the timed region
            u8 = None
        else:
            _, u8, _ = pipe.decoder.decode(lat)
        e2.record()
        return u8, (e0, e1, e2)

    def barrier():
        while pending:
            pending.pop().result()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        one_image(1000 + i)
    barrier()
    t0 = time.perf_counter()
    evs = []
    for i in range(steps):
        u8, ev = one_image(rank * 100000 + i)
        evs.append(ev)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0  # this rank's own time (reported per rank; `elapsed` below is the contract's figure)
    barrier()
    elapsed = time.perf_counter() - t0
    for e0, e1, e2 in evs:
        denoise_ms.append(e0.elapsed_time(e1))
        vae_ms.append(e1.elapsed_time(e2))
    per_rank = [steps * B / own]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([steps * B / own], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(t.item()) for t in allr]

    # ---- roofline of the dominant kernel (the block Linears' MFMA GEMM), HIP events on the launch stream ----
    S_i = (latent[0] // cfg.patch_size) * (latent[1] // cfg.patch_size)
    step_flops = mmdit_step_flops(cfg, S_t, S_i, rows)  # per image
    image_flops = num_steps * step_flops + vae_flops(vcfg, *latent)
    roofline = None
    if want_roofline and rank == 0:
        # the replay is bounded (at most 4 images: a 50-step image alone is ~10 000 launches); every launch of it is recorded
        # (the event pool grows on demand and dk_profile_read fails if a launch was dropped)
        n_replay = min(steps, max_replay)
        lib.dk_profile_enable(1)
        t1 = time.perf_counter()
        for i in range(n_replay):
            one_image(rank * 100000 + i)
        torch.cuda.synchronize()
        instr = time.perf_counter() - t1
        stats = {}
        for name, cls_id in (("gemm", 0), ("conv", 1), ("attention", 2), ("gemm_fp8", 3)):
            ms, work, n = C.c_double(), C.c_double(), C.c_int64()
            _lib.check(lib.dk_profile_read(cls_id, C.byref(ms), C.byref(work), C.byref(n)), "dk_profile_read")
            stats[name] = (ms.value, work.value, n.value)
        lib.dk_profile_enable(0)
        # same-box reference for the round-5 kernel choice: the same replay with generation 3 (gemm256v3.hip) on every block Linear -- boxes of
        # the pool differ by several per cent, this pair of numbers does not (bf16 headline workload only: one more second)
        gen3 = None
        if not fp8 and workload == "flux-schnell-1024" and B == 1:
            _lib.check(lib.dk_tune_set(b"gemm_v4", 0), "dk_tune_set")
            try:
                one_image(rank * 100000)  # warm
                lib.dk_profile_enable(1)
                for i in range(n_replay):
                    one_image(rank * 100000 + i)
                torch.cuda.synchronize()
                ms3, work3, n3 = C.c_double(), C.c_double(), C.c_int64()
                _lib.check(lib.dk_profile_read(0, C.byref(ms3), C.byref(work3), C.byref(n3)), "dk_profile_read")
                gen3 = {"achieved": round(work3.value / max(ms3.value, 1e-9) / 1e9, 1), "gemm_ms_per_image": round(ms3.value / n_replay / B, 2),
                        "what": "the same replay with dk_tune_set(gemm_v4, 0): gemm256v3.hip on every block Linear (round 4's kernel choice)"}
            finally:
                lib.dk_profile_enable(0)
                _lib.check(lib.dk_tune_set(b"gemm_v4", -1), "dk_tune_set")
            # ... and the shipped choice once more behind it (order check: a box that warms up or throttles during the replays shows here)
            one_image(rank * 100000)
            lib.dk_profile_enable(1)
            for i in range(n_replay):
                one_image(rank * 100000 + i)
            torch.cuda.synchronize()
            _lib.check(lib.dk_profile_read(0, C.byref(ms3), C.byref(work3), C.byref(n3)), "dk_profile_read")
            lib.dk_profile_enable(0)
            gen3["shipped_choice_replayed_again"] = round(work3.value / max(ms3.value, 1e-9) / 1e9, 1)
        dom = "gemm_fp8" if fp8 else "gemm"
        peak = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS
        ms, work, n = stats[dom]
        ach = work / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic, traffic_src = None, None
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the figure comes
        # from the committed rocprofv3 --pmc passes over this same command (default workload, one image per step) and is attached
        # to that configuration only
        pmc = os.path.join(ROOT, "profiles", "pmc_gemm_fp8_traffic.json" if fp8 else "pmc_gemm_traffic.json")
        if os.path.exists(pmc) and workload == "flux-schnell-1024" and B == 1:
            pj = json.load(open(pmc))
            traffic, traffic_src = pj.get("hbm_bytes_per_launch"), pj.get("source", "profiles/pmc_gemm_traffic.json")

        def sub(name, pk=PEAK_BF16_TFLOPS):
            t_ms, wk, cnt = stats[name]
            a = wk / max(t_ms, 1e-9) / 1e9
            return {"achieved": round(a, 1), "frac": round(a / pk, 4), "ms_per_image": round(t_ms / n_replay / B, 2), "launches": cnt}

        roofline = {
            "bound": "mfma",
            "kernel": ("dk_gemm256f8_kernel (e4m3 x e4m3 block-scaled 16x16x128-MFMA GEMM)" if fp8 else
                       "dk_gemm256v4_kernel<8 | 7> + dk_gemm256v3_kernel (bf16 16x16x32-MFMA GEMMs of the block Linears: one wave per SIMD with a generated asm body, "
                       "256- or 224-row tiles, for K >= 2048 launches of segment-uniform tiles -- every block Linear of FLUX; the 8-wave kernel for the others: "
                       "SD3's K = 1536, SD3.5-large's half column tiles, ragged text rows; small-M shapes: dk_gemm_bf16_kernel<0>)"),
            "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
            "traffic_source": traffic_src,
            "launches": n, "avg_launch_us": round(ms * 1e3 / max(n, 1), 2),
            "flops_per_launch": work / max(n, 1),
            "gemm_ms_per_image": round(ms / n_replay / B, 2),
            "attention": sub("attention"), "conv": sub("conv"),
            "replayed_images": n_replay * B,
            "instrumented_ms_per_step": round(instr / n_replay * 1e3, 2),
            "method": "HIP events around every launch on the launch stream, replay of the timed region",
        }
        if fp8:
            roofline["bf16_gemm"] = sub("gemm")
        if gen3 is not None:
            gen3["frac"] = round(gen3["achieved"] / peak, 4)
            roofline["same_box_generation3_only"] = gen3

    res = None
    if rank == 0:
        peak_whole = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS
        res = {
            "value": round(world * steps * B / elapsed, 4), "ms_per_step": round(elapsed / steps * 1e3, 2),
            "dtype": "fp8 e4m3 weights x MX-fp8 activations (fp32 accumulate), bf16 elsewhere" if fp8 else "bf16",
            "workload": f"{workload}: latent {latent[0]}x{latent[1]}, {num_steps} Euler steps, cfg_weight {cfg_weight}, "
                        f"text tokens {S_t}, batch {B} image{'s' if B > 1 else ''} per GPU per step, + VAE decode to uint8"
                        + (", guidance embedding on" if cfg.guidance_embed else "")
                        + (", decode of image i overlapped with the denoising of image i+1 (side stream)" if overlap_decode else ""),
            "denoise_ms_per_step": round(float(np.mean(denoise_ms)) / num_steps, 2),
            "vae_decode_ms": round(float(np.mean(vae_ms)), 2),
            "algorithmic_tflop_per_image": round(image_flops / 1e12, 2),
            "mfma_roofline_frac_whole_path": round(image_flops * steps * B / elapsed / (peak_whole * 1e12), 4),
            "weight_init_s": round(t_init, 2), "weight_bcast_s": round(t_bcast, 3), "weight_blob_gb": round(blob_bytes / 1e9, 2),
            "per_rank_images_per_s": [round(v, 4) for v in per_rank],
            "roofline": roofline, "image_flops": image_flops,
            "cpu_workload": {"cfg": replace(cfg, weight_dtype="bf16") if fp8 else cfg, "latent": latent, "num_steps": num_steps, "S_t": S_t, "rows": rows},
        }
    del pipe, weights, packed
    torch.cuda.empty_cache()
    return res


def dry_run(args):
    """--dry-run: see the flag's help.  Same statements as main() / run_workload() wherever they do not touch the device."""
    import torch
    import torch.distributed as dist
    from diffusionkit_amd import dist as dk
    from diffusionkit_amd.config import tiny_flux, tiny_vae
    from diffusionkit_amd.weights import pack_mmdit, pack_vae, synth_mmdit_weights, synth_vae_weights
    rank, local_rank, world = dk.init_distributed("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    B = (8 if world > 1 else 1) if args.batch == "auto" else int(args.batch)
    cfg, vcfg = tiny_flux(), tiny_vae()
    packed = None
    if rank == 0:
        packed = {"mmdit/" + k: v for k, v in pack_mmdit(cfg, synth_mmdit_weights(cfg, seed=1234), "cpu").items()}
        packed.update({"vae/" + k: v for k, v in pack_vae(vcfg, synth_vae_weights(vcfg, seed=1235), "cpu").items()})
    t0 = time.perf_counter()
    packed = dk.broadcast_weights(packed, "cpu", src=0)
    t_bcast = time.perf_counter() - t0
    blob_bytes = sum(v.numel() * v.element_size() for v in packed.values())
    seeds = dk.shard_seeds(list(range(world * B)), rank, world)  # configs[4]: rank r denoises seeds 8r .. 8r + 7 in one batched step loop

    def barrier():
        if world > 1:
            dist.barrier()

    def one_image(i):
        time.sleep(0.01 * (1 + rank))  # the slowest rank sets the time: MAX over ranks below

    for i in range(args.warmup):
        one_image(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_image(i)
    own = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [args.steps * B / own]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([args.steps * B / own], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(t.item()) for t in allr]
    if rank == 0:
        print(json.dumps({
            "metric": f"images/sec (whole node) {WORKLOAD_NAMES[args.workload]}", "dry_run": True,
            "value": round(world * args.steps * B / elapsed, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "none (dry run: device work replaced by a sleep)", "data": "none",
            "config": {"workload": f"dry run of the control path: batch {B} images per rank per step, seeds of rank 0: {seeds}",
                       "parallelism": f"dp{world} (independent images, weight broadcast at load)"},
            "world": world, "collective_backend": "gloo" if world > 1 else None, "weight_bcast_s": round(t_bcast, 3),
            "weight_blob_gb": round(blob_bytes / 1e9, 6), "per_rank_images_per_s": [round(v, 4) for v in per_rank],
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed images per rank")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="flux-schnell-1024", choices=["flux-schnell-1024", "sd3-medium-1024", "sd35-large-1024", "flux-dev-1024", "flux-schnell-512", "sd3-medium-512", "tiny"])
    ap.add_argument("--batch", default="1",
                    help="images per rank and step, denoised in one batched step loop (FLUX workloads).  Default 1 = BASELINE configs[1] "
                         "per GPU.  BASELINE configs[4] (FLUX.1-schnell, batch 64 sharded over 8 GPUs) is `--gpus 8 --batch 8`; "
                         "`--batch auto` picks that per-GPU shape (8) whenever --gpus > 1 and 1 on a single GPU")
    ap.add_argument("--fp8", action="store_true", help="fp8 (e4m3) weights + MX-fp8 activations on the block-scaled fp8 MFMA (BASELINE configs[3])")
    ap.add_argument("--fp8-policy", default="quality", choices=["quality", "speed"],
                    help="quality (default): the first 12 double-stream blocks keep bf16 Linears, >= 35 dB per step against the fp32 oracle; "
                         "speed: every block Linear in fp8 (32 dB per step)")
    ap.add_argument("--guidance-embed", action="store_true",
                    help="FLUX.1-dev guidance embedding on (config FLUX_DEV); off = the reference's behaviour, which runs dev on the schnell preset")
    ap.add_argument("--overlap-decode", action="store_true",
                    help="decode image i on a side stream while image i+1 is denoised (DiffusionPipeline.decode_async); the default, and the "
                         "headline, is the reference's order: denoise, then decode, image by image")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs of BASELINE configs[2] (SD3-medium 1024x1024, 50 steps, CFG 5) and configs[3] (FLUX.1-dev, 50 steps, "
                         "fp8 weights) that the default single-GPU headline run appends as `other_configs`")
    ap.add_argument("--res", type=int, default=None, metavar="PIXELS",
                    help="lab: run the chosen workload's model at PIXELS x PIXELS instead of its own resolution (a multiple of 16; dispatch sweeps, profiles/r06_res_sweep.md)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="dk_tune_set knob for A/B runs, e.g. --tune gemm_mf=8 (default kernels otherwise)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the multi-rank CONTROL path of this script on the host through gloo (rank spawn, rendezvous on 127.0.0.1, "
                         "--batch auto, packed-weight broadcast of the tiny model, barriers, MAX-over-ranks timing, per-rank gather, the JSON line) "
                         "with the device work replaced by a sleep -- what tests/test_dist_cpu.py runs so that the first 8-GPU launch is not the "
                         "first time this code runs.  The line carries \"dry_run\": true and no roofline.")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist
    from diffusionkit_amd import _lib
    from diffusionkit_amd import dist as dk

    rank, local_rank, world = dk.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU path"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(lib.dk_tune_set(k.encode(), int(v)), "dk_tune_set")
    B = (8 if world > 1 else 1) if args.batch == "auto" else int(args.batch)
    ctx = {"rank": rank, "world": world, "dev": dev, "lib": lib}

    head = run_workload(ctx, args.workload, args.fp8, B, args.steps, args.warmup, guidance_embed=args.guidance_embed,
                        overlap_decode=args.overlap_decode, want_roofline=not args.no_roofline, fp8_policy=args.fp8_policy, res=args.res)

    # ---- the other north-star configurations on the same driver-timed line (single-GPU headline runs only): short bounded legs,
    # same code path and timing contract as the headline (warm-up, barrier + synchronize brackets, decode inside the region) ----
    other = None
    if world == 1 and args.workload == "flux-schnell-1024" and not args.fp8 and B == 1 and not args.no_other_configs and not args.tune and not args.res:
        other = {}
        for key, (wl, fp8, n_img, b_leg) in {
                "sd3-medium-1024 (BASELINE configs[2])": ("sd3-medium-1024", False, 2, 1),
                "flux-dev-1024 fp8 (BASELINE configs[3]; precision policy: first 12 double blocks bf16, >= 35 dB per step)": ("flux-dev-1024", "quality", 2, 1),
                "flux-dev-1024 fp8, every block Linear in fp8 (32 dB per step)": ("flux-dev-1024", "speed", 1, 1),
                "sd35-large-1024 (the reference's third model family, mlx/config.py:72-74)": ("sd35-large-1024", False, 1, 1),
                # round 6: the resolution the reference's CLI defaults to, and the per-GPU shape between configs[1] and configs[4]
                "flux-schnell-512 (the reference CLI's default resolution, generate_images.py:15-30)": ("flux-schnell-512", False, 8, 1),
                "sd3-medium-512 (the reference CLI's default resolution)": ("sd3-medium-512", False, 2, 1),
                "flux-schnell-1024 batch 4 (four images per step loop on one GPU)": ("flux-schnell-1024", False, 2, 4)}.items():
            r = run_workload(ctx, wl, bool(fp8), b_leg, n_img, 1, want_roofline=not args.no_roofline, max_replay=1, fp8_policy=fp8 or "quality")
            rf = r["roofline"] or {}
            other[key] = {"metric": f"images/sec {WORKLOAD_NAMES[wl]}", "value": r["value"], "unit": "images/s", "steps": n_img, "warmup": 1,
                          "ms_per_step": r["ms_per_step"], "denoise_ms_per_step": r["denoise_ms_per_step"], "vae_decode_ms": r["vae_decode_ms"],
                          "dtype": r["dtype"], "workload": r["workload"], "algorithmic_tflop_per_image": r["algorithmic_tflop_per_image"],
                          "mfma_roofline_frac_whole_path": r["mfma_roofline_frac_whole_path"],
                          "roofline": {k: rf.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "gemm_ms_per_image", "attention", "conv", "bf16_gemm") if k in rf}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload != "tiny":
        cpu = cpu_baseline(head["cpu_workload"])

    if rank == 0:
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001  (version query only)
            rccl = None
        out = {
            "metric": f"images/sec (whole node) {WORKLOAD_NAMES[args.workload]}",
            "value": head["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": head["dtype"],
            "data": "synthetic (seeded random weights and conditioning; reference numpy noise draw)",
            "config": {"workload": head["workload"], "parallelism": f"dp{world} (independent images, weight broadcast at load)"},
            "denoise_ms_per_step": head["denoise_ms_per_step"], "vae_decode_ms": head["vae_decode_ms"],
            "algorithmic_tflop_per_image": head["algorithmic_tflop_per_image"],
            "mfma_roofline_frac_whole_path": head["mfma_roofline_frac_whole_path"],
            "weight_init_s": head["weight_init_s"], "weight_bcast_s": head["weight_bcast_s"], "weight_blob_gb": head["weight_blob_gb"],
            "world": world, "rccl_version": rccl, "collective_backend": "nccl (RCCL)" if world > 1 else None,
            "per_rank_images_per_s": head["per_rank_images_per_s"],
            "roofline": head["roofline"], "cpu_baseline": cpu, "other_configs": other,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
