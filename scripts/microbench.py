#!/usr/bin/env python
"""Per-kernel microbenchmarks at the BASELINE.json shapes (FLUX.1-schnell 1024^2, SD3-medium 1024^2,
VAE decode).  Prints one line per kernel: ms, TFLOP/s (or GB/s) and fraction of the gfx950 peak."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionkit_amd import ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(BF)
    rows = []
    gemms = [("flux qkv img", 4096, 9216, 3072), ("flux o img", 4096, 3072, 3072), ("flux fc1 img", 4096, 12288, 3072),
             ("flux fc2 img", 4096, 3072, 12288), ("flux qkv txt", 256, 9216, 3072), ("flux single qkv", 4352, 9216, 3072),
             ("flux single fc1", 4352, 12288, 3072), ("flux single l2", 4352, 3072, 15360), ("sd3 qkv img", 8192, 4608, 1536),
             ("sd3 fc1 img", 8192, 6144, 1536), ("sd3 fc2 img", 8192, 1536, 6144), ("sd3 qkv txt", 1178, 4608, 1536),
             ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]
    if not args.only or "gemm" in args.only:
        for name, M, N, K in gemms:
            x, w, b = rnd(M, K), rnd(N, K), rnd(N)
            out = torch.empty(M, N, dtype=BF, device=dev)
            ms = timeit(lambda: ops.linear(x, w, b, out=out))
            tf = 2.0 * M * N * K / ms / 1e9
            rows.append((f"gemm {name} {M}x{N}x{K}", ms, f"{tf:8.1f} TF/s  {tf / 2500 * 100:5.1f}% of bf16 MFMA peak"))
    if not args.only or "attn" in args.only:
        for name, B, H, S, D in [("flux joint", 1, 24, 4352, 128), ("sd3 joint cfg", 2, 24, 4685, 64), ("flux b4", 4, 24, 4352, 128)]:
            qkv = rnd(B, S, 3 * H * D)
            ms = timeit(lambda: ops.attention(qkv, H, D))
            tf = 4.0 * B * H * S * S * D / ms / 1e9
            rows.append((f"attn {name} B{B} H{H} S{S} D{D}", ms, f"{tf:8.1f} TF/s  {tf / 2500 * 100:5.1f}% of bf16 MFMA peak"))
    if not args.only or "conv" in args.only:
        for name, B, H, W, Cc, O, ups in [("vae 512@128", 1, 128, 128, 512, 512, False), ("vae 512@256", 1, 256, 256, 512, 512, False),
                                          ("vae up 512@128->256", 1, 128, 128, 512, 512, True), ("vae 256@512", 1, 512, 512, 256, 256, False),
                                          ("vae 128@1024", 1, 1024, 1024, 128, 128, False), ("vae out 128->3@1024", 1, 1024, 1024, 128, 3, False)]:
            x, w, b = rnd(B, H, W, Cc), rnd(O, 3, 3, Cc), rnd(O)
            ms = timeit(lambda: ops.conv3x3(x, w, b, upsample=ups), iters=5)
            Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
            tf = 2.0 * B * Ho * Wo * 9 * Cc * O / ms / 1e9
            gb = (x.numel() + B * Ho * Wo * O) * 2 / ms / 1e6
            rows.append((f"conv {name}", ms, f"{tf:8.1f} TF/s  {gb:7.0f} GB/s algorithmic"))
    if not args.only or "elt" in args.only:
        x, sh, sc = rnd(1, 4352, 3072), rnd(1, 3072), rnd(1, 3072)
        ms = timeit(lambda: ops.ln_modulate(x, sh, sc))
        rows.append(("ln_modulate 4352x3072", ms, f"{2 * x.numel() * 2 / ms / 1e6:8.0f} GB/s ({2 * x.numel() * 2 / ms / 1e6 / 8000 * 100:.0f}% of 8 TB/s)"))
        x3, sh3, sc3 = rnd(2, 4685, 1536), rnd(2, 1536), rnd(2, 1536)
        ms = timeit(lambda: ops.ln_modulate(x3, sh3, sc3))
        rows.append(("ln_modulate 2x4685x1536 (sd3)", ms, f"{2 * x3.numel() * 2 / ms / 1e6:8.0f} GB/s"))
        qkv = rnd(1, 4352, 9216)
        qw = rnd(128)
        tab = ops.rope_table(256, 64, 64, (16, 56, 56), 10000.0, dev)
        ms = timeit(lambda: ops.qk_norm_rope_(qkv, 24, 128, qw, qw, tab))
        byt = 2 * (2 * 4352 * 3072 * 2)
        rows.append(("qk_norm_rope 4352x(2x3072)", ms, f"{byt / ms / 1e6:8.0f} GB/s"))
        xg, gm = rnd(1, 1024, 1024, 128), rnd(128)
        ms = timeit(lambda: ops.groupnorm(xg, gm, gm, 32, 1e-5, True), iters=5)
        rows.append(("groupnorm+silu 1024^2x128", ms, f"{3 * xg.numel() * 2 / ms / 1e6:8.0f} GB/s (2 reads + 1 write)"))
    for name, ms, extra in rows:
        print(f"{name:44s} {ms:9.3f} ms  {extra}")


if __name__ == "__main__":
    main()
