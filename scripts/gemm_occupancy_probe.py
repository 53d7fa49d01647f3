"""How the one-wave-per-SIMD GEMM's time depends on the number of tiles in flight (round 6; the attention kernel's twin: attn_occupancy_probe.py):
N = 3072, K = 15360 (linear2) and K = 3072 (o_proj), 256-row tiles forced, M = 256 x (tiles / 12), cold weights.  A fixed time per tile would make 60 tiles
as slow as 252; a fixed aggregate rate makes a partly filled round nearly free."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
ops.tune("gemm_v4", 2)
ops.tune("gemm_mf", 8)
ops.tune("gemm_split", 0)
N = 3072
for K in (15360, 3072):
    wl = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(8)]
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    for rt in (5, 10, 15, 17, 20, 21, 22, 32, 42, 43, 64, 85):
        M = 256 * rt
        x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        best = 1e9
        for rnd in range(3):
            for i in range(2):
                ops.linear(x, wl[i % 8], b, out=y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                ops.linear(x, wl[i % 8], b, out=y)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 12)
        tiles = rt * 12
        print(f"K {K:5d}: {tiles:4d} tiles = {tiles / 256:.2f} rounds of 256 CUs: {best * 1e3:7.1f} us, {tiles * (K // 64) / (best * 1e3):7.1f} K-tile steps/us aggregate, "
              f"{2.0 * M * N * K / best / 1e9:6.0f} TF", flush=True)
