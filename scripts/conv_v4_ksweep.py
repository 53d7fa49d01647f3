"""conv256v4: time against the reduction length at a fixed output (512 x 512 pixels x 256 channels = 1024 tiles = 4 rounds of the CUs): the slope is
the K loop (us per 64-channel chunk = 9 K-tiles), the intercept the fixed cost per tile (prologue with the first halo, drain, tail)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
BF = torch.bfloat16
H = W = 512
O = 256
for mode in (2, 0):
    ops.tune("conv_v4", mode)
    for gn in (False, True):
        pts = []
        for C in (128, 256, 512, 1024):
            x = (torch.randn(1, H, W, C, device=dev, generator=g)).to(BF)
            w = (torch.randn(O, 9 * C, device=dev, generator=g) * 0.02).to(BF)
            b = torch.zeros(O, device=dev, dtype=BF)
            tab = ops.groupnorm_table(x, torch.ones(C, device=dev, dtype=BF), torch.zeros(C, device=dev, dtype=BF), 32, 1e-5) if gn else None
            best = 1e9
            for rnd in range(3):
                ops.conv3x3_gn(x, w, b, gn_table=tab)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv3x3_gn(x, w, b, gn_table=tab)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            pts.append((C // 64, best * 1e3))
        slope = (pts[-1][1] - pts[0][1]) / (pts[-1][0] - pts[0][0])
        icpt = pts[0][1] - slope * pts[0][0]
        rounds = 4 if mode == 2 else 8
        print(f"{'conv256v4' if mode == 2 else 'conv_halo'} {'gn+silu' if gn else 'plain  '}: " + "  ".join(f"C={64 * n}: {t:7.1f} us ({2.0 * H * W * 9 * 64 * n * O / t / 1e6:5.0f} TF)" for n, t in pts) +
              f" | per chunk and round {slope / rounds:6.3f} us = {slope / rounds / 9 * 1e3:5.0f} ns per K-tile, fixed per round {icpt / rounds:5.2f} us", flush=True)
ops.tune("conv_v4", 1)
