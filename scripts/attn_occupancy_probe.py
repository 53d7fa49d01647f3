"""How the one-wave-per-SIMD attention kernel's time depends on the number of query blocks in flight (round 6): S = 4352 (17 blocks of 256 queries per
head, 68 key tiles per block), H heads -> 17 H workgroups on 256 CUs.  If the chip delivered a fixed time per tile, 153 blocks (60 % of the CUs)
would take as long as 255; if it delivers a fixed AGGREGATE tile rate (power / fabric bound), the time follows the block count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
D, S = 128, 4352
g = torch.Generator(device=dev).manual_seed(0)
ops.tune("attn", 10)
ops.tune("attn_split", 0)
for H in (3, 6, 9, 12, 15, 18, 21, 24, 27, 30, 45, 60):
    qkv = torch.randn(1, S, 3 * H * D, device=dev, generator=g).to(torch.bfloat16)
    best = 1e9
    for rnd in range(3):
        for i in range(2):
            ops.attention(qkv, H, D)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10):
            ops.attention(qkv, H, D)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    nb = 17 * H
    print(f"H {H:2d}: {nb:4d} blocks = {nb / 256:.2f} rounds of 256 CUs: {best * 1e3:7.1f} us, {nb * 68 / (best * 1e3):6.1f} tiles/us aggregate, "
          f"{4.0 * H * S * S * D / best / 1e9:6.0f} TF", flush=True)
