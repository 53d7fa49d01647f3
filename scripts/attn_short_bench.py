"""Attention kernels on the short sequences of the sub-1024 resolutions (round 6): FLUX at 512 x 512 is S = 1280 (5 query blocks of 256 x 24 heads =
120 workgroups of the one-wave-per-SIMD kernel: half a round), batches of it fill the CUs.  modes: 4 lean kernel (128 queries per workgroup),
9 phase-alternating, 10 one wave per SIMD."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
H, D = 24, 128
g = torch.Generator(device=dev).manual_seed(0)
for B, S in ((1, 1280), (2, 1280), (3, 1280), (4, 1280), (8, 1280), (1, 2560), (2, 2560), (1, 768), (4, 768), (1, 4352)):
    qkv = torch.randn(B, S, 3 * H * D, device=dev, generator=g).to(torch.bfloat16)
    row = []
    ref = None
    for mode in (-1, 4, 9, 10):
        ops.tune("attn", mode)
        best = 1e9
        for rnd in range(3):
            for i in range(2):
                y = ops.attention(qkv, H, D)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(10):
                y = ops.attention(qkv, H, D)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        ops.tune("attn", -1)
        if ref is None:
            ref = y.float()
        err = float((y.float() - ref).abs().max())
        row.append(f"mode {mode:2d}: {best * 1e3:7.1f} us {4.0 * B * H * S * S * D / best / 1e9:6.0f} TF (|d| {err:.1e})")
    print(f"B {B} S {S}: " + " | ".join(row), flush=True)
