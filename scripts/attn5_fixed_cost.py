"""attention5 / attention4: time of ONE round of 256 workgroups against the number of key tiles (S = 1024 / 2048 / 4096 / 8192 with H = 64 / 32 / 16 / 8):
slope = time per key tile, intercept = fixed cost per workgroup (launch, Q load, first tiles' latency, output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionkit_amd import ops
dev = torch.device("cuda", 0)
D = 128
for mode in (10, 9):
    ops.tune("attn", mode)
    pts = []
    for S, H in ((1024, 64), (2048, 32), (4096, 16), (8192, 8)):
        qkv = torch.randn(1, S, 3 * H * D, device=dev).to(torch.bfloat16)
        best = 1e9
        for rnd in range(3):
            ops.attention(qkv, H, D)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attention(qkv, H, D)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        pts.append((S // 64, best * 1e3))
    slope = (pts[-1][1] - pts[0][1]) / (pts[-1][0] - pts[0][0])
    print(f"{'attn5' if mode == 10 else 'attn4'}: " + "  ".join(f"nt={n}: {t:6.1f} us" for n, t in pts) + f" | {slope * 1e3:5.0f} ns per tile, fixed {pts[0][1] - slope * pts[0][0]:5.1f} us", flush=True)
ops.tune("attn", -1)
