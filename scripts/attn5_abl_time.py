"""one round of attention5 workgroups (16 heads x 16 query blocks of 256 = 256 workgroups, 64 key tiles): us per launch for the library DK_HIP_LIB points at"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionkit_amd import ops
dev = torch.device("cuda", 0)
B, H, S, D = 1, 16, 4096, 128
qkv = torch.randn(B, S, 3 * H * D, device=dev).to(torch.bfloat16)
res = []
for mode in (10, 9):
    ops.tune("attn", mode)
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
    res.append(best * 1e3)
print(f"{os.environ.get('DK_HIP_LIB', 'shipped').split('/')[-2] if os.environ.get('DK_HIP_LIB') else 'shipped':8s} attn5 {res[0]:7.1f} us = {res[0] / 64 * 1e3:6.0f} ns per tile ({4.0 * B * H * S * S * D / res[0] / 1e6:5.0f} TF) | attn4 {res[1]:7.1f} us", flush=True)
