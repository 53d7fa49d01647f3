"""Attention kernel A/B (lab): dk_attention_bf16 under dk_tune_set("attn", mode) on the bench shapes, interleaved rounds in one
process (guide rule 24).  modes: 4 = dk_attn2 (lean kernel, 4 waves), 9 = dk_attn4 (phase-alternating, 8 waves, D = 128).
(Round 5: the pipelined kernel, mode 7 / 7b, lives in profiles/lab_kernels/attention3_pipelined.hip.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
shapes = [("flux B1 S4352 D128", 1, 24, 4352, 128), ("flux-dev B1 S4608 D128", 1, 24, 4608, 128), ("sd3 B2 S4685 D64", 2, 24, 4096 + 589, 64),
          ("flux B4", 4, 24, 4352, 128)]
modes = sys.argv[1:] or ["9", "4"]
if os.environ.get("ATTN_SHAPES") == "d64":  # the SD3 family (round 4: lean kernel against the lab build with profiles/lab_kernels/attention5_two_query_blocks.hip, modes 5 / 6)
    shapes = [("sd3 B2 S4685 H24 D64", 2, 24, 4096 + 589, 64), ("sd3.5-large B2 S4685 H38 D64", 2, 38, 4096 + 589, 64),
              ("sd3 512^2 B2 S1613 H24 D64", 2, 24, 1024 + 589, 64), ("sd3 B1 S4685 H24 D64", 1, 24, 4096 + 589, 64)]
elif os.environ.get("ATTN_SHAPES"):
    shapes = shapes[:int(os.environ["ATTN_SHAPES"])]
for name, B, H, S, D in shapes:
    qkv = (torch.randn(B, S, 3 * H * D, device=dev) * 1.0).to(torch.bfloat16)
    flops = 4.0 * B * H * S * S * D
    best = {m: 1e9 for m in modes}
    outs = {}
    for rnd in range(5):
        for m in modes:
            ops.tune("attn", int(m))
            for _ in range(2):
                y = ops.attention(qkv, H, D)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = ops.attention(qkv, H, D)
            e1.record()
            torch.cuda.synchronize()
            best[m] = min(best[m], e0.elapsed_time(e1) / 10)
            outs[m] = y
    ops.tune("attn", -1)
    ref = outs[modes[0]].float()
    print(name, "  ".join(f"m{m}: {best[m] * 1e3:7.1f} us {flops / best[m] / 1e9:7.1f} TF (max|d| vs m{modes[0]} {float((outs[m].float() - ref).abs().max()):.3g})" for m in modes), flush=True)
