"""Halo conv kernel microbench (lab): the VAE's fused conv shapes through ops.conv3x3_gn; DK_HIP_LIB selects an ablation build
(scripts/build_halo_abl.sh).  Prints us and TFLOP/s per shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    k, v = kv.split("=")
    ops.tune(k, int(v))
shapes = [("128->128 @1024^2", 1024, 1024, 128, 128), ("512->512 @256^2", 256, 256, 512, 512), ("256->256 @512^2", 512, 512, 256, 256),
          ("512->512 @128^2", 128, 128, 512, 512)]
if os.environ.get("SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SHAPES"].split(",")]
g = torch.Generator(device=dev).manual_seed(0)
out = []
for name, H, W, C, O in shapes:
    x = torch.randn(1, H, W, C, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(O, 9 * C, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    b = torch.zeros(O, device=dev, dtype=torch.bfloat16)
    gam, bet = torch.ones(C, device=dev, dtype=torch.bfloat16), torch.zeros(C, device=dev, dtype=torch.bfloat16)
    tab = ops.groupnorm_table(x, gam, bet, 32, 1e-5)
    best = 1e9
    for rnd in range(3):
        for _ in range(2):
            ops.conv3x3_gn(x, w, b, gn_table=tab, stats_groups=32)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv3x3_gn(x, w, b, gn_table=tab, stats_groups=32)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    out.append(f"{name}: {best * 1e3:7.1f} us {2.0 * H * W * 9 * C * O / best / 1e9:7.1f} TF")
print(os.environ.get("DK_HIP_LIB", "default lib").split("/")[-2] if os.environ.get("DK_HIP_LIB") else "default", os.environ.get("TUNE", ""), " | ".join(out), flush=True)
