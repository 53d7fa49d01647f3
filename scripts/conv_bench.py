"""VAE-decoder conv shapes (1024^2 image), dk_conv3x3_bf16 under dk_tune_set("gemm", mode): 128 = the 128^2-tile kernel, 9 = the 256^2
kernel's conv form, -1 automatic (lab)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
# (name, H_out, W_out, C, O, upsample, count per decode)
shapes = [("128^2 512->512", 128, 128, 512, 512, False, 10), ("256^2 up 512->512", 256, 256, 512, 512, True, 1), ("256^2 512->512", 256, 256, 512, 512, False, 6),
          ("512^2 up 512->512", 512, 512, 512, 512, True, 1), ("512^2 512->256", 512, 512, 512, 256, False, 1), ("512^2 256->256", 512, 512, 256, 256, False, 5),
          ("1024^2 up 256->256", 1024, 1024, 256, 256, True, 1), ("1024^2 256->128", 1024, 1024, 256, 128, False, 1), ("1024^2 128->128", 1024, 1024, 128, 128, False, 5)]
g = torch.Generator(device=dev).manual_seed(0)
modes = [int(m) for m in (sys.argv[1:] or ["128", "9"])]
tot = {m: 0.0 for m in modes}
for name, H, W, C, O, ups, cnt in shapes:
    hs, ws = (H // 2, W // 2) if ups else (H, W)
    x = (torch.randn(1, hs, ws, C, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(O, 3, 3, C, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    b = torch.zeros(O, device=dev, dtype=torch.bfloat16)
    flops = 2.0 * H * W * O * 9 * C
    line = []
    for m in modes:
        if m == 9 and O % 256:
            line.append("m9: n/a")
            continue
        ops.tune("gemm", m)
        best = 1e9
        for rnd in range(3):
            ops.conv3x3(x, w, b, upsample=ups)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv3x3(x, w, b, upsample=ups)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        ops.tune("gemm", -1)
        tot[m] += best * cnt
        line.append(f"m{m}: {best * 1e3:7.1f} us {flops / best / 1e9:6.1f} TF")
    print(f"{name:22s} x{cnt}: " + "   ".join(line), flush=True)
print("per decode (ms, shapes a mode cannot run excluded): " + "  ".join(f"m{m}: {tot[m]:.2f}" for m in modes))
