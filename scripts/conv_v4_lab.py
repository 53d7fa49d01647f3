"""conv256v4 (one wave per SIMD, asm body) against conv_halo (8 waves) through ops.conv3x3_gn: same inputs, dk_tune_set("conv_v4", 0 | 2).
The outputs must be bit-identical (same products, same chunk-major fp32 order, same rounding points); the GroupNorm partials of the two
kernels are sums in different orders (compared with a tolerance).  Then us and TFLOP/s per shape for both.  Small shapes first: a
scheduling bug shows up there before the big launches run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
BF = torch.bfloat16
# name, B, H, W (output), C, O, gn, res, upsample
shapes = [("small gn+res 128->256 @32x48 B2", 2, 32, 48, 128, 256, True, True, False),
          ("small plain ups 128->256 @32x48 B2", 2, 32, 48, 128, 256, False, False, True),
          ("small gn 256->512 @48x48", 1, 48, 48, 256, 512, True, False, False),
          ("small gn+res 128->128 @32x48 B2", 2, 32, 48, 128, 128, True, True, False),
          ("small plain ups 256->128 @32x32", 1, 32, 32, 256, 128, False, False, True),
          ("128->128 @1024^2 gn+res", 1, 1024, 1024, 128, 128, True, True, False),
          ("256->128 @1024^2 gn", 1, 1024, 1024, 256, 128, True, False, False),
          ("512->512 @128^2 gn+res", 1, 128, 128, 512, 512, True, True, False),
          ("512->512 @256^2 gn+res", 1, 256, 256, 512, 512, True, True, False),
          ("512->512 @256^2 ups", 1, 256, 256, 512, 512, False, False, True),
          ("512->512 @512^2 ups", 1, 512, 512, 512, 512, False, False, True),
          ("512->256 @512^2 gn", 1, 512, 512, 512, 256, True, False, False),
          ("256->256 @512^2 gn+res", 1, 512, 512, 256, 256, True, True, False),
          ("256->256 @1024^2 ups", 1, 1024, 1024, 256, 256, False, False, True)]
if os.environ.get("SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SHAPES"].split(",")]
allok = True
for name, B, H, W, C, O, gn, res, ups in shapes:
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    x = (torch.randn(B, Hs, Ws, C, device=dev, generator=g) * 1.5 + 0.3).to(BF)
    w = (torch.randn(O, 9 * C, device=dev, generator=g) * 0.02).to(BF)
    b = (torch.randn(O, device=dev, generator=g) * 0.1).to(BF)
    r = torch.randn(B, H, W, O, device=dev, generator=g).to(BF) if res else None
    gam = (1 + 0.1 * torch.randn(C, device=dev, generator=g)).to(BF)
    bet = (0.1 * torch.randn(C, device=dev, generator=g)).to(BF)
    tab = ops.groupnorm_table(x, gam, bet, 32, 1e-5) if gn else None
    sg = 0 if ups else 32

    def call():
        return ops.conv3x3_gn(x, w, b, gn_table=tab, res=r, stats_groups=sg, upsample=ups)
    outs, times = [], []
    for mode in (0, 2):
        ops.tune("conv_v4", mode)
        o = call()
        torch.cuda.synchronize()
        outs.append(o)
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        times.append(best)
    y0, y1 = (outs[0][0], outs[1][0]) if sg else (outs[0], outs[1])
    same = torch.equal(y0, y1)
    nbad = int((y0 != y1).sum())
    maxd = float((y0.float() - y1.float()).abs().max())
    st = ""
    if sg:
        p0, p1 = outs[0][1].double(), outs[1][1].double()
        st = f" partials rel {float((p0 - p1).norm() / p0.norm()):.2e}"
        same = same and float((p0 - p1).norm() / p0.norm()) < 1e-5
    fl = 2.0 * B * H * W * 9 * C * O
    allok &= same
    print(f"{name}: {'IDENTICAL' if nbad == 0 else f'DIFFERENT ({nbad} values, max {maxd:.3g})'}{st} | conv_halo {times[0] * 1e3:7.1f} us {fl / times[0] / 1e9:6.0f} TF"
          f" | conv256v4 {times[1] * 1e3:7.1f} us {fl / times[1] / 1e9:6.0f} TF", flush=True)
ops.tune("conv_v4", 1)
print("ALL IDENTICAL" if allok else "MISMATCH", flush=True)
