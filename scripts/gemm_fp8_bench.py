"""fp8 GEMM microbench (lab): dk_gemm_fp8 on the FLUX shapes, random e4m3 operands with random block scales; DK_HIP_LIB selects the
library build (ablation builds give wrong results but valid timings)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

from diffusionkit_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)

shapes = [("qkv img", 4096, 9216, 3072), ("o_proj img", 4096, 3072, 3072), ("fc1 img", 4096, 12288, 3072), ("fc2 img", 4096, 3072, 12288),
          ("linear1", 4352, 21504, 3072), ("linear2", 4352, 3072, 15360), ("square 8192", 8192, 8192, 8192)]
g = torch.Generator(device=dev).manual_seed(0)
out = []
for name, M, N, K in shapes:
    a8 = torch.randint(0, 256, (M, K), device=dev, dtype=torch.uint8, generator=g)
    a8 = torch.where((a8 & 0x7F) == 0x7F, torch.zeros_like(a8), a8)  # no NaN codes
    w8 = torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8, generator=g)
    w8 = torch.where((w8 & 0x7F) == 0x7F, torch.zeros_like(w8), w8)
    sc = torch.randint(118, 132, (ops.mx_scale_bytes(M, K),), device=dev, dtype=torch.uint8, generator=g)
    ws = torch.rand(N, device=dev, generator=g) * 1e-3
    # COLD_W=n: cycle through n copies of the weight (> the 256 MB Infinity Cache in total) so that every launch streams its W from HBM,
    # as in the model, where each of the 57 blocks has its own weights
    ncopy = int(os.environ.get("COLD_W", "1"))
    wlist = [w8] + [w8.clone() for _ in range(ncopy - 1)]
    best = 1e9
    for rnd in range(3):
        for i in range(3):
            ops.gemm_fp8(a8, sc, wlist[i % ncopy], ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24):
            ops.gemm_fp8(a8, sc, wlist[i % ncopy], ws)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 24)
    out.append(f"{name} {M}x{N}x{K}: {best * 1e3:7.1f} us {2.0 * M * N * K / best / 1e9:7.1f} TF")
print(os.environ.get("DK_HIP_LIB", "default lib"), "COLD_W=" + os.environ.get("COLD_W", "1"),  " | ".join(out), flush=True)
