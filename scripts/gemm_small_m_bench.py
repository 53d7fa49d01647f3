"""Small-launch GEMM sweep (round 6): the block Linears of FLUX at the resolutions below 1024 x 1024, where o_proj / fc2 /
linear2 are a fraction of a round of the CUs.  Per shape: the automatic choice, the one-wave-per-SIMD kernel forced (gemm_v4 = 2: no K split),
and the 8-wave kernel without its K split -- cold weights (COLD_W copies), the split workspace handed in as the engines do."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import _lib, ops

dev = torch.device("cuda", 0)
lib = _lib.load()
ncopy = int(os.environ.get("COLD_W", "8"))
g = torch.Generator(device=dev).manual_seed(0)
ws = torch.zeros(int(lib.dk_gemm_workspace_bytes()), dtype=torch.uint8, device=dev)
h = 3072
shapes = []
for tag, M in (("512^2", 1280), ("768^2", 2560), ("1024^2", 4352)):
    shapes += [(f"{tag} o", M, h, h), (f"{tag} fc2", M, h, 4 * h), (f"{tag} linear2", M, h, 5 * h), (f"{tag} qkv", M, 3 * h, h), (f"{tag} fc1", M, 4 * h, h),
               (f"{tag} linear1", M, 7 * h, h)]
modes = [("auto", {}), ("v4 forced", {"gemm_v4": 2}), ("v3 no split", {"gemm": 9, "gemm_split": 0}), ("v3 split", {"gemm": 9})]
if os.environ.get("SMALLK"):  # the 128 x 128-tile kernel on the launches that are a fraction of a round of 256 x 256 tiles (240 workgroups at M = 1280, N = 3072)
    modes = [("auto", {}), ("128^2 kernel", {"gemm": 128})]
    shapes = [s for s in shapes if not s[0].startswith("1024")]
if os.environ.get("FORCED"):  # the producer-chain split of a remainder of MORE than half the CUs (finisher [0, ks) + c producer pieces in turn on each spare CU): stream-K's 4/5 form
    modes = [("v4", {"gemm_v4": 2}), ("v3 no split", {"gemm": 9, "gemm_split": 0}), ("v3 mf8 no split", {"gemm": 9, "gemm_split": 0, "gemm_mf": 8}),
             ("v3 mf8 FORCED split", {"gemm": 9, "gemm_split": 1, "gemm_mf": 8}), ("v3 mf7 FORCED split", {"gemm": 9, "gemm_split": 1, "gemm_mf": 7})]
    shapes = [s for s in shapes if s[0].startswith("1024")]
for name, M, N, K in shapes:
    x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wl = [w] + [w.clone() for _ in range(ncopy - 1)]
    ref = x[:256].float() @ w.float().t()
    row = []
    for mname, tune in modes:
        for k, v in tune.items():
            ops.tune(k, v)
        best = 1e9
        for rnd in range(3):
            for i in range(3):
                ops.linear(x, wl[i % ncopy], b, out=y, workspace=ws)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(16):
                ops.linear(x, wl[i % ncopy], b, out=y, workspace=ws)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 16)
        ops.linear(x, w, b, out=y, workspace=ws)
        err = float((y[:256].float() - ref).norm() / ref.norm())
        for k in tune:
            ops.tune(k, -1)
        row.append(f"{mname} {best * 1e3:6.1f} us {2.0 * M * N * K / best / 1e9:6.0f} TF ({err:.0e})")
    print(f"{name:16s} {M}x{N}x{K}: " + " | ".join(row), flush=True)
