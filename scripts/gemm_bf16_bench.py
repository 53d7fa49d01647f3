"""bf16 GEMM microbench (lab): dk_gemm_bf16 on the FLUX / SD3 shapes; COLD_W=n cycles through n copies of the weight (more than the
256 MB Infinity Cache in total) so that every launch streams its W from HBM as in the model; DK_HIP_LIB selects the library build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
shapes = [("qkv img", 4096, 9216, 3072), ("o_proj img", 4096, 3072, 3072), ("fc1 img", 4096, 12288, 3072), ("fc2 img", 4096, 3072, 12288),
          ("linear1", 4352, 21504, 3072), ("linear2", 4352, 3072, 15360), ("sd3 qkv", 8192, 4608, 1536), ("sd3 fc1", 8192, 6144, 1536),
          ("sd3 fc2", 8192, 1536, 6144)]
if os.environ.get("SWEEP"):  # per-round fixed cost: the same 768 tiles (3 rounds of 256 CUs) at growing K; 204 tiles (one round)
    shapes = [(f"N12288 K{k}", 4096, 12288, k) for k in (512, 1024, 2048, 3072, 6144)] + [(f"N3072 K{k}", 4352, 3072, k) for k in (1024, 3072, 6144, 15360)]
if os.environ.get("MF"):
    ops.tune("gemm_mf", int(os.environ["MF"]))
g = torch.Generator(device=dev).manual_seed(0)
ncopy = int(os.environ.get("COLD_W", "1"))
epi = {"bias": ops.DK_EPI_BIAS, "gelu": ops.DK_EPI_BIAS_GELU}[os.environ.get("EPI", "bias")]
out = []
for name, M, N, K in shapes:
    x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wlist = [w] + [w.clone() for _ in range(ncopy - 1)]
    best = 1e9
    for rnd in range(3):
        for i in range(3):
            ops.linear(x, wlist[i % ncopy], b, out=y, epilogue=epi)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24):
            ops.linear(x, wlist[i % ncopy], b, out=y, epilogue=epi)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 24)
    chk = ""
    if os.environ.get("CHECK"):
        ref = x[:512].float() @ w.float().t()
        ops.linear(x, w, b, out=y, epilogue=epi)
        got = y[:512].float()
        if epi == ops.DK_EPI_BIAS:
            chk = f" relL2 {float((got - ref).norm() / ref.norm()):.1e}"
    out.append(f"{name} {M}x{N}x{K}: {best * 1e3:7.1f} us {2.0 * M * N * K / best / 1e9:7.1f} TF{chk}")
print(os.environ.get("DK_HIP_LIB", "default lib"), "MF=" + os.environ.get("MF", "auto"), "EPI=" + os.environ.get("EPI", "bias"), "COLD_W=" + os.environ.get("COLD_W", "1"), " | ".join(out), flush=True)
