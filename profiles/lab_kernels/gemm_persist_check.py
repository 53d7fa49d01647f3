import sys; sys.path.insert(0, "/root/repo")
import torch
from diffusionkit_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
ok = True
for M, N, K in ((4352, 3072, 3072), (4608, 12288, 3072), (4352, 3072, 12288), (1000, 1024, 2048)):
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    outs = []
    for per in (0, 1):
        ops.tune("gemm", 10); ops.tune("gemm_persist", per)
        outs.append(ops.linear(x, w, b))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.linear(x, w, b)
        e1.record(); torch.cuda.synchronize()
        print(M, N, K, "persist", per, f"{e0.elapsed_time(e1) * 100:.1f} us", f"{2.0 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9:.0f} TF")
    ok &= torch.equal(outs[0], outs[1])
ops.tune("gemm", -1); ops.tune("gemm_persist", 0)
print("IDENTICAL" if ok else "MISMATCH")
