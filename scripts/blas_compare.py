"""Vendor-library yardstick for the GEMM shapes of the hot path: torch.matmul (hipBLASLt / rocBLAS behind it) on the same
[M,K] x [N,K]^T bf16 products as scripts/gemm_lab.cpp, best of 5 x 20 launches.  Measurement only: the product never
calls a library GEMM."""
import sys
import torch

SHAPES = [("flux qkv img", 4096, 9216, 3072), ("flux o img", 4096, 3072, 3072), ("flux fc1 img", 4096, 12288, 3072),
          ("flux fc2 img", 4096, 3072, 12288), ("flux single linear1", 4352, 21504, 3072), ("flux single l2", 4352, 3072, 15360),
          ("sd3 qkv", 8192, 4608, 1536), ("sd3 fc2", 8192, 1536, 6144), ("square 4096", 4096, 4096, 4096),
          ("square 8192", 8192, 8192, 8192)]


def main():
    dev = torch.device("cuda:0")
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        wt = w.t()
        for _ in range(3):
            torch.matmul(a, wt, out=out)
        best = 1e30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            e0.record()
            for _ in range(iters):
                torch.matmul(a, wt, out=out)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
        print(f"{name:22s} {M:5d}x{N:5d}x{K:5d}  torch.matmul: {2.0 * M * N * K / best / 1e9:7.1f} TF  us: {best * 1e3:.1f}", flush=True)


if __name__ == "__main__":
    main()
