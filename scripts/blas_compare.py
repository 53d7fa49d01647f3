"""Vendor-library yardstick for the GEMM shapes of the hot path: torch.matmul (hipBLASLt / rocBLAS behind it) against dk_gemm_bf16 on
the same [M,K] x [N,K]^T bf16 products (plain GEMM + bias for ours, no epilogue for the library), in one process, interleaved.
COLD_W=n cycles through n copies of the weight -- in the model every launch streams its own weights from HBM.  Measurement only:
the product never calls a library GEMM.  DK_MODES=9,10,13 times dk_gemm_bf16 under several dk_tune_set("gemm", mode) values
(9 = gemm256v3.hip, 10 .. 13 = gemm256v4.hip's schedule variants; default: the automatic choice)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

SHAPES = [("flux qkv img", 4096, 9216, 3072), ("flux o img", 4096, 3072, 3072), ("flux fc1 img", 4096, 12288, 3072),
          ("flux fc2 img", 4096, 3072, 12288), ("flux single linear1", 4352, 21504, 3072), ("flux single l2", 4352, 3072, 15360),
          ("sd3 qkv", 8192, 4608, 1536), ("sd3 fc1", 8192, 6144, 1536), ("sd3 fc2", 8192, 1536, 6144), ("square 8192", 8192, 8192, 8192)]


def main():
    dev = torch.device("cuda:0")
    iters = 24
    ncopy = int(os.environ.get("COLD_W", "1"))
    print(f"COLD_W={ncopy}")
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fns = {"torch.matmul": lambda i: torch.matmul(a, ws[i % ncopy].t(), out=out)}
        modes = [int(m) for m in os.environ.get("DK_MODES", "-1").split(",")]
        for m in modes:
            def run(i, m=m):
                ops.tune("gemm", m)
                ops.linear(a, ws[i % ncopy], b, out=out)
            fns["dk_gemm_bf16" + (f"[{m}]" if m >= 0 else "")] = run
        best = {k: 1e30 for k in fns}
        for rnd in range(4):
            for k, fn in fns.items():
                for i in range(3):
                    fn(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(iters):
                    fn(i)
                e1.record()
                e1.synchronize()
                best[k] = min(best[k], e0.elapsed_time(e1) / iters)
        fl = 2.0 * M * N * K
        print(f"{name:20s} {M:5d}x{N:5d}x{K:5d}  " + "   ".join(f"{k}: {v * 1e3:6.1f} us {fl / v / 1e9:7.1f} TF" for k, v in best.items()) +
              "   ours / library = " + " ".join(f"{best['torch.matmul'] / v:.3f}" for k, v in best.items() if k != "torch.matmul"), flush=True)
    ops.tune("gemm", -1)


if __name__ == "__main__":
    main()
