"""Library yardstick for the joint-attention shapes of the hot path: torch.nn.functional.scaled_dot_product_attention
(whatever flash kernel PyTorch-ROCm dispatches to) on [B, H, S, D] bf16, best of 5 x 10 launches; TFLOP/s = 4 B H S^2 D / t.
Measurement only: the product never calls it."""
import torch
import torch.nn.functional as F

SHAPES = [("flux joint B1", 1, 24, 4352, 128), ("sd3 joint cfg B2", 2, 24, 4685, 64), ("flux B4", 4, 24, 4352, 128),
          ("flux-dev joint B1", 1, 24, 4608, 128)]


def main():
    dev = torch.device("cuda:0")
    for name, B, H, S, D in SHAPES:
        q, k, v = (torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
        for _ in range(3):
            F.scaled_dot_product_attention(q, k, v)
        best = 1e30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            e0.record()
            for _ in range(10):
                F.scaled_dot_product_attention(q, k, v)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print(f"{name:20s} B{B} H{H} S{S} D{D}  torch SDPA: {4.0 * B * H * S * S * D / best / 1e9:7.1f} TF ({best:.3f} ms)", flush=True)


if __name__ == "__main__":
    main()
