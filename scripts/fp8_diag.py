"""fp8 GEMM diagnostics (lab): unit / uniform scales to separate structural errors from the scale association."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionkit_amd import ops
from diffusionkit_amd.weights import quantize_weight_e4m3, dequantize_weight_e4m3
from tests import _fp8 as f8
dev = torch.device("cuda", 0)
BF = torch.bfloat16
for (M, N, K) in ((256, 256, 128), (256, 256, 256), (512, 512, 384), (384, 256, 384), (4352, 3072, 3072)):
    for mode in ("unit", "per_row", "per_block"):
        g = torch.Generator().manual_seed(M + N + K)
        v = (torch.randn(M, K, generator=g).clamp(-3, 3) * 16).to(torch.float8_e4m3fn)  # e4m3 payload, |v| <= 48
        qa = v.view(torch.uint8)
        if mode == "unit":
            e = torch.full((M, K // 32), 127, dtype=torch.uint8)
        elif mode == "per_row":
            e = (120 + (torch.arange(M) % 12))[:, None].expand(M, K // 32).contiguous().to(torch.uint8)
        else:
            e = (118 + torch.randint(0, 14, (M, K // 32), generator=g)).to(torch.uint8)
        a_dq = f8.mx8_decode(qa, e)
        w = (torch.randn(N, K, generator=g) * 0.02).to(BF)
        qw, ws = quantize_weight_e4m3(w)
        rows = (M + 127) // 128 * 128
        a8 = torch.zeros(rows, K, dtype=torch.uint8); a8[:M] = qa
        out = ops.gemm_fp8(a8.to(dev), f8.scales_to_array(e, rows).to(dev), qw.to(dev), ws.to(dev), M=M, k=K).float().cpu()
        ref = a_dq @ dequantize_weight_e4m3(qw, ws).t()
        err = (out - ref).abs()
        rel = float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref))
        bad_rows = (err.amax(1) > 0.05 * ref.abs().amax()).nonzero().flatten()
        bad_cols = (err.amax(0) > 0.05 * ref.abs().amax()).nonzero().flatten()
        print(f"{M}x{N}x{K} {mode:9s}: rel {rel:.3e}  bad rows {bad_rows.numel()} {bad_rows[:8].tolist()}  bad cols {bad_cols.numel()} {bad_cols[:8].tolist()}  finite {bool(torch.isfinite(out).all())}")
