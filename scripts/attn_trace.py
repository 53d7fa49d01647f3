"""lab: phase timeline of the phase-alternating attention kernel (a -DDK4_TRACE=1 build through DK_HIP_LIB, scripts/build_attn_abl.sh
K=4): s_memtime stamps of waves 0 (group A) and 4 (group B) of workgroup 0 around the phases of tiles 20..27 on the FLUX shape.
Every stamp costs ~70 cycles (scalar memory round trip), the split of the M phase another LDS latency."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
B, H, S, D = 1, 24, 4352, 128
qkv = torch.randn(B, S, 3 * H * D, device=dev).to(torch.bfloat16)
ws = ops.attention_workspace(dev)
ops.tune("attn", 9)
for _ in range(3):
    ws.zero_()
    y = ops.attention(qkv, H, D, workspace=ws)
torch.cuda.synchronize()
t = ws[:1024].view(torch.int64).cpu().view(2, 64)[:, :48].reshape(2, 8, 6)
t0 = int(t[0, 0, 0])
for g in range(2):
    print("group", "AB"[g], "(wave", 4 * g, "of workgroup 0; s_memtime ticks)")
    for j in range(8):
        row = [int(t[g, j, k]) - t0 for k in range(6)]
        nxt = int(t[g, j + 1, 0]) - t0 if j < 7 else None
        print(f"  tile {20 + j} @ {row[0]:6d}: load issue {row[1] - row[0]:5d}  M: P.V {row[2] - row[1]:5d}  QK {row[3] - row[2]:5d}  barrier {row[4] - row[3]:5d}  V {row[5] - row[4]:5d}" +
              (f"  barrier {nxt - row[5]:5d}  tile {nxt - row[0]:5d}" if nxt is not None else ""))
ws.zero_()
