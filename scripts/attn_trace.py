"""lab: phase timeline of the phase-alternating attention kernel (a -DDK4_TRACE=1 build through DK_HIP_LIB, scripts/build_attn_abl.sh
K=4): s_memtime stamps of all eight waves of workgroup 0 around the phases of tiles 20..27 on the FLUX shape (round 4: every wave, to see which
wave a barrier waits for).
Every stamp costs ~70 cycles (scalar memory round trip), the split of the M phase another LDS latency."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
B, H, S, D = 1, 24, 4352, 128
qkv = torch.randn(B, S, 3 * H * D, device=dev).to(torch.bfloat16)
ws = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)  # trace buffer (8 waves x 64 stamps)
ops.tune("attn", 9)
for _ in range(3):
    ws.zero_()
    y = ops.attention(qkv, H, D, workspace=ws)
torch.cuda.synchronize()
t = ws[:4096].view(torch.int64).cpu().view(8, 64)[:, :48].reshape(8, 8, 6)
t0 = int(t[:, 0, 0].min())
print("s_memtime ticks; per wave and tile: start | M phase (P.V + QK) | wait at barrier 1 | V phase | wait at barrier 2   (waves 0-3: group A, 4-7: group B; wave w and w + 4 share a SIMD)")
for j in range(8):
    print(f"tile {20 + j}")
    for w in range(8):
        r = [int(t[w, j, k]) - t0 for k in range(6)]
        nxt = int(t[w, j + 1, 0]) - t0 if j < 7 else None
        print(f"  wave {w}: @{r[0]:6d}  M {r[3] - r[0]:5d} (P.V {r[2] - r[1]:5d}, QK {r[3] - r[2]:5d})  wait {r[4] - r[3]:5d}  V {r[5] - r[4]:5d}" +
              (f"  wait {nxt - r[5]:5d}  period {nxt - r[0]:5d}" if nxt is not None else ""))
ws.zero_()
