"""Is the attention launch slower in the model because its K / V come from HBM?  (round 6)  FLUX shape (24 heads x 4352 tokens, D = 128): the same launch on ONE qkv buffer
(80 MB: warm in the 256 MB Infinity Cache after the first pass) and rotating over N buffers (N x 80 MB: every launch streams its K / V from HBM), and -- the model's
situation -- on a buffer that a 187 MB write (linear1's QKV + gelu outputs) has just passed over."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import ops

dev = torch.device("cuda", 0)
H, D, S = 24, 128, 4352
g = torch.Generator(device=dev).manual_seed(0)
bufs = [torch.randn(1, S, 3 * H * D, device=dev, generator=g).to(torch.bfloat16) for _ in range(6)]
junk = torch.empty(107 * 1024 * 1024 // 2, device=dev, dtype=torch.bfloat16)


def timed(fn, n=12):
    best = 1e9
    for rnd in range(3):
        fn(0)
        fn(1)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i in range(n):
            fn.pre(i)
            ev[i][0].record()
            fn(i)
            ev[i][1].record()
        torch.cuda.synchronize()
        best = min(best, sum(a.elapsed_time(b) for a, b in ev) / n)
    return best * 1e3


class Run:
    def __init__(self, nbuf, rewrite):
        self.nbuf, self.rewrite = nbuf, rewrite

    def pre(self, i):
        if self.rewrite:  # what linear1 does in front of the attention launch: the QKV buffer is REWRITTEN (80 MB) and 107 MB more go out beside it
            bufs[i % self.nbuf].mul_(1.0)
            junk.fill_(0.5)

    def __call__(self, i):
        ops.attention(bufs[i % self.nbuf], H, D)


for nbuf, rewrite, what in ((1, False, "one buffer, re-read (warm)"), (6, False, "six buffers in turn (480 MB: K / V from HBM)"),
                            (1, True, "one buffer, rewritten + 107 MB written beside it before every launch (the model's linear1)"),
                            (6, True, "six buffers in turn, each rewritten + 107 MB before its launch")):
    r = Run(nbuf, rewrite)
    t = timed(r)
    print(f"{what}: {t:7.1f} us  {4.0 * H * S * S * D / t / 1e6:6.0f} TF", flush=True)
