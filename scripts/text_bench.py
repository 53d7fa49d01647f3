#!/usr/bin/env python
"""Times the text-conditioning step at production sizes with seeded synthetic weights:
CLIP-L / CLIP-G (77 tokens) and the T5-XXL encoder (256 / 512 tokens), once per prompt in the reference
(mlx/__init__.py:197-251, 642-671).  Prints one JSON line per encoder."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from diffusionkit_amd import text as tx  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def clip_flops(c, n, b):
    d = c.model_dims
    return b * c.num_layers * (2 * n * d * d * 4 + 2 * n * d * 4 * d * 2 + 4 * n * n * d)


def t5_flops(c, n, b):
    d, inner = c.d_model, c.d_kv * c.num_heads
    return b * c.num_layers * (2 * n * d * inner * 4 + 2 * n * d * c.d_ff * 3 + 4 * n * n * inner)


def main():
    dev = torch.device("cuda", 0)
    for name, cfg in (("clip_l", tx.CLIP_L), ("clip_g", tx.CLIP_G)):
        eng = tx.CLIPTextEngine(cfg, tx.synth_clip_weights(cfg, device=dev), dev)
        for b in (1, 2):
            tok = torch.randint(1, 49000, (b, 77))
            ms = timed(lambda: eng(tok))
            print(json.dumps({"encoder": name, "batch": b, "tokens": 77, "ms": round(ms, 3),
                              "tflops": round(clip_flops(cfg, 77, b) / ms / 1e9, 1)}))
        del eng
    cfg = tx.T5_XXL
    eng = tx.T5EncoderEngine(cfg, tx.synth_t5_weights(cfg, device=dev), dev)
    for b, n in ((1, 256), (1, 512), (2, 512)):
        tok = torch.randint(1, 32000, (b, n))
        ms = timed(lambda: eng(tok), n=3)
        print(json.dumps({"encoder": "t5_xxl", "batch": b, "tokens": n, "ms": round(ms, 3), "tflops": round(t5_flops(cfg, n, b) / ms / 1e9, 1)}))


if __name__ == "__main__":
    main()
