"""VAE decode alone (lab): N decodes of a 128 x 128 latent -> 1024 x 1024 at the production channel plan, wall time per decode;
TUNE="key=value,..." sets dk_tune_set knobs (e.g. conv_halo=0 | 1 | 2).  Under `rocprofv3 --kernel-trace --stats` it is the
per-kernel breakdown of the decoder (profiles/r03_vae_kernel_stats*.md)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionkit_amd import _lib
from diffusionkit_amd.config import VAEDecoderConfig
from diffusionkit_amd.engine import VAEDecoderEngine
from diffusionkit_amd.weights import pack_vae, synth_vae_weights

dev = torch.device("cuda", 0)
lib = _lib.load()
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    k, v = kv.split("=")
    _lib.check(lib.dk_tune_set(k.encode(), int(v)), "tune")
B = int(os.environ.get("BATCH", "1"))
n = int(os.environ.get("N", "10"))
cfg = VAEDecoderConfig()
eng = VAEDecoderEngine(cfg, pack_vae(cfg, synth_vae_weights(cfg, seed=4321), dev))
z = torch.randn(B, 128, 128, 16, generator=torch.Generator().manual_seed(3)).to(dev)
for _ in range(2):
    eng.decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    eng.decode(z)
torch.cuda.synchronize()
print(f"TUNE={os.environ.get('TUNE', '')} batch {B}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per decode", flush=True)
