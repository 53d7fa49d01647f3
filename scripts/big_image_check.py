import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, numpy as np
from diffusionkit_amd.pipeline import FluxPipeline
dev = torch.device("cuda", 0)
pipe = FluxPipeline(w16=True, a16=True, device=dev)
for ls in ((96, 160), (192, 192)):
    t0 = time.time()
    img, log = pipe.generate_image("x", num_steps=2, latent_size=ls, seed=0, verbose=False)
    torch.cuda.synchronize()
    a = np.asarray(img)
    print(ls, img.size, a.mean().round(2), a.std().round(2), "iter", [round(t, 3) for t in log["denoising"]["iter_time"]], "decode", log["decoding"]["time"], "total", round(time.time() - t0, 2))
