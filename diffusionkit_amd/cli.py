"""``python -m diffusionkit_amd.cli``: the reference's ``diffusionkit-cli`` in front of this build's
pipelines.  Flags, defaults, the per-model height / width / shift tables, the FLUX "no CFG" rule, the benchmark-mode
warm-up and the error behaviour follow python/src/diffusionkit/mlx/scripts/generate_images.py:15-187.

There is no hub access on an MI355X box, so checkpoints are named explicitly: ``--local-ckpt`` is the MMDiT
``.safetensors`` (as in the reference) and ``--ckpt KEY=PATH`` adds the other parts (``vae_decoder``, ``vae_encoder``, ``clip_l``,
``clip_g``, ``t5``, ``t5_tokenizer``; ``tokenizer_l`` / ``tokenizer_g`` take ``vocab.json,merges.txt``).  Parts that are not named
run on seeded synthetic weights / synthetic conditioning and the log line says so -- useful for timing, not for pictures.
"""
from __future__ import annotations

import argparse
import logging
from typing import Dict, Optional, Sequence

logger = logging.getLogger("diffusionkit_amd.cli")

# generate_images.py:15-38
HEIGHT = {
    "argmaxinc/mlx-stable-diffusion-3-medium": 512,
    "argmaxinc/mlx-stable-diffusion-3.5-large": 1024,
    "argmaxinc/mlx-stable-diffusion-3.5-large-4bit-quantized": 1024,
    "argmaxinc/mlx-FLUX.1-schnell": 512,
    "argmaxinc/mlx-FLUX.1-schnell-4bit-quantized": 512,
    "argmaxinc/mlx-FLUX.1-dev": 512,
}
WIDTH = dict(HEIGHT)
SHIFT = {k: (1.0 if "FLUX" in k else 3.0) for k in HEIGHT}


def build_parser(model_versions: Sequence[str]) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="diffusionkit-amd-cli",
                                description="Generate images from a text (and an optional image) prompt on MI355X")
    p.add_argument("--prompt", required=True, help="Text prompt")
    p.add_argument("--image-path", type=str, default=None, help="Path to the image prompt")
    p.add_argument("--model-version", choices=tuple(model_versions), default="argmaxinc/mlx-FLUX.1-schnell",
                   help="Diffusion model version, e.g. FLUX-1.schnell, stable-diffusion-3-medium")
    p.add_argument("--steps", type=int, default=50, help="Number of diffusion steps.")
    p.add_argument("--cfg", type=float, default=5.0, help="Classifier-free guidance weight")
    p.add_argument("--negative_prompt", default="", help="Negative text prompt")
    p.add_argument("--preload-models", action="store_true", help="Accepted for compatibility: models are always resident in HBM.")
    p.add_argument("--output-path", "-o", default="out.png", help="Path to save the output image.")
    p.add_argument("--seed", type=int, help="Seed for the random number generator.")
    p.add_argument("--verbose", "-v", action="store_true", help="Print detailed information.")
    p.add_argument("--shift", type=float, help="Shift for diffusion sampling")
    p.add_argument("--t5", action="store_true", help="Engages T5 for stronger text embeddings.")
    p.add_argument("--height", type=int, help="Height of the output image")
    p.add_argument("--width", type=int, help="Width of the output image")
    p.add_argument("--no-low-memory-mode", action="store_false", dest="low_memory_mode",
                   help="Accepted for compatibility: nothing is offloaded with 288 GB of HBM.")
    p.add_argument("--benchmark-mode", action="store_true", help="Warm the kernels up with a one-step run first.")
    p.add_argument("--denoise", type=float, default=0.0,
                   help="Denoising factor when an input image is provided. (between 0.0 and 1.0)")
    p.add_argument("--local-ckpt", default=None, type=str, help="Path to the local mmdit checkpoint.")
    p.add_argument("--ckpt", action="append", default=[], metavar="KEY=PATH",
                   help="Further local checkpoint parts (vae_decoder, vae_encoder, clip_l, clip_g, t5, t5_tokenizer, "
                        "tokenizer_l=vocab.json,merges.txt, tokenizer_g=...). Repeatable.")
    p.add_argument("--device", default=None, help="HIP device, e.g. cuda:0 (default: the current device)")
    return p


def checkpoint_dict(local_ckpt: Optional[str], parts: Sequence[str]) -> Optional[Dict[str, object]]:
    """--local-ckpt / --ckpt KEY=PATH -> the ``local_ckpt`` dict DiffusionPipeline takes."""
    out: Dict[str, object] = {}
    if local_ckpt:
        out["mmdit"] = local_ckpt
    for kv in parts:
        if "=" not in kv:
            raise ValueError(f"--ckpt expects KEY=PATH, got {kv!r}")
        k, v = kv.split("=", 1)
        out[k] = tuple(v.split(",")) if k.startswith("tokenizer_") else v
    return out or None


def resolve(args) -> dict:
    """The argument post-processing of generate_images.py:114-150, as data (so that it can be tested without a GPU)."""
    cfg = args.cfg
    if "FLUX" in args.model_version and cfg > 0.0:
        logger.warning(f"Disabling CFG for {args.model_version} model.")
        cfg = 0.0
    if args.denoise < 0.0 or args.denoise > 1.0:
        raise ValueError("Denoising factor must be between 0.0 and 1.0")
    height = args.height or HEIGHT[args.model_version]
    width = args.width or WIDTH[args.model_version]
    assert height % 16 == 0, f"Height must be divisible by 16 ({height}/16={height/16})"
    assert width % 16 == 0, f"Width must be divisible by 16 ({width}/16={width/16})"
    return {"cfg": cfg, "shift": args.shift or SHIFT[args.model_version], "height": height, "width": width,
            "flux": "FLUX" in args.model_version, "low_memory_mode": args.low_memory_mode and not args.benchmark_mode}


def main(argv: Optional[Sequence[str]] = None, pipeline_overrides: Optional[dict] = None):
    """Returns (PIL.Image, log) after saving the image.  ``pipeline_overrides`` are extra keyword arguments for the pipeline
    constructor (tests pass tiny configs through it)."""
    from .config import MMDIT_CKPT
    args = build_parser(tuple(MMDIT_CKPT.keys())).parse_args(argv)
    logging.basicConfig(level=logging.INFO if args.verbose else logging.WARNING)
    r = resolve(args)
    from .pipeline import DiffusionPipeline, FluxPipeline
    pipeline_class = FluxPipeline if r["flux"] else DiffusionPipeline
    sd = pipeline_class(w16=True, shift=r["shift"], use_t5=args.t5, model_version=args.model_version,
                        low_memory_mode=r["low_memory_mode"], a16=True, local_ckpt=checkpoint_dict(args.local_ckpt, args.ckpt),
                        device=args.device, **(pipeline_overrides or {}))
    logger.info(f"Output image resolution will be {r['height']}x{r['width']}")
    latent_size = (r["height"] // 8, r["width"] // 8)
    if args.benchmark_mode:
        logger.info("Running in benchmark mode. Warming up the models. (generated latents will be discarded)")
        sd.generate_image(args.prompt, cfg_weight=r["cfg"], num_steps=1, seed=args.seed, negative_text=args.negative_prompt,
                          latent_size=latent_size, verbose=False)
        logger.info("Benchmark mode: Warming up the models done.")
    image, log = sd.generate_image(args.prompt, cfg_weight=r["cfg"], num_steps=args.steps, seed=args.seed,
                                   negative_text=args.negative_prompt, latent_size=latent_size, image_path=args.image_path,
                                   denoise=args.denoise, verbose=args.verbose)
    if log["text_encoding"].get("synthetic"):
        logger.warning("no text-encoder checkpoints were named (--ckpt clip_l=... t5=...): the conditioning is synthetic")
    image.save(args.output_path)
    logger.info(f"Saved the image to {args.output_path}")
    return image, log


if __name__ == "__main__":
    main()
