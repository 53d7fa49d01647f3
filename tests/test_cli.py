"""diffusionkit_amd.cli against the reference CLI's argument handling (mlx/scripts/generate_images.py:15-187)."""
import numpy as np
import pytest
import torch

from diffusionkit_amd import cli
from diffusionkit_amd.config import MMDIT_CKPT, tiny_flux, tiny_sd3, tiny_vae, tiny_vae_encoder


def parse(argv):
    return cli.build_parser(tuple(MMDIT_CKPT.keys())).parse_args(argv)


def test_defaults_follow_the_reference_tables():
    a = parse(["--prompt", "a cat"])
    assert (a.model_version, a.steps, a.cfg, a.output_path, a.denoise, a.t5) == ("argmaxinc/mlx-FLUX.1-schnell", 50, 5.0, "out.png", 0.0, False)
    r = cli.resolve(a)
    assert r == {"cfg": 0.0, "shift": 1.0, "height": 512, "width": 512, "flux": True, "low_memory_mode": True}  # CFG off for FLUX
    a = parse(["--prompt", "x", "--model-version", "argmaxinc/mlx-stable-diffusion-3-medium", "--benchmark-mode"])
    r = cli.resolve(a)
    assert (r["cfg"], r["shift"], r["height"], r["flux"], r["low_memory_mode"]) == (5.0, 3.0, 512, False, False)
    a = parse(["--prompt", "x", "--model-version", "argmaxinc/mlx-stable-diffusion-3.5-large", "--shift", "2.5", "--height", "768"])
    r = cli.resolve(a)
    assert (r["shift"], r["height"], r["width"]) == (2.5, 768, 1024)
    assert set(cli.HEIGHT) == set(MMDIT_CKPT)  # every model version of the reference has its defaults


def test_argument_errors():
    with pytest.raises(ValueError, match="between 0.0 and 1.0"):
        cli.resolve(parse(["--prompt", "x", "--denoise", "1.5"]))
    with pytest.raises(AssertionError, match="divisible by 16"):
        cli.resolve(parse(["--prompt", "x", "--height", "520"]))
    with pytest.raises(SystemExit):
        parse([])  # --prompt is required
    with pytest.raises(SystemExit):
        parse(["--prompt", "x", "--model-version", "nope"])
    with pytest.raises(ValueError, match="KEY=PATH"):
        cli.checkpoint_dict(None, ["vae_decoder"])


def test_checkpoint_dict():
    assert cli.checkpoint_dict(None, []) is None
    d = cli.checkpoint_dict("flux.safetensors", ["vae_decoder=ae.safetensors", "tokenizer_l=vocab.json,merges.txt"])
    assert d == {"mmdit": "flux.safetensors", "vae_decoder": "ae.safetensors", "tokenizer_l": ("vocab.json", "merges.txt")}


@pytest.mark.gpu
def test_cli_end_to_end_tiny(tmp_path):
    """The whole command (benchmark-mode warm-up, txt2img, then img2img from the image it just wrote) on tiny configs."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    over = dict(mmdit_config=tiny_flux(), vae_config=tiny_vae(), vae_encoder_config=tiny_vae_encoder(), text_len=20)
    out = tmp_path / "a.png"
    img, log = cli.main(["--prompt", "a cat", "--steps", "3", "--seed", "7", "--height", "128", "--width", "192", "-o", str(out),
                         "--benchmark-mode"], pipeline_overrides=over)
    assert out.exists() and img.size == (192, 128) and len(log["denoising"]["iter_time"]) == 3
    img2, _ = cli.main(["--prompt", "a cat", "--steps", "3", "--seed", "7", "--height", "128", "--width", "192", "-o", str(out)],
                       pipeline_overrides=over)
    assert np.array_equal(np.asarray(img), np.asarray(img2))  # the warm-up run leaves no state behind
    out2 = tmp_path / "b.png"
    img3, log3 = cli.main(["--prompt", "a cat", "--steps", "4", "--seed", "7", "--height", "128", "--width", "192", "-o", str(out2),
                           "--image-path", str(out), "--denoise", "0.5"], pipeline_overrides=over)
    assert out2.exists() and len(log3["denoising"]["iter_time"]) == 2  # int(4 * (1 - 0.5)) steps dropped
    over_sd3 = dict(mmdit_config=tiny_sd3(), vae_config=tiny_vae(), text_len=20)
    img4, log4 = cli.main(["--prompt", "a cat", "--model-version", "argmaxinc/mlx-stable-diffusion-3-medium", "--steps", "2", "--seed", "1",
                           "--height", "64", "--width", "64", "-o", str(tmp_path / "c.png"), "--negative_prompt", "blurry"],
                          pipeline_overrides=over_sd3)
    assert img4.size == (64, 64) and len(log4["denoising"]["iter_time"]) == 2


def test_module_entry_point_prints_help():
    """``python -m diffusionkit_amd.cli --help`` works without a GPU (argument parsing happens before anything is loaded)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "diffusionkit_amd.cli", "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--model-version" in r.stdout and "--denoise" in r.stdout
