"""Scalar known-answer tests: schedule / noise / latent-format helpers of both the product host
code (diffusionkit_amd.sampler, pipeline helpers) and the oracle restatement against values
derived from the reference's formulas (tests/golden/kat_scalars.json)."""
import json
import os

import numpy as np
import pytest
import torch

from diffusionkit_amd import sampler as S
from oracle import pipeline as op

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_scalars.json")))


def test_flux_sigmas_and_bf16_timesteps():
    sig = S.get_sigmas(S.FluxSampler(shift=1.0), 4)
    assert np.allclose(sig, KAT["flux_sigmas_n4_shift1"], atol=0)
    assert np.allclose(op.get_sigmas(1.0, True, 4).numpy(), KAT["flux_sigmas_n4_shift1"], atol=0)
    ts = torch.tensor(sig * 1000).to(torch.bfloat16).float().tolist()
    assert ts == KAT["flux_timesteps_bf16_n4"] == [1000.0, 752.0, 500.0, 250.0, 0.0]  # quirk Q1


def test_sd3_sigmas_shift_applied_twice():
    smp = S.ModelSamplingDiscreteFlow(shift=3.0)
    assert abs(float(smp.sigma_min) - KAT["sd3_sigma_min_shift3"]) < 1e-9
    assert abs(float(smp.sigma_max) - 1.0) < 1e-7
    for impl in (S.get_sigmas(smp, 4), op.get_sigmas(3.0, False, 4).numpy()):
        assert np.allclose(impl, KAT["sd3_sigmas_n4_shift3"], atol=2e-7)
        assert np.allclose(impl, [1, 0.85769236, 0.60215056, 0.00892857, 0], atol=1e-6)
    ts16 = torch.tensor(S.get_sigmas(smp, 4) * 1000).to(torch.float16).float().tolist()
    assert ts16 == KAT["sd3_timesteps_fp16_n4"]
    s50 = S.get_sigmas(smp, 50)
    assert len(s50) == 51
    assert np.allclose(s50[:6], KAT["sd3_sigmas_n50_shift3_first6"], atol=2e-7)
    assert np.allclose(s50[-3:], KAT["sd3_sigmas_n50_shift3_last3"], atol=2e-7)
    assert np.allclose(op.get_sigmas(3.0, False, 50).numpy(), s50, atol=2e-7)


def test_flux_sampler_table_endpoints():
    f = S.FluxSampler(shift=1.0)
    assert float(f.sigma_min) == 0.0 and float(f.sigma_max) == 1.0
    assert len(S.get_sigmas(f, 50)) == 51
    assert S.max_denoise(f, S.get_sigmas(f, 4))


def test_noise_seed_layout():
    n = op.get_noise(0, 4, 4)
    assert n.shape == (1, 4, 4, 16)
    assert np.allclose(n[0, 0, 0, :4].numpy(), KAT["noise_seed0_nhwc_0_0_0_first4"], atol=1e-6)
    assert np.allclose(n.permute(0, 3, 1, 2)[0, 0, 0, :4].numpy(), KAT["noise_seed0_nchw_0_0_0_first4"], atol=1e-6)


def test_latent_formats_and_empty_latent():
    x = torch.tensor([1.0, -2.0])
    for name, (scale, shift) in KAT["latent_format"].items():
        assert torch.allclose(op.process_out(x, name), x / scale + shift)
    assert float(op.get_empty_latent(2, 2).flatten()[0]) == pytest.approx(KAT["empty_latent_value"])


def test_psnr_definition():
    ex = KAT["psnr_example"]
    assert op.compute_psnr(np.array(ex["ref"]), np.array(ex["proxy"])) == pytest.approx(ex["value"], rel=1e-9)


def test_noise_scaling_and_denoised():
    f = S.FluxSampler()
    assert f.noise_scaling(np.float32(1.0), np.float32(2.0), np.float32(0.0609)) == pytest.approx(2.0)
    assert f.calculate_denoised(0.5, 2.0, 3.0) == pytest.approx(2.0)
