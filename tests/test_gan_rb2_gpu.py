"""h.resblock == '2': ResBlock2 generators (models/vocoder/hifigan/models.py:51-72,100: two units x <- x + conv_d(lrelu(x)) per block) and
the VITS decoder built with resblock='2' (models/synthesizer/models/vits.py:251, sublayer/vits_modules.py:225-249) against the outputs
of the reference modules themselves (tests/golden/gan_rb2.npz, tests/golden/make_golden.py gan_rb2) and the oracle at full width.
Gates as tests/test_gan_gpu.py: fp32 audio RMS <= 1e-4 and relative RMS <= 1e-3; fp16 relative RMS <= 5e-3."""
import os

import numpy as np
import pytest
import torch

import hiputil
import synth
from oracle import gan as og

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "gan_rb2.npz")


def _hifigan(h, frames, batch, seed, dtype):
    from mockingbird_amd.vocoder.gan import GanGenerator
    st = synth.gan_state(h, "hifigan", seed=seed)
    gen = GanGenerator(h, st["generator"], 0, dtype=dtype)
    mel = torch.from_numpy(synth.mel_input(frames, batch, seed=seed + 1))
    y = gen(mel.cuda())
    torch.cuda.synchronize()
    return y.cpu(), st, mel


@pytest.mark.parametrize("dtype,rms_tol,rel_tol", [("f32", 1e-4, 1e-3), ("f16", 1e-2, 5e-3)])
@pytest.mark.parametrize("uic,frames,batch,seed", synth.GAN_RB2_CASES)
def test_hifigan_resblock2_matches_reference_golden(cuda, lib, uic, frames, batch, seed, dtype, rms_tol, rel_tol):
    h = synth.small(synth.HIFIGAN_RB2, uic)
    y, _, _ = _hifigan(h, frames, batch, seed, dtype)
    gold = torch.from_numpy(np.load(GOLD)[f"hifigan_rb2_uic{uic}_f{frames}_b{batch}_s{seed}"])
    assert y.shape == gold.shape
    e = hiputil.relerr(y, gold)
    assert e["nan"] == 0 and e["rms"] <= rms_tol and e["rel_rms"] <= rel_tol, e


@pytest.mark.parametrize("fuse", ["all", "cm", "none"])
def test_hifigan_resblock2_plans_vs_oracle(cuda, lib, monkeypatch, fuse):
    """The time-major plan (default), the channel-major plan and one launch per conv are the same generator."""
    monkeypatch.setenv("MBHIP_GAN_FUSE", fuse)
    h = synth.small(synth.HIFIGAN_RB2, 128)
    y, st, mel = _hifigan(h, 23, 2, 11, "f32")
    with torch.no_grad():
        ref = og.hifigan_forward(og.fold_weight_norm_state(st["generator"]), h, mel)
    e = hiputil.relerr(y, ref)
    assert e["nan"] == 0 and e["rms"] <= 1e-4 and e["rel_rms"] <= 1e-3, e


@pytest.mark.parametrize("dtype,rms_tol,rel_tol", [("f32", 1e-4, 1e-3), ("f16", 1e-2, 5e-3)])
@pytest.mark.parametrize("case", synth.VITS_RB2_CASES)
def test_vits_decoder_resblock2_matches_reference_golden(cuda, lib, case, dtype, rms_tol, rel_tol):
    from mockingbird_amd.vits import VitsGenerator
    name, uic, frames, batch, use_g, seed = case
    if dtype == "f16" and uic >> 4 < 8:
        pytest.skip("the fp16 storage path needs channel counts that are multiples of 8 (the last stage has uic / 16)")
    h = dict(synth.VITS_DEC_RB2)
    h["upsample_initial_channel"] = uic
    st = synth.vits_dec_state(h, seed=seed)
    gen = VitsGenerator(st, h["initial_channel"], h["resblock"], h["resblock_kernel_sizes"], h["resblock_dilation_sizes"],
                        h["upsample_rates"], uic, h["upsample_kernel_sizes"], gin_channels=h["gin_channels"], dtype=dtype)
    z, spk = synth.vits_latent(frames, batch, seed=seed + 1)
    y = gen(torch.from_numpy(z).cuda(), torch.from_numpy(spk).cuda() if use_g else None).cpu()
    gold = torch.from_numpy(np.load(GOLD)["vits_" + name])
    assert y.shape == gold.shape
    e = hiputil.relerr(y, gold)
    assert e["nan"] == 0 and e["rms"] <= rms_tol and e["rel_rms"] <= rel_tol, e
