"""Parity of the VITS decoder (models/synthesizer/models/vits.py:245-298 Generator.forward) on the GAN
kernels: mb_gan_forward_ex with the cond(g) channel bias, against the oracle restatement and the golden
outputs of the reference module itself (tests/golden/vits.npz).  fp32 gate: audio RMS <= 1e-4 and relative
RMS <= 1e-3 (as for HiFi-GAN); fp16 gate: relative RMS <= 5e-3."""
import os

import numpy as np
import pytest
import torch

import hiputil
import synth
from oracle import gan as og

pytestmark = pytest.mark.gpu


def _make(uic, seed, dtype):
    from mockingbird_amd.vits import VitsGenerator
    h = dict(synth.VITS_DEC)
    h["upsample_initial_channel"] = uic
    st = synth.vits_dec_state(h, seed=seed)
    gen = VitsGenerator(st, h["initial_channel"], h["resblock"], h["resblock_kernel_sizes"], h["resblock_dilation_sizes"],
                        h["upsample_rates"], uic, h["upsample_kernel_sizes"], gin_channels=h["gin_channels"], dtype=dtype)
    return gen, h, st


@pytest.mark.parametrize("case", synth.VITS_CASES, ids=lambda c: c[0])
def test_vits_decoder_matches_reference_golden(cuda, lib, case):
    name, uic, frames, batch, use_g, seed = case
    gold = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", "vits.npz"))[name])
    gen, h, st = _make(uic, seed, "f32")
    z, spk = synth.vits_latent(frames, batch, seed=seed + 1)
    y = gen(torch.from_numpy(z).cuda(), torch.from_numpy(spk).cuda() if use_g else None).cpu()
    assert y.shape == gold.shape == (batch, 1, frames * 256)
    e = hiputil.relerr(y, gold)
    assert e["nan"] == 0 and e["rms"] <= 1e-4 and e["rel_rms"] <= 1e-3, e


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-3), ("f16", 5e-3)])
def test_vits_decoder_full_width_vs_oracle(cuda, lib, dtype, tol):
    """Full 512-channel decoder, 37 latent frames, batch 3 with three different speakers (the per-utterance
    cond bias), fp32 and fp16 paths against the oracle."""
    gen, h, st = _make(512, 7, dtype)
    z, spk = synth.vits_latent(37, 3, seed=9)
    y = gen(torch.from_numpy(z).cuda(), torch.from_numpy(spk).cuda()).cpu()
    with torch.no_grad():
        ref = og.vits_generator_forward(og.fold_weight_norm_state(st), h, torch.from_numpy(z), torch.from_numpy(spk))
    e = hiputil.relerr(y, ref)
    print("vits", dtype, e)
    assert e["nan"] == 0 and e["rel_rms"] <= tol, e
    # the speaker embedding matters: a different g changes the output
    y2 = gen(torch.from_numpy(z).cuda(), torch.from_numpy(spk[[1, 2, 0]].copy()).cuda()).cpu()
    assert float((y2 - y).abs().max()) > 1e-3
