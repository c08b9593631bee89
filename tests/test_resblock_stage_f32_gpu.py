"""Parity of the fused ResBlock group at the reference's precision (mb_resblock_stage_f32, resblock_stage_f32.hip: fp32 tensors, error-
compensated fp16 MFMA products on LDS-resident hi / lo operands, fp32 residual chain in registers) against float64 ATen convs of the same
fp32 weights and inputs.
Reference: Generator.forward's loop over self.resblocks, models/vocoder/hifigan/models.py:139-145, models/vocoder/fregan/generator.py:150-157;
ResBlock1.forward hifigan/models.py:39-46, fregan/generator.py:43-50.
Gate: max |delta| <= 2e-5 * max(1, output RMS) -- the size of the fp32 reference's own rounding through 6-8 convs (the audio gate is
1e-4 RMS on the final waveform)."""
import pytest
import torch
import torch.nn.functional as F

import hiputil

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _make(C, ks, dils, seed=0, wscale=1.0):
    chains = []
    for j, k in enumerate(ks):
        units = []
        for u, d in enumerate(dils[j]):
            s = seed + 100 * j + 10 * u
            units.append((_rand(C, C, k, seed=s + 1) / (C * k) ** 0.5 * wscale, 0.1 * _rand(C, seed=s + 2),
                          _rand(C, C, k, seed=s + 3) / (C * k) ** 0.5 * wscale, 0.1 * _rand(C, seed=s + 4), d))
        chains.append(units)
    return chains


def _ref(x, chains, slope, T_valid=None, out_scale=None):
    """float64; x [B, C, T]; positions >= T_valid[b] are zero padding for every conv (ragged batches)."""
    B, _, T = x.shape
    mask = torch.ones(B, 1, T, dtype=torch.float64)
    if T_valid is not None:
        for b, tv in enumerate(T_valid):
            mask[b, :, tv:] = 0
    out = 0
    for units in chains:
        xr = x.double() * mask
        for (w1, b1, w2, b2, d) in units:
            k = w1.shape[-1]
            h = F.conv1d(F.leaky_relu(xr, slope) * mask, w1.double(), b1.double(), padding=d * (k - 1) // 2, dilation=d)
            h = F.leaky_relu(h, slope) * mask
            xr = F.conv1d(h, w2.double(), b2.double(), padding=(k - 1) // 2) + xr
        out = out + xr * (1.0 / len(chains) if out_scale is None else out_scale)
    return out


CASES = [
    # (B, C, T, kernel sizes, dilations per ResBlock)
    (2, 32, 1500, (3, 7, 11), ((1, 3, 5),) * 3),      # HiFi-GAN V1's 32-channel group, several tiles per item
    (1, 32, 100, (3, 7, 11), ((1, 3, 5),) * 3),       # shorter than one tile
    (1, 32, 136 * 3, (3, 7, 11), ((1, 3, 5),) * 3),   # exactly three tiles
    (1, 32, 2000, (3, 7, 11), ((1, 3, 5, 7),) * 3),   # Fre-GAN's four dilations
    (1, 32, 700, (7,), ((1, 2),)), (1, 32, 900, (5, 3), ((2, 1, 1), (1, 1, 4))),
    (2, 64, 1000, (3,), ((1, 3, 5),)), (1, 64, 700, (7,), ((1, 3, 5),)),   # 64 channels: a ResBlock per launch
    (1, 64, 900, (11,), ((5,),)), (2, 64, 333, (11,), ((1,),)),          # ... and a single unit (what k = 11 runs as)
]


@pytest.mark.parametrize("B,C,T,ks,dils", CASES)
def test_resblock_stage_f32_matches_float64(cuda, lib, B, C, T, ks, dils):
    x = _rand(B, C, T, seed=1)
    chains = _make(C, ks, dils, seed=7)
    y = hiputil.resblock_stage_f32_hip(x, chains, slope=0.1)
    ref = _ref(x, chains, 0.1)
    rms = float(ref.pow(2).mean().sqrt())
    err = float((y.double() - ref).abs().max())
    assert int(torch.isnan(y).sum()) == 0 and err <= 2e-5 * max(1.0, rms), (err, rms)


def test_resblock_stage_f32_small_activations_and_tiny_weights(cuda, lib):
    """|x| ~ 1e-3 (the scaled residual keeps such operands at 22 bits) and weights of 1e-4 (the per-conv power-of-two scale)."""
    x = _rand(1, 32, 800, seed=2) * 1e-3
    chains = _make(32, (3, 7, 11), ((1, 3, 5),) * 3, seed=9)
    chains = [[(w1, b1 * 1e-3, w2, b2 * 1e-3, d) for (w1, b1, w2, b2, d) in units] for units in chains]
    y = hiputil.resblock_stage_f32_hip(x, chains, slope=0.1)
    ref = _ref(x, chains, 0.1)
    rms = float(ref.pow(2).mean().sqrt())
    assert float((y.double() - ref).abs().max()) <= 2e-5 * rms, (float((y.double() - ref).abs().max()), rms)
    chains = _make(64, (3,), ((1, 3, 5),), seed=5, wscale=1e-4)
    x = _rand(1, 64, 500, seed=3)
    y = hiputil.resblock_stage_f32_hip(x, chains, slope=0.1)
    ref = _ref(x, chains, 0.1)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.pow(2).mean().sqrt()))


def test_resblock_stage_f32_ragged_accumulate_and_scale(cuda, lib):
    """Items shorter than the padded batch (positions beyond an item's length are zero padding for every conv and are not stored), the
    accumulate flag (y += result: the mean over the ResBlocks of a 64-channel stage) and an explicit out_scale."""
    B, C, T = 3, 32, 1200
    valid = [6, 3, 5]          # x 200 positions
    x = _rand(B, C, T, seed=2)
    chains = _make(C, (3, 7, 11), ((1, 3, 5),) * 3, seed=9)
    y = hiputil.resblock_stage_f32_hip(x, chains, slope=0.1, valid=valid, valid_mul=200)
    tv = [v * 200 for v in valid]
    ref = _ref(x, chains, 0.1, T_valid=tv)
    for b in range(B):
        yb, rb = y[b, :, :tv[b]], ref[b, :, :tv[b]]
        assert int(torch.isnan(yb).sum()) == 0 and float((yb.double() - rb).abs().max()) <= 2e-5 * max(1.0, float(rb.pow(2).mean().sqrt())), b
        assert bool(torch.isnan(y[b, :, tv[b]:]).all())  # untouched
    chains = _make(64, (7,), ((1, 3, 5),), seed=13)
    x = _rand(2, 64, 450, seed=4)
    base = _rand(2, 64, 450, seed=5)
    y = hiputil.resblock_stage_f32_hip(x, chains, slope=0.1, out_scale=1.0 / 3.0, accumulate_into=base)
    ref = base.double() + _ref(x, chains, 0.1, out_scale=1.0 / 3.0)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.pow(2).mean().sqrt()))


def test_generator_uses_the_fused_stages_and_matches_the_per_conv_path(cuda, lib, monkeypatch):
    """GanGenerator (fp32) with the fused stages (default) against MBHIP_GAN_FUSE=units (one conv1d_split_kernel launch per conv): the
    same generator, the same weights -- RMS of the difference <= 2e-6 of the waveform's RMS (both are fp32-grade)."""
    import numpy as np
    import synth
    from mockingbird_amd.vocoder.gan import GanGenerator
    h = synth.HIFIGAN_16K
    st = synth.gan_state(h, "hifigan", seed=3)["generator"]
    mel = torch.from_numpy(synth.mel_input(37, 2, seed=1)).cuda()
    monkeypatch.delenv("MBHIP_GAN_FUSE", raising=False)
    y = GanGenerator(h, st, 0)(mel).cpu().double()
    monkeypatch.setenv("MBHIP_GAN_FUSE", "units")
    y0 = GanGenerator(h, st, 0)(mel).cpu().double()
    assert y.shape == y0.shape and int(torch.isnan(y).sum()) == 0
    rms = float(y0.pow(2).mean().sqrt())
    assert float((y - y0).pow(2).mean().sqrt()) <= 2e-6 * max(rms, 1e-3), (float((y - y0).pow(2).mean().sqrt()), rms)
