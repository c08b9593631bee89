"""Parity of the fused ResBlock unit at the reference's precision on time-major tensors (mb_resblock_pair_split,
resblock_pair_split.hip: fp32 [B][T][C] in HBM, lrelu / split by the support waves, error-compensated fp16 MFMA products, the
intermediate in LDS) against float64 ATen convs of the same fp32 weights and inputs.
Reference: one (convs1[i], convs2[i]) iteration of ResBlock1.forward, models/vocoder/hifigan/models.py:39-46,
models/vocoder/fregan/generator.py:43-50; out_scale / accumulate = the mean over ResBlocks, models.py:139-145.
Gate: max |delta| <= 1e-5 * max(1, output RMS) -- the fp32 reference's own rounding through two convs (the audio gate is 1e-4 RMS on
the final waveform)."""
import pytest
import torch
import torch.nn.functional as F

import hiputil

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _ref(x, w1, b1, w2, b2, d, slope, T_valid=None):
    """float64; positions >= T_valid[b] are the item's zero padding for both convs (ragged batches)."""
    B, _, T = x.shape
    mask = torch.ones(B, 1, T, dtype=torch.float64)
    if T_valid is not None:
        for b, tv in enumerate(T_valid):
            mask[b, :, tv:] = 0
    k = w1.shape[-1]
    xr = x.double() * mask
    h = F.conv1d(F.leaky_relu(xr, slope), w1.double(), b1.double(), padding=d * (k - 1) // 2, dilation=d)
    h = F.leaky_relu(h, slope) * mask
    return F.conv1d(h, w2.double(), b2.double(), padding=(k - 1) // 2) + xr


def _weights(C, k, seed=2, wscale=1.0):
    w1 = _rand(C, C, k, seed=seed) / (C * k) ** 0.5 * wscale
    w2 = _rand(C, C, k, seed=seed + 1) / (C * k) ** 0.5 * wscale
    return w1, 0.1 * _rand(C, seed=seed + 2), w2, 0.1 * _rand(C, seed=seed + 3)


CASES = [
    # (B, C, T, k, dil): every (C, tile) instance, ragged T, T shorter than a tile, Fre-GAN dilation 7
    (2, 256, 1000, 3, 1), (1, 256, 130, 11, 5), (1, 256, 300, 7, 7), (3, 256, 500, 7, 3),
    (2, 128, 333, 7, 3), (1, 128, 2500, 11, 5), (1, 128, 64, 3, 1),
    (1, 64, 700, 11, 7), (3, 64, 2000, 3, 5), (1, 64, 37, 7, 1),
    (3, 32, 1000, 3, 5), (1, 32, 4100, 11, 1), (1, 32, 9, 7, 3), (1, 32, 3000, 11, 5),
    (2, 16, 3000, 11, 7), (1, 16, 777, 3, 1), (1, 16, 5, 7, 5),  # Fre-GAN last stage: half an MFMA tile of channels
]


@pytest.mark.parametrize("B,C,T,k,d", CASES)
def test_resblock_pair_split_matches_float64(cuda, lib, B, C, T, k, d):
    x = _rand(B, C, T, seed=1)
    w1, b1, w2, b2 = _weights(C, k)
    y = hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d, slope=0.1)
    ref = _ref(x, w1, b1, w2, b2, d, 0.1)
    rms = float(ref.pow(2).mean().sqrt())
    err = float((y.double() - ref).abs().max())
    assert int(torch.isnan(y).sum()) == 0 and err <= 1e-5 * max(1.0, rms), (err, rms)


@pytest.mark.parametrize("tile", [1, 2])
@pytest.mark.parametrize("C,T,k,d", [(256, 700, 7, 3), (128, 900, 3, 5), (64, 600, 11, 1), (32, 1500, 7, 5)])
def test_both_tile_candidates(cuda, lib, monkeypatch, tile, C, T, k, d):
    """MBHIP_DIAG=spair_tile=<1|2> forces the first / second tile candidate of a channel count (the makespan estimate picks
    between them): both must give the same sums up to fp32 rounding of the tile-independent arithmetic (they are bit-identical:
    a position's sums do not depend on the tile it is in)."""
    x = _rand(2, C, T, seed=3)
    w1, b1, w2, b2 = _weights(C, k, seed=11)
    monkeypatch.setenv("MBHIP_DIAG", f"spair_tile={tile}")
    y = hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d)
    monkeypatch.delenv("MBHIP_DIAG")
    y0 = hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d)
    ref = _ref(x, w1, b1, w2, b2, d, 0.1)
    rms = float(ref.pow(2).mean().sqrt())
    assert float((y.double() - ref).abs().max()) <= 1e-5 * max(1.0, rms)
    assert torch.equal(y, y0)


@pytest.mark.parametrize("B,C,T,k,d", [(32, 128, 1100, 3, 1), (64, 32, 1100, 7, 3), (32, 256, 1000, 11, 5)])
def test_whole_rounds_on_the_larger_tile_rest_on_the_smaller(cuda, lib, monkeypatch, B, C, T, k, d):
    """A launch whose tiles leave the last round of persistent workgroups mostly empty is cut in two: rows [0, j NB_a) of every item on the
    larger tile (whole rounds), the rest on the smaller one (MBHIP_DIAG=spair_split=2 forces the cut wherever the shapes allow it,
    =0 forbids it).  Same sums per position: bit-identical to the single launch, ragged lengths and an accumulating launch included."""
    x = _rand(B, C, T, seed=5)
    w1, b1, w2, b2 = _weights(C, k, seed=21)
    valid = [T - 7 * (b % 5) for b in range(B)]
    acc = _rand(B, C, T, seed=6)
    outs = []
    for mode in ("2", "0"):
        monkeypatch.setenv("MBHIP_DIAG", f"spair_split={mode}")
        y = hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d)
        yr = hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d, valid=valid, out_scale=0.5, accumulate_into=acc)
        outs.append((y, yr))
    monkeypatch.delenv("MBHIP_DIAG")
    assert torch.equal(outs[0][0], outs[1][0])
    for b in range(B):
        assert torch.equal(outs[0][1][b, :, :valid[b]], outs[1][1][b, :, :valid[b]])
    ref = _ref(x[:4], w1, b1, w2, b2, d, 0.1)
    rms = float(ref.pow(2).mean().sqrt())
    assert float((outs[0][0][:4].double() - ref).abs().max()) <= 1e-5 * max(1.0, rms)


def test_scale_and_accumulate(cuda, lib):
    """Last unit of a ResBlock: y = acc + (x + conv2(...)) / num_kernels  (models.py:141-145)."""
    B, C, T, k, d = 2, 64, 900, 7, 5
    x = _rand(B, C, T, seed=1)
    w1, b1, w2, b2 = _weights(C, k)
    acc = _rand(B, C, T, seed=6)
    y = hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d, out_scale=1.0 / 3.0, accumulate_into=acc)
    ref = acc.double() + _ref(x, w1, b1, w2, b2, d, 0.1) / 3.0
    err = float((y.double() - ref).abs().max())
    assert int(torch.isnan(y).sum()) == 0 and err <= 1e-5 * max(1.0, float(ref.pow(2).mean().sqrt())), err


@pytest.mark.parametrize("C", [256, 64, 32])
def test_ragged_batch(cuda, lib, C):
    """Items shorter than the padded batch: positions beyond d_valid[b] * valid_mul read as zero padding and are not written."""
    T, k, d, mul = 1200, 7, 3, 5
    valid = [240, 101, 7]
    x = _rand(3, C, T, seed=4)
    w1, b1, w2, b2 = _weights(C, k, seed=21)
    y = hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d, valid=valid, valid_mul=mul)
    ref = _ref(x, w1, b1, w2, b2, d, 0.1, [v * mul for v in valid])
    rms = float(ref.pow(2).mean().sqrt())
    for b, v in enumerate(valid):
        n = v * mul
        assert float((y[b, :, :n].double() - ref[b, :, :n]).abs().max()) <= 1e-5 * max(1.0, rms)
        assert bool(torch.isnan(y[b, :, n:]).all())


def test_small_activations_and_tiny_weights(cuda, lib):
    """|x| ~ 1e-3 (the scaled residual keeps such operands at 22 bits) and weights of 1e-4 (the per-conv power-of-two scale)."""
    C, T, k, d = 128, 800, 7, 3
    x = _rand(1, C, T, seed=2) * 1e-3
    w1, b1, w2, b2 = _weights(C, k, seed=9)
    y = hiputil.resblock_pair_split_hip(x, w1, b1 * 1e-3, w2, b2 * 1e-3, dilation=d)
    ref = _ref(x, w1, b1 * 1e-3, w2, b2 * 1e-3, d, 0.1)
    rms = float(ref.pow(2).mean().sqrt())
    assert float((y.double() - ref).abs().max()) <= 2e-5 * rms, (float((y.double() - ref).abs().max()), rms)
    w1t, w2t = w1 * 1e-4, w2 * 1e-4
    x1 = _rand(1, C, T, seed=5)
    y = hiputil.resblock_pair_split_hip(x1, w1t, b1, w2t, b2, dilation=d)
    ref = _ref(x1, w1t, b1, w2t, b2, d, 0.1)
    assert float((y.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.pow(2).mean().sqrt()))


def test_range_counter(cuda, lib, monkeypatch):
    """MBHIP_CONV_RANGE_CHECK=1: values beyond fp16's range (the split clamps them) are counted, as conv1d.hip counts them."""
    from mockingbird_amd import _lib
    monkeypatch.setenv("MBHIP_CONV_RANGE_CHECK", "1")
    L = _lib.lib()
    C, T, k, d = 64, 300, 3, 1
    x = _rand(1, C, T, seed=2)
    w1, b1, w2, b2 = _weights(C, k)
    hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d)
    L.mb_conv1d_range_events(1)
    hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d)
    assert L.mb_conv1d_range_events(1) == 0
    x[0, 5, 100] = 1e6
    hiputil.resblock_pair_split_hip(x, w1, b1, w2, b2, dilation=d)
    assert L.mb_conv1d_range_events(1) > 0


def test_layout_round_trip(cuda, lib):
    x = _rand(3, 80, 333, seed=8)
    t = hiputil.f32_cm_to_tm(x)
    assert torch.equal(t.cpu(), x.transpose(1, 2).contiguous())
    assert torch.equal(hiputil.f32_tm_to_cm(t).cpu(), x)


def test_rejects_bad_shapes(cuda, lib):
    from mockingbird_amd._lib import MbHipError
    x = _rand(1, 48, 100, seed=1)
    w = _rand(48, 48, 3, seed=2)
    with pytest.raises(MbHipError, match="unsupported"):
        hiputil.resblock_pair_split_hip(x, w, torch.zeros(48), w, torch.zeros(48))
