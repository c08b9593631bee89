"""Parity of the one-launch ResBlock group of a narrow GAN stage (mb_resblock_stage_f16: every unit of every parallel ResBlock
on LDS-resident activations, x read once, the mean written once) against ATen CPU convs on the SAME fp16-rounded operands,
with the activations rounded to fp16 exactly where the kernel rounds them (lrelu(x), h after bias + lrelu, x after every unit,
the running mean after every ResBlock).
Reference: Generator.forward's loop over self.resblocks, models/vocoder/hifigan/models.py:139-145,
models/vocoder/fregan/generator.py:150-157; ResBlock1.forward hifigan/models.py:39-46, fregan/generator.py:43-50.
Gate: |delta| <= 2e-3 + 2^-8 * max(1, |ref|) (a few fp16 roundings of the residual stream)."""
import pytest
import torch
import torch.nn.functional as F

import hiputil

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _h(t):
    return t.half().float()


def _make(C, ks, dils, seed=0):
    chains = []
    for j, k in enumerate(ks):
        units = []
        for u, d in enumerate(dils[j]):
            s = seed + 100 * j + 10 * u
            units.append((_rand(C, C, k, seed=s + 1) / (C * k) ** 0.5, 0.1 * _rand(C, seed=s + 2),
                          _rand(C, C, k, seed=s + 3) / (C * k) ** 0.5, 0.1 * _rand(C, seed=s + 4), d))
        chains.append(units)
    return chains


def _ref(x, chains, slope, T_valid=None):
    """x [B, C, T]; positions >= T_valid[b] are zero padding for every conv (ragged batches)."""
    B, _, T = x.shape
    mask = torch.ones(B, 1, T)
    if T_valid is not None:
        for b, tv in enumerate(T_valid):
            mask[b, :, tv:] = 0
    out = None
    for units in chains:
        xr = _h(x) * mask
        for (w1, b1, w2, b2, d) in units:
            k = w1.shape[-1]
            xt = _h(F.leaky_relu(xr, slope)) * mask
            h = F.conv1d(xt, _h(w1), b1, padding=d * (k - 1) // 2, dilation=d)
            h = _h(F.leaky_relu(_h(h), slope)) * mask
            xr = _h(F.conv1d(h, _h(w2), b2, padding=(k - 1) // 2) + xr)
        o = xr / len(chains)
        out = _h(o) if out is None else _h(out + o)
    return out


CASES = [
    # (B, C, T, kernel sizes, dilations per ResBlock): HiFi-GAN V1 and Fre-GAN groups, tiles shorter / longer than T, one chain
    (2, 32, 1500, (3, 7, 11), ((1, 3, 5),) * 3),
    (1, 32, 100, (3, 7, 11), ((1, 3, 5),) * 3),
    (1, 32, 520 * 3, (3, 7, 11), ((1, 3, 5),) * 3),
    (1, 32, 2000, (3, 7, 11), ((1, 3, 5, 7),) * 3),
    (2, 16, 2500, (3, 7, 11), ((1, 3, 5, 7),) * 3),
    (1, 16, 77, (3, 7, 11), ((1, 3, 5),) * 3),
    (1, 32, 700, (7,), ((1, 2),)),
    (1, 32, 900, (5, 3), ((2, 1, 1), (1, 1, 4))),
    # the wider stages' single ResBlocks (two 32-row output tiles per wave; at 128 channels two waves along the channels)
    (2, 64, 1000, (3,), ((1, 3, 5),)), (1, 64, 700, (7,), ((1, 3, 5),)), (1, 64, 200, (3,), ((1, 3, 5, 7),)),
    (2, 128, 500, (3,), ((1, 3, 5),)), (1, 128, 90, (3,), ((1, 3, 5),)),
]


@pytest.mark.parametrize("B,C,T,ks,dils", CASES)
def test_resblock_stage_matches_aten(cuda, lib, B, C, T, ks, dils):
    x = _rand(B, C, T, seed=1)
    chains = _make(C, ks, dils, seed=7)
    y = hiputil.resblock_stage_f16_hip(x, chains, slope=0.1)
    ref = _ref(x, chains, 0.1)
    dlt = (y.double() - ref.double()).abs()
    tol = 2e-3 + 2.0 ** -8 * ref.double().abs().clamp(min=1.0)
    assert int(torch.isnan(y).sum()) == 0 and int((dlt > tol).sum()) == 0, float(dlt.max())


def test_resblock_stage_ragged_batch(cuda, lib):
    """Items shorter than the padded batch: positions beyond an item's length are zero padding for every conv of every unit
    and are not stored (gan.hip passes frames[b] x samples-per-frame)."""
    B, C, T = 3, 32, 1200
    valid = [6, 3, 5]          # x 200 positions
    x = _rand(B, C, T, seed=2)
    chains = _make(C, (3, 7, 11), ((1, 3, 5),) * 3, seed=9)
    y = hiputil.resblock_stage_f16_hip(x, chains, slope=0.1, valid=valid, valid_mul=200)
    tv = [v * 200 for v in valid]
    ref = _ref(x, chains, 0.1, T_valid=tv)
    for b in range(B):
        yb, rb = y[b, :, :tv[b]], ref[b, :, :tv[b]]
        dlt = (yb.double() - rb.double()).abs()
        tol = 2e-3 + 2.0 ** -8 * rb.double().abs().clamp(min=1.0)
        assert int(torch.isnan(yb).sum()) == 0 and int((dlt > tol).sum()) == 0, (b, float(dlt.max()))
        assert bool(torch.isnan(y[b, :, tv[b]:]).all())  # untouched


def test_resblock_stage_equals_unit_launches(cuda, lib):
    """The same group through nine mb_resblock_pair_f16 launches (the path of the wide stages): same result up to the
    roundings the two paths place differently (one rounding of the residual per unit instead of two)."""
    B, C, T = 1, 32, 3000
    x = _rand(B, C, T, seed=3)
    chains = _make(C, (3, 7, 11), ((1, 3, 5),) * 3, seed=11)
    y = hiputil.resblock_stage_f16_hip(x, chains, slope=0.1)
    acc = None
    for units in chains:
        xr = x
        for i, (w1, b1, w2, b2, d) in enumerate(units):
            last = i == len(units) - 1
            if last:
                acc = hiputil.resblock_pair_f16_hip(xr, w1, b1, w2, b2, dilation=d, out_scale=1.0 / 3.0, accumulate_into=acc) \
                    if acc is not None else hiputil.resblock_pair_f16_hip(xr, w1, b1, w2, b2, dilation=d, out_scale=1.0 / 3.0)
            else:
                xr = hiputil.resblock_pair_f16_hip(xr, w1, b1, w2, b2, dilation=d)
    dlt = (y.double() - acc.double()).abs()
    tol = 3e-3 + 2.0 ** -8 * acc.double().abs().clamp(min=1.0)
    assert int((dlt > tol).sum()) == 0, float(dlt.max())


def test_resblock_stage_accumulates_into_the_stage_output(cuda, lib):
    """A wider stage runs one launch per ResBlock: y = y + ResBlock_j(x) / num_kernels (models.py:141-145)."""
    B, C, T = 2, 64, 900
    x = _rand(B, C, T, seed=4)
    chains = _make(C, (7,), ((1, 3, 5),), seed=13)
    acc = _rand(B, C, T, seed=5)
    y = hiputil.resblock_stage_f16_hip(x, chains, slope=0.1, out_scale=1.0 / 3.0, accumulate_into=acc)
    ref = _h(_h(acc) + _h(_ref(x, chains, 0.1) * 1.0 / 3.0 * 1.0))  # _ref divides by len(chains) = 1: scale 1/3 here
    dlt = (y.double() - ref.double()).abs()
    tol = 2e-3 + 2.0 ** -8 * ref.double().abs().clamp(min=1.0)
    assert int(torch.isnan(y).sum()) == 0 and int((dlt > tol).sum()) == 0, float(dlt.max())


def test_resblock_stage_rejects_bad_shapes(cuda, lib):
    from mockingbird_amd._lib import MbHipError
    x = _rand(1, 48, 100, seed=1)
    with pytest.raises(MbHipError, match="unsupported"):
        hiputil.resblock_stage_f16_hip(x, _make(48, (3,), ((1,),)))
    x = _rand(1, 128, 300, seed=1)
    with pytest.raises(MbHipError, match="unsupported"):  # the reach of k = 11 leaves no rows in a 128-row window
        hiputil.resblock_stage_f16_hip(x, _make(128, (11,), ((1, 3, 5),)))
