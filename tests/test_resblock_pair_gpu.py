"""Parity of the fused fp16 ResBlock unit (mb_resblock_pair_f16: conv1 dilated -> leaky_relu -> conv2
-> + x in one launch, intermediate in LDS) against ATen CPU convs on the SAME fp16-rounded operands,
with the intermediate rounded to fp16 exactly where the kernel rounds it (after bias + leaky_relu).
Reference: ResBlock1.forward, models/vocoder/hifigan/models.py:39-46, fregan/generator.py:43-50.
Gate: |delta| <= 1e-3 + 2^-9 * max(1, |ref|) (fp16 rounding of the stored result and of h)."""
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


def _ref(x, w1, b1, w2, b2, d, slope):
    k = w1.shape[-1]
    xt = F.leaky_relu(_h(x), slope)
    h = F.conv1d(_h(xt), _h(w1), b1, padding=d * (k - 1) // 2, dilation=d)
    h = _h(F.leaky_relu(h, slope))
    return F.conv1d(h, _h(w2), b2, padding=(k - 1) // 2) + _h(x)


CASES = [
    # (B, C, T, k, dil): every (C, tile) instance, ragged T, T shorter than a tile, Fre-GAN dilation 7
    (2, 256, 1000, 3, 1), (1, 256, 130, 11, 5), (1, 256, 300, 7, 7),
    (32, 256, 1000, 7, 3),  # the bench shape of the first stage: 160-row tiles win the makespan estimate (<256, 5, 1>)
    (2, 128, 333, 7, 3), (1, 128, 5000, 11, 5), (1, 128, 64, 3, 1),
    (1, 64, 700, 11, 7), (3, 64, 2000, 3, 5), (1, 64, 37, 7, 1),
    (3, 32, 1000, 3, 5), (1, 32, 4100, 11, 1), (1, 32, 9, 7, 3),
    (2, 16, 3000, 11, 7), (1, 16, 777, 3, 1), (1, 16, 5, 7, 5),  # Fre-GAN last stage: half an MFMA tile of channels
]


@pytest.mark.parametrize("B,C,T,k,d", CASES)
def test_resblock_pair_matches_aten(cuda, lib, B, C, T, k, d):
    x = _rand(B, C, T, seed=1)
    w1 = _rand(C, C, k, seed=2) / (C * k) ** 0.5
    w2 = _rand(C, C, k, seed=3) / (C * k) ** 0.5
    b1, b2 = 0.1 * _rand(C, seed=4), 0.1 * _rand(C, seed=5)
    y = hiputil.resblock_pair_f16_hip(x, w1, b1, w2, b2, dilation=d, slope=0.1)
    ref = _ref(x, w1, b1, w2, b2, d, 0.1)
    e = hiputil.relerr(y, ref)
    dlt = (y.double() - ref.double()).abs()
    tol = 1e-3 + 2.0 ** -9 * ref.double().abs().clamp(min=1.0)
    assert e["nan"] == 0 and int((dlt > tol).sum()) == 0, (e, float(dlt.max()))


def test_resblock_pair_scale_and_accumulate(cuda, lib):
    """Last unit of a ResBlock: y = acc + (x + conv2(...)) / num_kernels  (models.py:141-145)."""
    B, C, T, k, d = 2, 64, 900, 7, 5
    x = _rand(B, C, T, seed=1)
    w1 = _rand(C, C, k, seed=2) / (C * k) ** 0.5
    w2 = _rand(C, C, k, seed=3) / (C * k) ** 0.5
    b1, b2 = 0.1 * _rand(C, seed=4), 0.1 * _rand(C, seed=5)
    acc = _rand(B, C, T, seed=6)
    y = hiputil.resblock_pair_f16_hip(x, w1, b1, w2, b2, dilation=d, out_scale=1.0 / 3.0, accumulate_into=acc)
    ref = _h(acc) + _ref(x, w1, b1, w2, b2, d, 0.1) / 3.0
    dlt = (y.double() - ref.double()).abs()
    tol = 1e-3 + 2.0 ** -9 * ref.double().abs().clamp(min=1.0)
    assert int(torch.isnan(y).sum()) == 0 and int((dlt > tol).sum()) == 0, float(dlt.max())


def test_resblock_pair_rejects_bad_shapes(cuda, lib):
    from mockingbird_amd._lib import MbHipError
    x = _rand(1, 48, 100, seed=1)
    w = _rand(48, 48, 3, seed=2)
    with pytest.raises(MbHipError, match="unsupported"):
        hiputil.resblock_pair_f16_hip(x, w, torch.zeros(48), w, torch.zeros(48))


@pytest.mark.parametrize("C", [32, 64, 128])
def test_bias_block_of_the_previous_launch_is_never_used(cuda, lib, C):
    """The kernel stages (b1, b2) in LDS and starts its accumulators from them.  Alternate launches with very different
    biases: a first tile that read the LDS before the fill was fenced (a race this kernel once had: one tile of a launch now
    and then took the previous launch's bias block) shows up as an error of the bias difference."""
    T, k, d = 3000, 7, 3
    x = _rand(1, C, T, seed=1)
    w1 = _rand(C, C, k, seed=2) / (C * k) ** 0.5
    w2 = _rand(C, C, k, seed=3) / (C * k) ** 0.5
    biases = [(0.1 * _rand(C, seed=4) + 3.0, 0.1 * _rand(C, seed=5) - 2.0), (0.1 * _rand(C, seed=6) - 3.0, 0.1 * _rand(C, seed=7) + 2.0)]
    refs = [_ref(x, w1, b1, w2, b2, d, 0.1) for b1, b2 in biases]
    for rep in range(6):
        b1, b2 = biases[rep & 1]
        y = hiputil.resblock_pair_f16_hip(x, w1, b1, w2, b2, dilation=d, slope=0.1)
        dlt = (y.double() - refs[rep & 1].double()).abs()
        tol = 4e-3 + 2.0 ** -8 * refs[rep & 1].double().abs().clamp(min=1.0)
        assert int((dlt > tol).sum()) == 0, (rep, float(dlt.max()))
