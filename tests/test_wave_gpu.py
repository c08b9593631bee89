"""Waveform wire format on device (mb_wave_peak_normalize / mb_wave_pack_pcm16) vs the numpy oracle.
Integer outputs are gated bit-exact; the normalised float waveform is gated bit-exact too (two IEEE
operations per sample in the array's own type)."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import wave as owv

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _t(x):
    return torch.from_numpy(x).cuda()


@pytest.mark.parametrize("case", synth.WAVE_CASES)
def test_pack_pcm16_matches_oracle_and_golden(cuda, lib, case):
    from mockingbird_amd.vocoder import wave
    dtype, n, peak, seed = case
    x = synth.wave_input(dtype, n, peak, seed)
    gold = np.load(os.path.join(G, "wave.npz"))
    key = f"{dtype}_n{n}_s{seed}"
    enc = wave.pack_pcm16(_t(x), "encode_16bits").cpu().numpy()
    sav = wave.pack_pcm16(_t(x), "save_wav").cpu().numpy()
    snd = wave.pack_pcm16(_t(x), "sndfile").cpu().numpy()
    assert enc.dtype == np.int16 and enc.shape == x.shape
    assert np.array_equal(enc, gold["encode16_" + key]) and np.array_equal(enc, owv.encode_16bits(x))
    assert np.array_equal(sav, gold["savewav_" + key]) and np.array_equal(sav, owv.save_wav_pcm(x))
    assert np.array_equal(snd, owv.sndfile_pcm16(x))


@pytest.mark.parametrize("case", synth.WAVE_CASES)
def test_peak_normalize_matches_oracle(cuda, lib, case):
    from mockingbird_amd.vocoder import wave
    dtype, n, peak, seed = case
    x = synth.wave_input(dtype, n, peak, seed)
    y = wave.peak_normalize_(_t(x.copy())).cpu().numpy()
    ref = owv.peak_normalize(x)
    assert y.dtype == ref.dtype == x.dtype
    assert np.array_equal(y, ref)
    assert abs(float(np.abs(y).max()) - 0.97) < 1e-6


def test_full_size_properties(cuda, lib):
    """BASELINE-size waveform (batch 32 x 200 frames x hop 200 = 1.28 M samples): pack(normalise(x)) is
    idempotent under a second normalise, odd (pack(-x) == -pack(x) away from the clip rails, round-half-even
    and truncation are both odd), and monotone."""
    from mockingbird_amd.vocoder import wave
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.tanh(torch.randn(32 * 200 * 200, device="cuda", generator=g) * 0.7)
    y = wave.peak_normalize_(x.clone())
    assert float(y.abs().max()) == pytest.approx(0.97, abs=1e-6)
    for mode in ("sndfile", "encode_16bits", "save_wav"):
        p, q = wave.pack_pcm16(y, mode), wave.pack_pcm16(-y, mode)
        assert torch.equal(p, -q)
        order = torch.argsort(y)
        assert bool((p[order][1:] >= p[order][:-1]).all())
    # peak * fl(32767 / peak) may round just below 32767 and truncate: numpy does the same
    assert int(wave.pack_pcm16(y, "save_wav").abs().max()) in (32766, 32767)


def test_errors(cuda, lib):
    from mockingbird_amd.vocoder import wave
    from mockingbird_amd._lib import MbHipError
    with pytest.raises(MbHipError, match="no CPU path"):
        wave.pack_pcm16(torch.zeros(4))
    with pytest.raises(MbHipError, match="float32 or float64"):
        wave.pack_pcm16(torch.zeros(4, dtype=torch.float16, device="cuda"))
    with pytest.raises(ValueError):
        wave.pack_pcm16(torch.zeros(4, device="cuda"), "pcm24")
    with pytest.raises(ValueError, match="zero-size"):
        wave.peak_normalize_(torch.zeros(0, device="cuda"))
    assert wave.pack_pcm16(torch.zeros(0, device="cuda")).numel() == 0
    # default mode = the pinned encode_16bits (wavernn/audio.py:38-39), not the unpinned libsndfile restatement
    y = torch.linspace(-1.2, 1.2, 4097, device="cuda")
    assert torch.equal(wave.pack_pcm16(y), wave.pack_pcm16(y, "encode_16bits"))
    # all-zero waveform: the reference's 0/0 -> NaN
    assert bool(torch.isnan(wave.peak_normalize_(torch.zeros(8, device="cuda"))).all())


@pytest.mark.parametrize("normalize,pcm16", [(0.97, "encode_16bits"), (0.97, None), (None, "sndfile"), (None, None)])
def test_finish_batch_equals_the_one_waveform_calls(cuda, lib, normalize, pcm16):
    """wave.finish_batch (mb_wave_finish_batch: the tails of a batch of requests in two launches) is bit for bit insert_breaks ->
    peak_normalize_ -> pack_pcm16 per item: several sentences per item, a boundary past the end (numpy-style clipping), ragged views
    of one vocoder output, an all-zero item (0 / 0 = NaN like numpy)."""
    from mockingbird_amd.vocoder import wave
    g = torch.Generator().manual_seed(11)
    base = (torch.rand(5, 1, 9000, generator=g) * 2 - 1).cuda() * torch.tensor([0.3, 1.7, 0.01, 0.9, 0.0]).view(5, 1, 1).cuda()
    lens = [9000, 6400, 7777, 1, 5000]
    wavs = [base[i, :, :n] for i, n in enumerate(lens)]
    breaks = [[10, 20, 5], [25], [40, 10], [1], [12, 8]]  # x hop 256: item 2's second boundary lies past its 7777 samples
    for br in (breaks, None):
        got = wave.finish_batch(wavs, br, 256, 16000, 0.15, normalize, pcm16)
        for i, w in enumerate(wavs):
            r = w.reshape(-1)
            if br is not None:
                r = wave.insert_breaks(r, br[i], 256, 16000, 0.15)
            r = r.contiguous().clone()
            if normalize is not None:
                wave.peak_normalize_(r, normalize)
            if pcm16 is not None:
                r = wave.pack_pcm16(r, pcm16)
            a, b = got[i].cpu(), r.cpu()
            assert a.dtype == b.dtype and a.shape == b.shape
            assert torch.equal(a, b) or (a.is_floating_point() and torch.equal(torch.isnan(a), torch.isnan(b))
                                          and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)))
    with pytest.raises(ValueError):
        wave.finish_batch(wavs, None, 256, 16000, 0.15, 0.97, "save_wav")
