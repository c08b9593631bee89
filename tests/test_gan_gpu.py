"""HiFi-GAN / Fre-GAN generator forward: HIP (mb_gan_forward via the facade classes) vs the oracle.
Gate (north_star / SURVEY.md section 8d): audio RMS(delta) <= 1e-4, relative RMS <= 1e-3 on O(1) fixtures."""
import numpy as np
import pytest
import torch

import hiputil
import synth
from oracle import gan as og

pytestmark = pytest.mark.gpu

RMS_TOL, REL_TOL = 1e-4, 1e-3


def _run(kind, h, frames, batch, seed, dtype="f32"):
    from mockingbird_amd.vocoder.gan import GanGenerator
    st = synth.gan_state(h, kind, seed=seed)
    gen = GanGenerator(h, st["generator"], 0 if kind == "hifigan" else 1, dtype=dtype)
    mel = torch.from_numpy(synth.mel_input(frames, batch, seed=seed + 1))
    with torch.no_grad():
        w = og.fold_weight_norm_state(st["generator"])
        ref = (og.hifigan_forward if kind == "hifigan" else og.fregan_forward)(w, h, mel)
    y = gen(mel.cuda())
    torch.cuda.synchronize()
    return y.cpu(), ref


@pytest.mark.parametrize("kind,cfg", [("hifigan", synth.HIFIGAN_16K), ("fregan", synth.FREGAN_16K)])
@pytest.mark.parametrize("uic,frames,batch", [(64, 16, 2), (64, 37, 1), (512, 12, 1), (512, 200, 1)])
def test_gan_forward_matches_oracle(cuda, lib, kind, cfg, uic, frames, batch):
    """(512, 200, 1) is BASELINE configs[0] at full size: the shipped config, one (80, 200) mel."""
    h = synth.small(cfg, uic)
    y, ref = _run(kind, h, frames, batch, seed=3)
    assert y.shape == ref.shape == (batch, 1, frames * 200)
    e = hiputil.relerr(y, ref)
    assert e["nan"] == 0 and e["rms"] <= RMS_TOL and e["rel_rms"] <= REL_TOL, e


F16_REL_TOL = 5e-3  # fp16 gate, relative RMS, reported separately


@pytest.mark.parametrize("fuse", ["nochain", "units", "none"])
def test_gan_f32_fuse_levels_agree(cuda, lib, monkeypatch, fuse):
    """MBHIP_GAN_FUSE (read at create time) on the fp32-result path: every level of fusion is the same generator -- against the
    oracle at the fp32 gates and within 2e-6 of the default's waveform RMS."""
    h = synth.small(synth.HIFIGAN_16K, 128)
    monkeypatch.delenv("MBHIP_GAN_FUSE", raising=False)
    y0, ref = _run("hifigan", h, 31, 2, seed=7)
    monkeypatch.setenv("MBHIP_GAN_FUSE", fuse)
    y1, _ = _run("hifigan", h, 31, 2, seed=7)
    e = hiputil.relerr(y1, ref)
    assert e["nan"] == 0 and e["rms"] <= RMS_TOL and e["rel_rms"] <= REL_TOL, e
    d = (y1.double() - y0.double()).pow(2).mean().sqrt()
    assert float(d) <= 2e-6 * max(float(y0.double().pow(2).mean().sqrt()), 1e-3), float(d)


def test_gan_fuse_switch_rejects_unknown_values(cuda, lib, monkeypatch):
    """A misspelt MBHIP_GAN_FUSE must fail the create call loudly, not silently run the default."""
    from mockingbird_amd.vocoder.gan import GanGenerator
    from mockingbird_amd import _lib
    h = synth.small(synth.HIFIGAN_16K, 64)
    st = synth.gan_state(h, "hifigan", seed=1)
    monkeypatch.setenv("MBHIP_GAN_FUSE", "unit")
    with pytest.raises(_lib.MbHipError) as ei:
        GanGenerator(h, st["generator"], 0)
    assert "MBHIP_GAN_FUSE" in str(ei.value)


# the fp16 path stores activations as 16-byte rows of 8 channels: every conv needs c_in % 8 == 0,
# i.e. upsample_initial_channel >= 128 (HiFi-GAN, 4 halvings) / 256 (Fre-GAN, 5 halvings)
@pytest.mark.parametrize("kind,cfg,uic,frames,batch", [
    ("hifigan", synth.HIFIGAN_16K, 128, 16, 2), ("hifigan", synth.HIFIGAN_16K, 256, 37, 1),
    ("hifigan", synth.HIFIGAN_16K, 512, 12, 1), ("fregan", synth.FREGAN_16K, 256, 16, 2),
    ("fregan", synth.FREGAN_16K, 256, 37, 1), ("fregan", synth.FREGAN_16K, 512, 12, 1),
    ("hifigan", synth.HIFIGAN_16K, 512, 200, 1), ("fregan", synth.FREGAN_16K, 512, 200, 2)])
def test_gan_forward_f16_matches_oracle(cuda, lib, kind, cfg, uic, frames, batch):
    """fp16 MFMA path (BASELINE configs[4]) against the fp32 oracle."""
    h = synth.small(cfg, uic)
    y, ref = _run(kind, h, frames, batch, seed=3, dtype="f16")
    assert y.shape == ref.shape == (batch, 1, frames * 200)
    e = hiputil.relerr(y, ref)
    print("f16 parity", kind, uic, frames, batch, e)
    assert e["nan"] == 0 and e["rel_rms"] <= F16_REL_TOL, e


def _len24k(h, frames):
    t = frames
    for u, k in zip(h["upsample_rates"], h["upsample_kernel_sizes"]):
        t = t * u + 2 * ((k - 1) // 2) - (k - 1)
    return t


@pytest.mark.parametrize("uic,frames,batch,seed", list(synth.GAN24K_CASES) + [(128, 31, 3, 9)])
def test_hifigan_24k_interp_upsampler_matches_oracle(cuda, lib, uic, frames, batch, seed):
    """h.sampling_rate == 24000 (hifigan/models.py:107-118): nearest-interpolate + Conv1d upsampler, folded
    into the conv's read.  Even upsample kernels leave each stage one sample short, as the reference's does."""
    h = synth.small(synth.HIFIGAN_24K, uic)
    y, ref = _run("hifigan", h, frames, batch, seed=seed)
    assert y.shape == ref.shape == (batch, 1, _len24k(h, frames))
    assert y.shape[-1] < frames * 300
    e = hiputil.relerr(y, ref)
    assert e["nan"] == 0 and e["rms"] <= RMS_TOL and e["rel_rms"] <= REL_TOL, e


@pytest.mark.parametrize("uic,frames,batch", [(128, 16, 2), (512, 23, 1)])
def test_hifigan_24k_f16_matches_oracle(cuda, lib, uic, frames, batch):
    h = synth.small(synth.HIFIGAN_24K, uic)
    y, ref = _run("hifigan", h, frames, batch, seed=4, dtype="f16")
    assert y.shape == ref.shape == (batch, 1, _len24k(h, frames))
    e = hiputil.relerr(y, ref)
    print("f16 24k parity", uic, frames, batch, e)
    assert e["nan"] == 0 and e["rel_rms"] <= F16_REL_TOL, e


def test_gan_f16_rejects_narrow_channels(cuda, lib):
    """A model whose last stage is narrower than 8 channels cannot take the fp16 layout: creation
    fails loudly (no silent fp32 fallback)."""
    from mockingbird_amd.vocoder.gan import GanGenerator
    from mockingbird_amd._lib import MbHipError
    h = synth.small(synth.HIFIGAN_16K, 64)
    st = synth.gan_state(h, "hifigan", seed=1)["generator"]
    with pytest.raises(MbHipError, match="multiples of 8"):
        GanGenerator(h, st, 0, dtype="f16")


def test_gan_f16_full_size_properties(cuda, lib):
    """BASELINE-size run of the fp16 path (batch 4 x (80,200), full 512-channel model): bounded output,
    batch items independent (item k alone == item k inside the batch, bit for bit), and close to the
    fp32 MFMA path on the same input."""
    from mockingbird_amd.vocoder.gan import GanGenerator
    h = synth.HIFIGAN_16K
    st = synth.gan_state(h, "hifigan", seed=5)["generator"]
    g16, g32 = GanGenerator(h, st, 0, dtype="f16"), GanGenerator(h, st, 0, dtype="f32")
    mel = torch.from_numpy(synth.mel_input(200, 4, seed=0)).cuda()
    y16, y32 = g16(mel), g32(mel)
    y1 = g16(mel[2:3])
    torch.cuda.synchronize()
    assert torch.equal(y1[0], y16[2])
    assert float(y16.abs().max()) <= 1.0 and int(torch.isnan(y16).sum()) == 0
    e = hiputil.relerr(y16, y32)
    print("f16 vs f32 path, full size", e)
    assert e["rel_rms"] <= F16_REL_TOL, e


@pytest.mark.parametrize("kind,cfg", [("hifigan", synth.HIFIGAN_16K), ("fregan", synth.FREGAN_16K)])
def test_gan_time_translation_property(cuda, lib, kind, cfg):
    """Size-independent property at full length (BASELINE config 0: 80x200): the generator is
    fully convolutional, so the interior of the output for a mel equals the output for the
    same mel embedded in a longer zero-padded one, away from the receptive-field border."""
    from mockingbird_amd.vocoder.gan import GanGenerator
    h = cfg
    st = synth.gan_state(h, kind, seed=5)
    gen = GanGenerator(h, st["generator"], 0 if kind == "hifigan" else 1)
    mel = torch.from_numpy(synth.mel_input(200, 1, seed=0)).cuda()
    y = gen(mel)
    # same content shifted by 8 frames inside a longer input
    longer = torch.zeros(1, 80, 232, device="cuda")
    longer[:, :, 8:208] = mel
    y2 = gen(longer)
    torch.cuda.synchronize()
    a = y[0, 0, 40 * 200:160 * 200].cpu()
    b = y2[0, 0, 48 * 200:168 * 200].cpu()
    e = hiputil.relerr(a, b)
    assert e["nan"] == 0 and e["rms"] <= 1e-5, e
    assert float(y.abs().max()) <= 1.0 and float(y.pow(2).mean()) > 1e-4


def test_facade_signature_and_errors(cuda, lib, tmp_path):
    """models/vocoder/hifigan/inference.py:22-74 semantics: json lookup beside the checkpoint,
    exception text when unloaded, numpy in -> (float32 numpy, sample_rate) out."""
    import importlib, json
    import mockingbird_amd.vocoder.hifigan.inference as inf
    inf = importlib.reload(inf)
    assert not inf.is_loaded()
    with pytest.raises(Exception, match="Please load hifi"):
        inf.infer_waveform(np.zeros((80, 10), np.float32))
    h = synth.small(synth.HIFIGAN_16K, 32)
    torch.save(synth.gan_state(h, "hifigan", seed=1), tmp_path / "g_test.pt")
    (tmp_path / "config.json").write_text(json.dumps(h))
    inf.load_model(tmp_path / "g_test.pt", verbose=False)
    assert inf.is_loaded() and inf.output_sample_rate == 16000
    mel = synth.mel_input(9, 1, seed=2)[0]
    wav, sr = inf.infer_waveform(mel)
    assert sr == 16000 and wav.dtype == np.float32 and wav.shape == (1800,)
    wav2, _ = inf.infer_waveform(torch.from_numpy(mel))  # CPU tensor input (run.py:90)
    assert np.array_equal(wav, wav2)
    outs, _ = inf.infer_waveform_batch([mel, mel[:, :5], mel])
    assert outs[0].shape == (1800,) and outs[1].shape == (1000,)
    assert np.allclose(outs[0], wav, atol=1e-6) and np.array_equal(outs[0], outs[2])


def test_fregan_f16_config4_share_properties(cuda, lib):
    """BASELINE configs[4] per-GPU share: Fre-GAN fp16, batch 8 x mel (80,3000) (64 utterances over 8 GPUs).
    Size-independent properties at full size: bounded finite output of 8 x 600000 samples, batch items
    independent (item k alone == item k inside the batch, bit for bit: tiles never straddle utterances), and
    agreement with the fp32 MFMA path on one full-length item within the fp16 gate."""
    from mockingbird_amd.vocoder.gan import GanGenerator
    h = synth.FREGAN_16K
    st = synth.gan_state(h, "fregan", seed=5)["generator"]
    g16, g32 = GanGenerator(h, st, 1, dtype="f16"), GanGenerator(h, st, 1, dtype="f32")
    mel = torch.from_numpy(synth.mel_input(3000, 8, seed=0)).cuda()
    y16 = g16(mel)
    y1 = g16(mel[5:6])
    y32 = g32(mel[5:6])
    torch.cuda.synchronize()
    assert y16.shape == (8, 1, 600000)
    assert float(y16.abs().max()) <= 1.0 and int(torch.isnan(y16).sum()) == 0
    assert torch.equal(y1[0], y16[5])
    e = hiputil.relerr(y1, y32)
    print("fregan f16 vs f32, 3000 frames", e)
    assert e["rel_rms"] <= F16_REL_TOL, e


def test_fregan_f16_config4_share_vs_oracle(cuda, lib):
    """BASELINE configs[4] per-GPU share at FULL size against the oracle itself: Fre-GAN fp16, batch 8 x mel (80,3000); the
    first and the last item of the batch against oracle.gan.fregan_forward (fp32 ATen CPU, 1.15 TFLOP per item), fp16 gate
    (relative RMS <= 5e-3).  The HiFi-GAN `hifigan_f16` bench object's shape (32 x 200) rides along."""
    from mockingbird_amd.vocoder.gan import GanGenerator
    h = synth.FREGAN_16K
    st = synth.gan_state(h, "fregan", seed=5)["generator"]
    g16 = GanGenerator(h, st, 1, dtype="f16")
    mel = torch.from_numpy(synth.mel_input(3000, 8, seed=0))
    y16 = g16(mel.cuda()).cpu()
    w = og.fold_weight_norm_state(st)
    for k in (0, 7):
        with torch.no_grad():
            ref = og.fregan_forward(w, h, mel[k:k + 1])
        e = hiputil.relerr(y16[k:k + 1], ref)
        print("fregan f16 8x3000 item", k, e)
        assert e["nan"] == 0 and e["rel_rms"] <= F16_REL_TOL, (k, e)
    hh = synth.HIFIGAN_16K
    sth = synth.gan_state(hh, "hifigan", seed=6)["generator"]
    gh = GanGenerator(hh, sth, 0, dtype="f16")
    melh = torch.from_numpy(synth.mel_input(200, 32, seed=1))
    yh = gh(melh.cuda()).cpu()
    with torch.no_grad():
        refh = og.hifigan_forward(og.fold_weight_norm_state(sth), hh, melh[::31])  # items 0 and 31
    e = hiputil.relerr(yh[::31], refh)
    assert e["nan"] == 0 and e["rel_rms"] <= F16_REL_TOL, e


@pytest.mark.parametrize("kind,cfg,uic,frames,batch", [("hifigan", synth.HIFIGAN_16K, 256, 23, 2),
                                                       ("fregan", synth.FREGAN_16K, 512, 12, 1)])
def test_gan_f16_unfused_path_matches_oracle(cuda, lib, monkeypatch, kind, cfg, uic, frames, batch):
    """MBHIP_GAN_FUSE=none (read at create time): every ResBlock conv as its own conv1d_f16 launch -- the path
    shapes without a fused ResBlock-unit instance take -- against the fp32 oracle, and against the fused path."""
    h = synth.small(cfg, uic)
    fused, ref = _run(kind, h, frames, batch, seed=5, dtype="f16")
    monkeypatch.setenv("MBHIP_GAN_FUSE", "none")
    unfused, _ = _run(kind, h, frames, batch, seed=5, dtype="f16")
    e = hiputil.relerr(unfused, ref)
    assert e["nan"] == 0 and e["rel_rms"] <= F16_REL_TOL, e
    d = hiputil.relerr(unfused, fused)
    assert d["rel_rms"] <= F16_REL_TOL, d


@pytest.mark.parametrize("kind,cfg,uic,frames,batch", [("hifigan", synth.HIFIGAN_16K, 512, 23, 2),
                                                       ("fregan", synth.FREGAN_16K, 512, 12, 1)])
def test_gan_f16_per_unit_launches_match_group_launches(cuda, lib, monkeypatch, kind, cfg, uic, frames, batch):
    """MBHIP_GAN_FUSE=units (read at create time): every ResBlock unit as its own mb_resblock_pair_f16
    launch -- the production path of round 2, and today's path for any shape the one-launch stage / chain kernels do not take
    -- against the oracle and against the default (narrow stages and k = 3 ResBlocks as one launch each).  Run twice: the unit
    kernel keeps a bias block in LDS, consecutive launches carry different ones (see test_resblock_pair_gpu.py)."""
    h = synth.small(cfg, uic)
    grouped, ref = _run(kind, h, frames, batch, seed=5, dtype="f16")
    monkeypatch.setenv("MBHIP_GAN_FUSE", "units")
    for _ in range(2):
        units, _ = _run(kind, h, frames, batch, seed=5, dtype="f16")
        e = hiputil.relerr(units, ref)
        assert e["nan"] == 0 and e["rel_rms"] <= F16_REL_TOL, e
        d = hiputil.relerr(units, grouped)
        assert d["rel_rms"] <= F16_REL_TOL, d


@pytest.mark.parametrize("kind,cfg,uic,dtype", [("hifigan", synth.HIFIGAN_16K, 64, "f32"), ("fregan", synth.FREGAN_16K, 64, "f32"),
                                                ("hifigan", synth.HIFIGAN_16K, 256, "f16"), ("fregan", synth.FREGAN_16K, 256, "f16")])
def test_ragged_batch_equals_single_runs(cuda, lib, kind, cfg, uic, dtype):
    """mb_gan_forward_ragged: utterances of different lengths in one launch sequence.  The generators are not causal
    (a zero-padded mel changes the tail), so every conv masks positions beyond an item's length as ITS zero padding:
    item i must equal forward(mel_i) -- bit for bit on the fp32 path (same products, same order per output position),
    within fp16 rounding of the oracle on the fp16 path -- and the oracle."""
    from mockingbird_amd.vocoder.gan import GanGenerator
    h = synth.small(cfg, uic)
    st = synth.gan_state(h, kind, seed=7)
    gen = GanGenerator(h, st["generator"], 0 if kind == "hifigan" else 1, dtype=dtype)
    frames = [37, 5, 23, 37, 1, 16]
    mels = [torch.from_numpy(synth.mel_input(f, 1, seed=30 + i)[0]) for i, f in enumerate(frames)]
    outs = gen.forward_ragged(mels)
    w = og.fold_weight_norm_state(st["generator"])
    for m, f, y in zip(mels, frames, outs):
        assert tuple(y.shape) == (1, f * 200)
        single = gen(m[None].cuda())[0]
        with torch.no_grad():
            ref = (og.hifigan_forward if kind == "hifigan" else og.fregan_forward)(w, h, m[None])[0]
        e = hiputil.relerr(y, ref)
        if dtype == "f32":
            assert torch.equal(y, single), (f, float((y - single).abs().max()))
            assert e["nan"] == 0 and e["rms"] <= RMS_TOL and e["rel_rms"] <= REL_TOL, (f, e)
        else:
            d = hiputil.relerr(y, single)
            assert d["rel_rms"] <= 1e-6 or d["max_abs"] <= 2e-3, (f, d)
            assert e["nan"] == 0 and e["rel_rms"] <= F16_REL_TOL, (f, e)


def test_infer_waveform_batch_ragged_facade(cuda, lib, tmp_path):
    """hifigan.infer_waveform_batch on mels of different lengths (BASELINE configs[3]: tail-trimmed spectrograms) ==
    infer_waveform per utterance, with the device-side tail (breaks, normalise, PCM) applied per item."""
    import importlib
    import json
    import mockingbird_amd.vocoder.hifigan.inference as voc
    voc = importlib.reload(voc)
    h = synth.small(synth.HIFIGAN_16K, 64)
    torch.save(synth.gan_state(h, "hifigan", seed=1), tmp_path / "g_test.pt")
    (tmp_path / "config.json").write_text(json.dumps(h))
    voc.load_model(tmp_path / "g_test.pt", verbose=False)
    frames = [31, 12, 31, 7, 19]
    mels = [synth.mel_input(f, 1, seed=50 + i)[0] for i, f in enumerate(frames)]
    wavs, sr = voc.infer_waveform_batch(mels)
    assert sr == 16000 and len(wavs) == len(frames)
    for m, f, wv in zip(mels, frames, wavs):
        single, _ = voc.infer_waveform(m)
        assert wv.shape == (f * 200,) and wv.dtype == np.float32 and np.array_equal(wv, single)
    pcm, _ = voc.infer_waveform_batch(mels, normalize=0.97, pcm16="encode_16bits", breaks=[[f] for f in frames], break_hop=256)
    from oracle import wave as owv
    for wv, pc in zip(wavs, pcm):
        ref = owv.encode_16bits(owv.peak_normalize(np.concatenate([wv, np.zeros(2400, np.float32)]), np.float32(0.97)))
        assert pc.dtype == np.int16 and np.array_equal(pc, ref)


def test_handles_on_the_shared_pool_streams_from_two_threads(cuda, lib):
    """Every handle borrows its side streams from ONE pool per device (csrc/common.h pool_stream): two fp32 generators whose forked
    ResBlock chains land on the same three streams, driven from two host threads on two caller streams at once, and a WaveRNN loop
    on pool stream 0 between them, must each give exactly what they give alone (sharing a stream only orders the work)."""
    import threading
    from mockingbird_amd.vocoder.gan import GanGenerator
    from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
    h = synth.small(synth.HIFIGAN_16K, 128)
    gens = [GanGenerator(h, synth.gan_state(h, "hifigan", seed=s)["generator"], 0, dtype="f32") for s in (3, 4)]
    mels = [torch.from_numpy(synth.mel_input(40, 3, seed=10 + i)).cuda() for i in range(2)]
    w = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
    wmel = torch.from_numpy(synth.wavernn_mel(30, seed=100) / 4.0).cuda()
    alone = [g(m).cpu() for g, m in zip(gens, mels)]
    w_alone = w.generate_samples(wmel, True, 2000, 200, seed=7).cpu()
    torch.cuda.synchronize()
    outs, errs = [None, None, None], []

    def run_gan(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(6):
                    y = gens[i](mels[i])
                st.synchronize()
            outs[i] = y.cpu()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    def run_wrn():
        try:
            for _ in range(3):
                y = w.generate_samples(wmel, True, 2000, 200, seed=7)
            outs[2] = y.cpu()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run_gan, args=(0,)), threading.Thread(target=run_gan, args=(1,)), threading.Thread(target=run_wrn)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert torch.equal(outs[0], alone[0]) and torch.equal(outs[1], alone[1])
    assert torch.equal(outs[2], w_alone)
