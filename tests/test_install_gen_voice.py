"""The drop-in boundary, exercised by the reference's OWN caller: mockingbird_amd.install() aliases the four facade
modules, then /root/reference/gen_voice.py runs as the script it is (gen_voice.py:49-126: load the three models,
Synthesizer.load_preprocess_wav -> encoder.embed_utterance, split the text, gen_one_wav :15-47:
synthesize_spectrograms(style_idx=-1, min_stop_token=4, steps=400) -> np.concatenate -> vocoder.infer_waveform ->
breaks -> trim -> normalise -> sf.write).

CPU test: the two device models are replaced by deterministic host stand-ins (the kernels are covered by the -m gpu
parity tests); everything between the reference script and the device classes -- aliasing, class statics, text front
end, chunking, padding, seeds, tail trim, facade globals -- is the shipped code.  The third-party packages the
reference imports and this image lacks (librosa, soundfile, cn2an, webrtcvad-backed encoder) are stubbed.
Needs the reference checkout (absent on the GPU box: skipped there)."""
import os
import runpy
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

import synth

REF = Path("/root/reference")
pytestmark = pytest.mark.skipif(not (REF / "gen_voice.py").exists(), reason="reference checkout not present")


class _HostTacotron:
    """Stand-in for TacotronDevice: same generate() contract, computed on the host."""
    r = 2

    def __init__(self, state, device, **kw):
        self.calls = []

    def generate(self, chars, speaker_embedding, steps=2000, style_idx=0, min_stop_token=5, enc_masks=None,
                 dropout=None, seed=None):
        B, T = chars.shape
        self.calls.append(dict(B=B, T=T, steps=steps, style_idx=style_idx, min_stop_token=min_stop_token, seed=seed))
        F = min(steps, 40)
        g = torch.Generator().manual_seed(int(seed) % (2 ** 31))
        mel = torch.randn(B, 80, F, generator=g) - 1.0
        mel[:, :, F - 7:] = -4.0  # silent tail: trimmed by the facade (inference.py:136-137)
        return mel, mel.clone(), torch.zeros(B, F // 2, T)


class _HostGan:
    """Stand-in for GanGenerator: [B, 80, F] -> [B, 1, 200 F]."""
    hop = 200

    def __init__(self, h, state, kind, top_k=4, dtype="f32"):
        pass

    def __call__(self, mel):
        B, _, F = mel.shape
        t = torch.arange(F * 200, dtype=torch.float32)
        return (0.3 * torch.sin(0.01 * t))[None, None].repeat(B, 1, 1) * mel.abs().mean()


def _stub_modules(monkeypatch, written):
    sf = types.ModuleType("soundfile")
    sf.write = lambda fn, wav, sr, *a, **k: written.append((fn, np.asarray(wav), sr))
    monkeypatch.setitem(sys.modules, "soundfile", sf)
    cn = types.ModuleType("cn2an")
    cn.transform = lambda txt, mode: txt
    monkeypatch.setitem(sys.modules, "cn2an", cn)
    lr = types.ModuleType("librosa")
    lr.load = lambda path, sr=None: (np.sin(np.arange(16000) * 0.05).astype(np.float32) * 0.4, sr)
    lrf = types.ModuleType("librosa.filters")
    lr.filters = lrf
    monkeypatch.setitem(sys.modules, "librosa", lr)
    monkeypatch.setitem(sys.modules, "librosa.filters", lrf)
    enc = types.ModuleType("models.encoder.inference")
    enc.loaded = []
    enc.load_model = lambda p: enc.loaded.append(p)
    enc.embed_utterance = lambda wav, return_partials=False: ((np.ones(256, np.float32) / 16.0), None, None)
    enc.preprocess_wav = lambda wav, *a: wav
    monkeypatch.setitem(sys.modules, "models.encoder.inference", enc)
    return enc


def test_reference_gen_voice_runs_unchanged_against_the_aliased_facades(monkeypatch, tmp_path):
    import mockingbird_amd
    import mockingbird_amd.synthesizer.inference as syn_mod
    import mockingbird_amd.vocoder.gan as gan_mod
    import importlib
    importlib.reload(importlib.import_module("mockingbird_amd.vocoder.hifigan.inference"))
    monkeypatch.syspath_prepend(str(REF))
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.") or m == "utils" or m.startswith("utils.")
                 or m == "monotonic_align"]:
        monkeypatch.delitem(sys.modules, name)
    written = []
    enc = _stub_modules(monkeypatch, written)
    monkeypatch.setattr(syn_mod, "TacotronDevice", _HostTacotron)
    monkeypatch.setattr(gan_mod, "GanGenerator", _HostGan)
    names = mockingbird_amd.install()
    for n in names:
        monkeypatch.setitem(sys.modules, n, sys.modules[n])  # undone at teardown
    assert "models.synthesizer.inference" in names
    # the hard-coded checkpoint paths of gen_voice.py:121-123, relative to the working directory
    monkeypatch.chdir(tmp_path)
    (tmp_path / "synthesizer/saved_models").mkdir(parents=True)
    (tmp_path / "vocoder/saved_models/pretrained").mkdir(parents=True)
    torch.save(synth.tacotron_state(seed=3), tmp_path / "synthesizer/saved_models/mandarin.pt")
    import json
    torch.save(synth.gan_state(synth.small(synth.HIFIGAN_16K, 64), "hifigan", seed=1),
               tmp_path / "vocoder/saved_models/pretrained/g_hifigan.pt")
    (tmp_path / "vocoder/saved_models/pretrained/config.json").write_text(json.dumps(synth.small(synth.HIFIGAN_16K, 64)))
    txt = tmp_path / "in.txt"
    txt.write_text("hello world, this is a test\nsecond line.\n")
    monkeypatch.setattr(sys, "argv", ["gen_voice.py", str(txt), "voice.wav"])
    torch.manual_seed(11)
    ns = runpy.run_path(str(REF / "gen_voice.py"), run_name="gen_voice_under_test")
    # ---- what the script did, through OUR facades ----
    assert ns["Synthesizer"] is syn_mod.Synthesizer and ns["vocoder"] is sys.modules["models.vocoder.hifigan.inference"]
    assert enc.loaded == [Path("encoder/saved_models/pretrained.pt")]
    assert len(written) == 1
    fn, wav, sr = written[0]
    assert sr == 16000 == syn_mod.Synthesizer.sample_rate and fn.endswith("_1_voice.wav.wav")  # gen_voice.py:44-45: "%s_%d_%s.wav"
    assert wav.ndim == 1 and np.isfinite(wav).all() and abs(np.abs(wav).max() - 0.97) < 1e-6  # gen_voice.py:41
    # 3 sentences ("hello world" / "this is a test" / "second line."): <= 33 frames each after the tail trim,
    # 200 samples per frame, one 0.15 s break per sentence
    gap = int(0.15 * 16000)
    assert (len(wav) - 3 * gap) % 200 == 0 and 0 < (len(wav) - 3 * gap) // 200 <= 3 * 33
    assert (wav[-gap:] == 0).all()


def test_statics_and_seed_semantics_of_the_synthesizer_facade(monkeypatch, tmp_path):
    """Class statics exist and delegate to the reference's CPU code (inference.py:144-181); calls with no explicit
    seed draw their RNG keys from torch's global generator: reproducible under torch.manual_seed, different
    otherwise, and different for every chunk of 16."""
    import mockingbird_amd.synthesizer.inference as syn_mod
    monkeypatch.syspath_prepend(str(REF))
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.") or m == "utils" or m.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, name)
    _stub_modules(monkeypatch, [])
    wav = syn_mod.Synthesizer.load_preprocess_wav(tmp_path / "x.wav")
    assert wav.shape == (16000,) and np.isfinite(wav).all() and 0 < np.abs(wav).max() <= 0.9 + 1e-6  # rescaled, then logMMSE
    for name in ("make_spectrogram", "griffin_lim"):
        assert callable(getattr(syn_mod.Synthesizer, name))
    monkeypatch.setattr(syn_mod, "TacotronDevice", _HostTacotron)
    torch.save(synth.tacotron_state(seed=3), tmp_path / "t.pt")
    s = syn_mod.Synthesizer(tmp_path / "t.pt", verbose=False)
    texts = [f"utterance number {i}" for i in range(35)]  # 3 chunks of synthesis_batch_size = 16
    emb = [np.ones(256, np.float32)] * 35
    torch.manual_seed(3)
    a = s.synthesize_spectrograms(texts, emb)
    seeds_a = [c["seed"] for c in s._model.calls]
    torch.manual_seed(3)
    b = s.synthesize_spectrograms(texts, emb)
    seeds_b = [c["seed"] for c in s._model.calls[3:]]
    c = s.synthesize_spectrograms(texts, emb)
    seeds_c = [c_["seed"] for c_ in s._model.calls[6:]]
    assert [c_["B"] for c_ in s._model.calls[:3]] == [16, 16, 3]
    assert seeds_a == seeds_b and len(set(seeds_a)) == 3 and not set(seeds_c) & set(seeds_a)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(a) == 35
    assert all(m.shape[0] == 80 and m.shape[1] == 33 and m.dtype == np.float32 for m in a)  # 40 frames - 7 trimmed
    assert s._model.calls[0]["style_idx"] == 0 and s._model.calls[0]["min_stop_token"] == 5 and s._model.calls[0]["steps"] == 2000


def test_install_can_alias_the_voice_conversion_model_package(monkeypatch):
    """install(ppg2mel=True): `models.ppg2mel` (run.py:13,45,77; control/mkgui/app_vc.py:12,144) becomes the HIP-backed
    module -- MelDecoderMOLv2 with the reference's constructor keywords, load_model -- while the package's reference
    submodules (utils, train, ...) still import from the reference's own directory."""
    import importlib
    import inspect
    import mockingbird_amd
    monkeypatch.syspath_prepend(str(REF))
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    for k in saved:
        monkeypatch.delitem(sys.modules, k, raising=False)
    try:
        names = mockingbird_amd.install(ppg2mel=True)
        assert "models.ppg2mel" in names
        conv = importlib.import_module("models.ppg2mel")
        import mockingbird_amd.ppg2mel as mine
        assert conv is mine and conv.MelDecoderMOLv2 is mine.MelDecoderMOLv2 and callable(conv.load_model)
        ref_params = ["num_speakers", "spk_embed_dim", "bottle_neck_feature_dim", "encoder_dim", "encoder_downsample_rates",
                      "attention_rnn_dim", "decoder_rnn_dim", "num_decoder_rnn_layer", "concat_context_to_last", "prenet_dims",
                      "num_mixtures", "frames_per_step", "mask_padding"]
        got = list(inspect.signature(conv.MelDecoderMOLv2.__init__).parameters)[1:]
        assert got[:len(ref_params)] == ref_params, got  # models/ppg2mel/__init__.py:22-37, keyword for keyword
        post = importlib.import_module("models.ppg2mel.utils.cnn_postnet")  # reference CPU code, still reachable
        assert post.__file__.startswith(str(REF)) and hasattr(post, "Postnet")
        assert "models.ppg2mel" not in mockingbird_amd.install()  # opt-in only
    finally:
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            sys.modules.pop(k, None)
        sys.modules.update(saved)
        if str(REF / "models" / "ppg2mel") in mockingbird_amd.ppg2mel.__path__:
            mockingbird_amd.ppg2mel.__path__.remove(str(REF / "models" / "ppg2mel"))
