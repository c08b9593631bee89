"""Generate the golden fixtures under tests/golden/ by running the REAL reference
(/root/reference, imported read-only with the shims of tests/refimport.py) on seeded synthetic
checkpoints (tests/synth.py).  Run in the build container only:

    python tests/golden/make_golden.py

Fixtures hold inputs' seeds/shapes and the reference OUTPUTS only (weights are regenerated from
the seed by tests/synth.py), so they stay small.  torch version is recorded: the arithmetic
lives in ATen CPU kernels."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import refimport
import synth

refimport.setup()
torch.set_num_threads(1)  # deterministic reduction order for the fixtures


def gan():
    from utils.util import AttrDict
    from models.vocoder.hifigan.models import Generator
    from models.vocoder.fregan.generator import FreGAN
    out = {}
    for kind, cfg, cls in (("hifigan", synth.HIFIGAN_16K, Generator), ("fregan", synth.FREGAN_16K, FreGAN)):
        for uic, frames, batch, seed in ((64, 16, 2, 3), (512, 8, 1, 4)):
            h = synth.small(cfg, uic)
            st = synth.gan_state(h, kind, seed=seed)
            g = cls(AttrDict(h))
            g.load_state_dict(st["generator"])
            g.eval()
            g.remove_weight_norm()
            mel = torch.from_numpy(synth.mel_input(frames, batch, seed=seed + 1))
            with torch.no_grad():
                y = g(mel)
            out[f"{kind}_uic{uic}_f{frames}_b{batch}_s{seed}"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "gan.npz"), torch_version=torch.__version__, **out)
    print("gan.npz", {k: v.shape for k, v in out.items()})


def gan_rb2():
    """h.resblock == '2': ResBlock2 generators (hifigan/models.py:51-72,100) and the VITS decoder built with resblock='2' (vits.py:251)."""
    import types
    from utils.util import AttrDict
    from models.vocoder.hifigan.models import Generator
    out = {}
    for uic, frames, batch, seed in synth.GAN_RB2_CASES:
        h = synth.small(synth.HIFIGAN_RB2, uic)
        st = synth.gan_state(h, "hifigan", seed=seed)
        g = Generator(AttrDict(h))
        g.load_state_dict(st["generator"])
        g.eval()
        g.remove_weight_norm()
        mel = torch.from_numpy(synth.mel_input(frames, batch, seed=seed + 1))
        with torch.no_grad():
            y = g(mel)
        out[f"hifigan_rb2_uic{uic}_f{frames}_b{batch}_s{seed}"] = y.numpy()
    for name in ("loguru", "monotonic_align"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "loguru":
                m.logger = types.SimpleNamespace(info=print, debug=print, warning=print, error=print)
            sys.modules[name] = m
    from models.synthesizer.models.vits import Generator as VGen
    for name, uic, frames, batch, use_g, seed in synth.VITS_RB2_CASES:
        h = dict(synth.VITS_DEC_RB2)
        h["upsample_initial_channel"] = uic
        g = VGen(h["initial_channel"], h["resblock"], h["resblock_kernel_sizes"], h["resblock_dilation_sizes"],
                 h["upsample_rates"], uic, h["upsample_kernel_sizes"], gin_channels=h["gin_channels"])
        g.load_state_dict(synth.vits_dec_state(h, seed=seed))
        g.eval()
        z, spk = synth.vits_latent(frames, batch, seed=seed + 1)
        with torch.no_grad():
            y = g(torch.from_numpy(z), g=torch.from_numpy(spk) if use_g else None)
        out["vits_" + name] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "gan_rb2.npz"), torch_version=torch.__version__, **out)
    print("gan_rb2.npz", {k: (v.shape, float(np.abs(v).mean())) for k, v in out.items()})


def gan24k():
    """HiFi-GAN with h.sampling_rate == 24000: Interpolate+Conv1d upsampler (models.py:107-118)."""
    from utils.util import AttrDict
    from models.vocoder.hifigan.models import Generator
    out = {}
    for uic, frames, batch, seed in synth.GAN24K_CASES:
        h = synth.small(synth.HIFIGAN_24K, uic)
        st = synth.gan_state(h, "hifigan", seed=seed)
        g = Generator(AttrDict(h))
        g.load_state_dict(st["generator"])
        g.eval()
        g.remove_weight_norm()
        mel = torch.from_numpy(synth.mel_input(frames, batch, seed=seed + 1))
        with torch.no_grad():
            y = g(mel)
        out[f"hifigan24k_uic{uic}_f{frames}_b{batch}_s{seed}"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "gan24k.npz"), torch_version=torch.__version__, **out)
    print("gan24k.npz", {k: (v.shape, float(np.abs(v).mean())) for k, v in out.items()})


def wavernn():
    from models.vocoder.wavernn.models.fatchord_version import WaveRNN
    from models.vocoder.wavernn import hparams as hp
    st = synth.wavernn_state(seed=5)
    m = WaveRNN(rnn_dims=hp.voc_rnn_dims, fc_dims=hp.voc_fc_dims, bits=hp.bits, pad=hp.voc_pad,
                upsample_factors=hp.voc_upsample_factors, feat_dims=hp.num_mels, compute_dims=hp.voc_compute_dims,
                res_out_dims=hp.voc_res_out_dims, res_blocks=hp.voc_res_blocks, hop_length=hp.hop_length,
                sample_rate=hp.sample_rate, mode=hp.voc_mode)
    m.load_state_dict(st["model_state"])
    m.eval()
    out = {}
    mel = synth.wavernn_mel(30, seed=2)
    quiet = lambda *a: None
    torch.manual_seed(11)
    out["batched_f30_t600_o100_seed11"] = m.generate(torch.from_numpy(mel[None] / 4.0), True, 600, 100, True, quiet)
    torch.manual_seed(3)
    out["unbatched_f27_seed3"] = m.generate(torch.from_numpy(mel[None, :, :27] / 4.0), False, 0, 0, True, quiet)
    m.eval()  # generate() leaves the module in train mode (fatchord_version.py:255)
    with torch.no_grad():
        mp = m.pad_tensor(torch.from_numpy(mel[None] / 4.0).transpose(1, 2), pad=2, side="both")
        mels, aux = m.upsample(mp.transpose(1, 2))
    out["cond_mels_f30_stride97"] = mels[0, ::97].numpy()
    out["cond_aux_f30_stride97"] = aux[0, ::97].numpy()
    # ---- MOL mode (fatchord_version.py:213-220): the sampler alone, then generate() of a MOL model ----
    from models.vocoder.distribution import sample_from_discretized_mix_logistic
    torch.manual_seed(5)
    lg = torch.randn(7, 30) * 2
    lg[:, 20:] -= 3
    out["mol_direct_logits"] = lg.numpy()
    torch.manual_seed(9)
    out["mol_direct_sample_seed9"] = sample_from_discretized_mix_logistic(lg.unsqueeze(0).transpose(1, 2)).view(-1).numpy()
    stm = synth.wavernn_state(synth.WAVERNN_HP_MOL, seed=6)
    mm = WaveRNN(rnn_dims=hp.voc_rnn_dims, fc_dims=hp.voc_fc_dims, bits=hp.bits, pad=hp.voc_pad,
                 upsample_factors=hp.voc_upsample_factors, feat_dims=hp.num_mels, compute_dims=hp.voc_compute_dims,
                 res_out_dims=hp.voc_res_out_dims, res_blocks=hp.voc_res_blocks, hop_length=hp.hop_length,
                 sample_rate=hp.sample_rate, mode="MOL")
    mm.load_state_dict(stm["model_state"])
    mm.eval()
    torch.manual_seed(13)
    out["mol_batched_f30_t600_o100_seed13"] = mm.generate(torch.from_numpy(mel[None] / 4.0), True, 600, 100, True, quiet)
    np.savez_compressed(os.path.join(HERE, "wavernn.npz"), torch_version=torch.__version__, **out)
    print("wavernn.npz", {k: v.shape for k, v in out.items()})


def maximum_path():
    from oracle import build_ref
    build_ref.build()
    ref = build_ref.load()
    out = {}
    for b, tt, ts, seed in ((3, 17, 5, 1), (4, 200, 60, 2), (2, 400, 120, 3)):
        rng = np.random.default_rng(seed)
        neg = (rng.standard_normal((b, tt, ts)) * 3).astype(np.float32)
        tys = rng.integers(max(1, tt // 2), tt + 1, b).astype(np.int32)
        txs = np.array([rng.integers(1, min(ts, ty) + 1) for ty in tys], np.int32)
        v, p = neg.copy(), np.zeros(neg.shape, np.int32)
        ref.maximum_path_c(p, v, tys, txs)
        k = f"b{b}_t{tt}_s{ts}_seed{seed}"
        out[k + "_path"] = p.astype(np.int8)
        out[k + "_tys"], out[k + "_txs"] = tys, txs
        out[k + "_valsum"] = np.array([np.float64(v.astype(np.float64).sum())])
    np.savez_compressed(os.path.join(HERE, "maximum_path.npz"), **out)
    print("maximum_path.npz", list(out))


def tacotron():
    from models.synthesizer.models.tacotron import Tacotron
    from models.synthesizer.hparams import hparams
    from models.synthesizer.utils.symbols import symbols
    m = Tacotron(embed_dims=hparams.tts_embed_dims, num_chars=len(symbols), encoder_dims=hparams.tts_encoder_dims,
                 decoder_dims=hparams.tts_decoder_dims, n_mels=hparams.num_mels, fft_bins=hparams.num_mels,
                 postnet_dims=hparams.tts_postnet_dims, encoder_K=hparams.tts_encoder_K, lstm_dims=hparams.tts_lstm_dims,
                 postnet_K=hparams.tts_postnet_K, num_highways=hparams.tts_num_highways, dropout=hparams.tts_dropout,
                 stop_threshold=hparams.tts_stop_threshold, speaker_embedding_size=hparams.speaker_embedding_size)
    m.load_state_dict(synth.tacotron_state(seed=3)["model_state"])
    m.eval()
    out = {}
    for name, (B, tmin, tmax, steps, style, mst, seed) in {"b3_style-1": (3, 20, 30, 40, -1, 11, 5),
                                                           "b2_style0_stop": (2, 14, 18, 60, 0, 4.0, 6)}.items():
        seqs, emb = synth.tacotron_inputs(B, tmin, tmax, seed=seed)
        T = max(len(s) for s in seqs)
        chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long()
        spk = torch.tensor(np.stack(emb))
        torch.manual_seed(seed)
        with torch.no_grad():
            mel, lin, att = m.generate(chars, spk, steps=steps, style_idx=style, min_stop_token=mst)
        out[name + "_mel"], out[name + "_linear"], out[name + "_attn"] = mel.numpy(), lin.numpy(), att.numpy()
    np.savez_compressed(os.path.join(HERE, "tacotron.npz"), torch_version=torch.__version__, **out)
    print("tacotron.npz", {k: v.shape for k, v in out.items()})


def ppg2mel():
    """models/ppg2mel/rnn_decoder_mol.py Decoder.inference (B = 1) / inference_batched (B > 1) of the REAL
    module on the synthetic state; the prenet dropout draws from torch's global RNG (seed recorded)."""
    m = refimport.import_ppg2mel_decoder()
    hp = synth.PPG2MEL_HP
    out = {}
    for name, B, T, wseed, sb, mseed, rseed in synth.PPG2MEL_CASES:
        d = m.Decoder(enc_dim=hp["enc_dim"], num_mels=hp["num_mels"], frames_per_step=hp["frames_per_step"],
                      attention_rnn_dim=hp["attention_rnn_dim"], decoder_rnn_dim=hp["decoder_rnn_dim"],
                      prenet_dims=list(hp["prenet_dims"]), num_mixtures=hp["num_mixtures"],
                      encoder_down_factor=hp["encoder_down_factor"], num_decoder_rnn_layer=hp["num_decoder_rnn_layer"],
                      use_stop_tokens=True, concat_context_to_last=hp["concat_context_to_last"])
        d.load_state_dict(synth.ppg2mel_decoder_state(hp, seed=wseed, stop_bias=sb))
        d.eval()
        mem = torch.from_numpy(synth.ppg2mel_memory(B, T, seed=mseed))
        torch.manual_seed(rseed)
        with torch.no_grad():
            mel, al = d.inference(mem) if B == 1 else d.inference_batched(mem)
        out[name + "_mel"], out[name + "_align"] = mel.numpy(), al.numpy()
    # the whole MelDecoderMOLv2.inference (models/ppg2mel/__init__.py:166-192) of the real package
    pkg = refimport.import_ppg2mel_package()
    nh = synth.PPG2MEL_NET_HP
    for name, B, T, wseed, sb, iseed, rseed in synth.PPG2MEL_MODEL_CASES:
        mm = pkg.MelDecoderMOLv2(num_speakers=1, spk_embed_dim=nh["spk_dim"], bottle_neck_feature_dim=nh["bnf_dim"],
                                 encoder_dim=nh["enc_dim"], encoder_downsample_rates=list(nh["downsample_rates"]))
        mm.load_state_dict(synth.ppg2mel_model_state(hp, nh, seed=wseed, stop_bias=sb), strict=False)
        mm.eval()
        bnf, lf0, spk = (torch.from_numpy(a) for a in synth.ppg2mel_inputs(B, T, seed=iseed))
        with torch.no_grad():
            x = mm.bnf_prenet(bnf.transpose(1, 2)).transpose(1, 2) + mm.pitch_convs(lf0.transpose(1, 2)).transpose(1, 2)
            spk_e = torch.nn.functional.normalize(spk).unsqueeze(1).expand(-1, x.size(1), -1)
            out[name + "_memory"] = mm.reduce_proj(torch.cat([x, spk_e], dim=-1)).numpy()
            torch.manual_seed(rseed)
            mel, melp, al = mm.inference(bnf, logf0_uv=lf0, spembs=spk)
        out[name + "_mel"], out[name + "_mel_postnet"], out[name + "_align"] = mel.numpy(), melp.numpy(), al.numpy()
    np.savez_compressed(os.path.join(HERE, "ppg2mel.npz"), torch_version=torch.__version__, **out)
    print("ppg2mel.npz", {k: v.shape for k, v in out.items()})


def vits():
    """models/synthesizer/models/vits.py Generator (the VITS decoder) of the REAL reference on the synthetic
    state (loguru / monotonic_align are stubbed: the module imports them at file scope, the decoder does not
    use them)."""
    import types
    for name in ("loguru", "monotonic_align"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "loguru":
                m.logger = types.SimpleNamespace(info=print, debug=print, warning=print, error=print)
            sys.modules[name] = m
    from models.synthesizer.models.vits import Generator
    out = {}
    for name, uic, frames, batch, use_g, seed in synth.VITS_CASES:
        h = dict(synth.VITS_DEC)
        h["upsample_initial_channel"] = uic
        g = Generator(h["initial_channel"], h["resblock"], h["resblock_kernel_sizes"], h["resblock_dilation_sizes"],
                      h["upsample_rates"], uic, h["upsample_kernel_sizes"], gin_channels=h["gin_channels"])
        g.load_state_dict(synth.vits_dec_state(h, seed=seed))
        g.eval()
        z, spk = synth.vits_latent(frames, batch, seed=seed + 1)
        with torch.no_grad():
            y = g(torch.from_numpy(z), g=torch.from_numpy(spk) if use_g else None)
        out[name] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "vits.npz"), torch_version=torch.__version__, **out)
    print("vits.npz", {k: v.shape for k, v in out.items()})


def wave():
    """encode_16bits (wavernn/audio.py:38-39) and save_wav (synthesizer/audio.py:12-15, through
    scipy.io.wavfile and back) run on seeded waveforms."""
    import tempfile
    from scipy.io import wavfile
    import models.vocoder.wavernn.audio as wa
    import models.synthesizer.audio as sa
    out = {}
    for dtype, n, peak, seed in synth.WAVE_CASES:
        x = synth.wave_input(dtype, n, peak, seed)
        key = f"{dtype}_n{n}_s{seed}"
        out["encode16_" + key] = wa.encode_16bits(x.copy())
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "x.wav")
            sa.save_wav(x.copy(), path, 16000)
            sr, pcm = wavfile.read(path)
        assert sr == 16000 and pcm.dtype == np.int16
        out["savewav_" + key] = pcm
    np.savez_compressed(os.path.join(HERE, "wave.npz"), numpy_version=np.__version__, **out)
    print("wave.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["gan", "gan_rb2", "gan24k", "wavernn", "maximum_path", "tacotron", "ppg2mel", "vits", "wave"]
    for w in which:
        if w in globals():
            globals()[w]()
