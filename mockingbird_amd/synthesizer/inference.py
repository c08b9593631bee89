"""Drop-in for models/synthesizer/inference.py:15-185: same ``Synthesizer`` class and
``synthesize_spectrograms`` signature/returns; the Tacotron decoder loop and CBHG postnet
(tacotron.py:264-283) run as HIP kernels through libmbhip.so (mb_taco_decode)."""
import ctypes as C
from pathlib import Path
from typing import List, Union

import numpy as np
import torch

from .. import _lib, weights
from .hparams import hparams
from .text import symbols, text_to_sequence, to_pinyin


fresh_seed = _lib.fresh_seed


class TacotronDevice:
    """Weights resident in HBM inside an ``mb_taco`` handle: text encoder, global style tokens,
    decoder loop and postnet all run as HIP kernels (no torch op touches the data path)."""

    def __init__(self, state_dict, device, r=None, dropout=None):
        if not torch.cuda.is_available():
            raise _lib.MbHipError("Tacotron: no MI355X visible; this build has no CPU path")
        self.device = device
        self.cfg = weights.taco_config(state_dict, r=r, dropout=hparams.tts_dropout if dropout is None else dropout)
        self.r = self.cfg.r
        L = _lib.lib()
        ws = weights.taco_weight_list(state_dict, self.cfg)
        n = L.mb_taco_num_weights(C.byref(self.cfg))
        if n != len(ws):
            raise _lib.MbHipError(f"weight list has {len(ws)} tensors, ABI expects {n}")
        for i, w in enumerate(ws):
            want = L.mb_taco_weight_numel(C.byref(self.cfg), i)
            if w.numel() != want:
                raise _lib.MbHipError(f"weight {i}: {tuple(w.shape)} has {w.numel()} elements, expected {want}")
        h = C.c_void_p()
        _lib.check(L.mb_taco_create(C.byref(self.cfg), _lib.host_ptr_array(ws), len(ws), C.byref(h)), "mb_taco_create")
        self._h = h
        self._ws = None
        self._ews = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mb_taco_destroy(h)
            except Exception:  # interpreter shutdown: module globals are already torn down
                pass
            self._h = None

    def decode(self, memory, memory_proj, chars, steps, min_stop_token, dropout=None, seed=None):
        """HIP decoder loop + postnet.  Returns (mel_outputs, linear, attn) trimmed to the frames produced.
        seed=None draws the Philox key of the always-on PreNet dropout from torch's global generator (the
        reference's F.dropout consumes that generator, pre_net.py:23,26: torch.manual_seed steers both)."""
        seed = fresh_seed() if seed is None else seed
        B, T, _ = memory.shape
        r = self.r
        max_steps = (steps + r - 1) // r * r  # range(0, steps, r) emits r frames per iteration (tacotron.py:264)
        dev = memory.device
        L = _lib.lib()
        need = L.mb_taco_workspace_bytes(self._h, B, T, max_steps)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        mel = torch.empty(B, self.cfg.n_mels, max_steps, device=dev)
        lin = torch.empty(B, self.cfg.n_mels, max_steps, device=dev)
        attn = torch.empty(B, max_steps // r, T, device=dev)
        chars32 = chars.to(torch.int32).contiguous()
        if dropout is not None:
            dropout = dropout.to(dev, torch.float32).contiguous()
            if tuple(dropout.shape[1:]) != (2, B, 2 * self.cfg.decoder_dims) or dropout.shape[0] < max_steps // r:
                raise _lib.MbHipError(f"dropout masks must be [>= {max_steps // r}, 2, {B}, {2 * self.cfg.decoder_dims}]")
        nf = C.c_int(0)
        _lib.check(L.mb_taco_decode(self._h, _lib.ptr(memory), _lib.ptr(memory_proj), _lib.ptr(chars32), B, T, max_steps,
                                    float(min_stop_token), _lib.ptr(dropout), int(seed), _lib.ptr(mel), _lib.ptr(lin),
                                    _lib.ptr(attn), C.byref(nf), _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()),
                   "mb_taco_decode")
        F = nf.value
        ms, its = C.c_float(), C.c_int()
        if L.mb_taco_last_loop_ms(self._h, C.byref(ms), C.byref(its)) == 0:  # production-dims loop only
            self.last_loop_ms, self.last_loop_iterations = ms.value, its.value
            self.last_loop_launches_per_iteration = L.mb_taco_last_loop_form(self._h)  # 4: fused front + folded rnn_input, 5: fused front, 7: one launch each
            self.last_loop_f16_products = bool(L.mb_taco_last_loop_f16(self._h) == 1)  # split-f16 products on the K >= 1024 tiles
        pms = C.c_float()
        self.last_postnet_ms = pms.value if L.mb_taco_last_postnet_ms(self._h, C.byref(pms)) == 0 else None  # CBHG + post_proj (HIP events)
        return mel[:, :, :F], lin[:, :, :F], attn[:, :F // r]

    def encode(self, chars, speaker_embedding, style_idx=0, enc_masks=None, seed=None):
        """Front half of Tacotron.forward (tacotron.py:234-255) in HIP: -> (encoder_seq [B,T,P],
        encoder_seq_proj [B,T,D]).  enc_masks: optional [2, B, T, encoder_dims] PreNet keep masks."""
        seed = fresh_seed() if seed is None else seed
        if not self.cfg.has_encoder:
            raise _lib.MbHipError("checkpoint has no encoder weights")
        dev = chars.device
        B, T = chars.shape
        spk = speaker_embedding.to(dev, torch.float32).contiguous()
        L = _lib.lib()
        need = L.mb_taco_encode_workspace_bytes(self._h, B, T)
        if self._ews is None or self._ews.numel() < need or self._ews.device != dev:
            self._ews = torch.empty(need, dtype=torch.uint8, device=dev)
        memory = torch.empty(B, T, self.cfg.project_dims, device=dev)
        memory_proj = torch.empty(B, T, self.cfg.decoder_dims, device=dev)
        chars32 = chars.to(torch.int32).contiguous()
        if enc_masks is not None:
            enc_masks = torch.as_tensor(enc_masks).to(dev, torch.float32).contiguous()
            if tuple(enc_masks.shape) != (2, B, T, self.cfg.encoder_dims):
                raise _lib.MbHipError(f"encoder masks must be {(2, B, T, self.cfg.encoder_dims)}, got {tuple(enc_masks.shape)}")
        _lib.check(L.mb_taco_encode(self._h, _lib.ptr(chars32), _lib.ptr(spk), int(style_idx), B, T,
                                    _lib.ptr(enc_masks), int(seed) ^ 0x5EED, _lib.ptr(memory), _lib.ptr(memory_proj),
                                    _lib.ptr(self._ews), self._ews.numel(), _lib.stream_ptr()), "mb_taco_encode")
        return memory, memory_proj

    def generate(self, chars, speaker_embedding, steps=2000, style_idx=0, min_stop_token=5, enc_masks=None,
                 dropout=None, seed=None):
        """Tacotron.generate (tacotron.py:295-298) -> (mel_outputs, linear, attn_scores)."""
        seed = fresh_seed() if seed is None else seed
        memory, memory_proj = self.encode(chars, speaker_embedding, style_idx, enc_masks, seed)
        return self.decode(memory, memory_proj, chars, steps, min_stop_token, dropout, seed)


class Synthesizer:
    sample_rate = hparams.sample_rate
    hparams = hparams

    def __init__(self, model_fpath: Path, verbose=True):
        self.model_fpath = Path(model_fpath)
        self.verbose = verbose
        # inference.py:30-33 picks cuda when available; the model itself is built lazily by load(), which is
        # where a missing GPU becomes an error (there is no CPU path to fall back to)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if self.verbose:
            print("Synthesizer using device:", self.device)
        self._model = None

    def is_loaded(self):
        return self._model is not None

    def load(self):
        found = list(self.model_fpath.parent.rglob("*.json"))  # inference.py:47-50
        if len(found) > 0 and found[0].exists():
            hparams.loadJson(found[0])
        checkpoint = torch.load(str(self.model_fpath), map_location="cpu")
        state = checkpoint["model_state"] if "model_state" in checkpoint else checkpoint["model"]  # base.py:51-54
        self._model = TacotronDevice(state, self.device)  # raises MbHipError without a GPU: there is no CPU path
        self._step = int(state["step"].item()) if "step" in state else 0
        if self.verbose:
            print("Loaded synthesizer \"%s\" trained to step %d" % (self.model_fpath.name, self._step))

    def synthesize_spectrograms(self, texts: List[str], embeddings: Union[np.ndarray, List[np.ndarray]],
                                return_alignments=False, style_idx=0, min_stop_token=5, steps=2000, chunk_size=None, device_out=False):
        """Reference signature (inference.py:75) plus two additive keywords: device_out = the (trimmed) spectrograms stay in HBM as
        device tensors for a vocoder facade on the same GPU (pipeline.gen_wavs) instead of going through numpy; chunk_size = utterances per decoder loop
        (default hparams.synthesis_batch_size = 16, hparams.py:56, as the reference chunks).  The encoder runs on the
        device too, so nothing limits a chunk to 16: 32 utterances in ONE loop cost about what 16 do (the loop is
        latency-bound), two chunks of 16 cost twice that.  The batch-wide stop rule (tacotron.py:275) then spans the
        larger chunk, exactly as it would in the reference with a larger synthesis_batch_size."""
        if not self.is_loaded():
            self.load()
        print("Read " + str(texts))
        texts = to_pinyin(texts)
        print("Synthesizing " + str(texts))
        inputs = [text_to_sequence(text, hparams.tts_cleaner_names) for text in texts]
        specs, alignments = self.synthesize_from_tokens(inputs, embeddings, style_idx, min_stop_token, steps, chunk_size=chunk_size,
                                                        device_out=device_out)
        if self.verbose:
            print("\n\nDone.\n")
        return (specs, alignments) if return_alignments else specs

    def synthesize_from_tokens(self, inputs, embeddings, style_idx=0, min_stop_token=5, steps=2000, enc_masks=None,
                               dropout=None, seed=None, chunk_size=None, device_out=False):
        """inference.py:104-139 from token id sequences (benchmarks feed these directly).  seed=None: every
        chunk draws its own RNG key from torch's global generator (fresh_seed); an explicit seed is advanced
        per chunk so that no two chunks share dropout masks."""
        if not self.is_loaded():
            self.load()
        if not isinstance(embeddings, list):
            embeddings = [embeddings]
        bs = int(chunk_size) if chunk_size else hparams.synthesis_batch_size
        if bs < 1:
            raise ValueError(f"chunk_size must be >= 1, got {chunk_size}")
        specs, alignments = [], None
        for i in range(0, len(inputs), bs):
            batch = inputs[i:i + bs]
            if self.verbose:
                print(f"\n| Generating {i // bs + 1}/{(len(inputs) + bs - 1) // bs}")
            max_text_len = max(len(t) for t in batch)
            chars = np.stack([pad1d(t, max_text_len) for t in batch])
            speaker_embeds = np.stack(embeddings[i:i + bs])
            chars = torch.tensor(chars).long().to(self.device)
            speaker_embeddings = torch.tensor(speaker_embeds).float().to(self.device)
            chunk_seed = fresh_seed() if seed is None else int(seed) + 0x9E3779B97F4A7C15 * (i // bs) & (2 ** 63 - 1)
            _, mels, alignments = self._model.generate(chars, speaker_embeddings, style_idx=style_idx,
                                                       min_stop_token=min_stop_token, steps=steps,
                                                       enc_masks=enc_masks, dropout=dropout, seed=chunk_seed)
            if device_out:  # the same trim from the per-frame maxima (B x T floats cross PCIe instead of B x 80 x T), views of the device tensor
                mels = mels.detach()
                fmax = mels.amax(dim=1).cpu().numpy()
                for b in range(mels.shape[0]):
                    n = fmax.shape[1]
                    while fmax[b, n - 1] < hparams.tts_stop_threshold:  # (raises IndexError on an all-silent mel, as the reference does)
                        n -= 1
                        if n == 0:
                            raise IndexError("index -1 is out of bounds for axis 1 with size 0")
                    specs.append(mels[b, :, :n])
                continue
            mels = mels.detach().cpu().numpy()
            for m in mels:
                while np.max(m[:, -1]) < hparams.tts_stop_threshold:  # trim silence (inference.py:136-137)
                    m = m[:, :-1]
                specs.append(m)
        return specs, alignments


    # ---- host-side signal utilities of the reference class (inference.py:144-181).  north_star keeps the
    # speaker-encoder front end, mel extraction and Griffin-Lim on the reference's CPU path: these statics
    # delegate to the reference's own modules (models/synthesizer/audio.py, utils/logmmse.py, librosa), which
    # stay importable next to the aliased facade (mockingbird_amd.install() replaces only inference.py).
    @staticmethod
    def _reference_audio():
        try:
            from models.synthesizer import audio  # the reference checkout on sys.path
        except Exception as e:  # pragma: no cover - depends on the caller's environment
            raise ImportError("Synthesizer.load_preprocess_wav / make_spectrogram / griffin_lim run the reference's CPU "
                              "code (models/synthesizer/audio.py, librosa): put the MockingBird checkout and its "
                              f"requirements on sys.path ({e})") from e
        return audio

    @staticmethod
    def load_preprocess_wav(fpath):
        """inference.py:144-159: load at hparams.sample_rate, rescale, logMMSE denoise from the first/last 0.15 s."""
        Synthesizer._reference_audio()
        import librosa
        from utils import logmmse
        wav = librosa.load(path=str(fpath), sr=hparams.sample_rate)[0]
        if hparams.rescale:
            wav = wav / np.abs(wav).max() * hparams.rescaling_max
        if len(wav) > hparams.sample_rate * (0.3 + 0.1):
            noise_wav = np.concatenate([wav[:int(hparams.sample_rate * 0.15)], wav[-int(hparams.sample_rate * 0.15):]])
            profile = logmmse.profile_noise(noise_wav, hparams.sample_rate)
            wav = logmmse.denoise(wav, profile)
        return wav

    @staticmethod
    def make_spectrogram(fpath_or_wav: Union[str, Path, np.ndarray]):
        """inference.py:161-173: mel spectrogram as fed to the synthesizer in training."""
        audio = Synthesizer._reference_audio()
        if isinstance(fpath_or_wav, (str, Path)):
            wav = Synthesizer.load_preprocess_wav(fpath_or_wav)
        else:
            wav = fpath_or_wav
        return audio.melspectrogram(wav, hparams).astype(np.float32)

    @staticmethod
    def griffin_lim(mel):
        """inference.py:175-181: Griffin-Lim inversion with the parameters in hparams."""
        return Synthesizer._reference_audio().inv_mel_spectrogram(mel, hparams)


def pad1d(x, max_len, pad_value=0):
    return np.pad(x, (0, max_len - len(x)), mode="constant", constant_values=pad_value)
