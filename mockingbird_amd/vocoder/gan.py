"""Device-side HiFi-GAN / Fre-GAN generator behind the C ABI (mb_gan_*).

Shared by ``vocoder.hifigan.inference`` and ``vocoder.fregan.inference`` --
the two reference facades are line-for-line identical glue around different
generators (models/vocoder/hifigan/inference.py:22-74,
models/vocoder/fregan/inference.py:22-74).
"""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import torch

from .. import _lib, weights

KIND_HIFIGAN, KIND_FREGAN = 0, 1


class GanGenerator:
    """Owns an ``mb_gan`` handle; ``forward`` mirrors ``Generator.forward`` /
    ``FreGAN.forward`` for a [B, 80, F] float32 device tensor."""

    def __init__(self, config: dict, state_dict: dict, kind: int, top_k: int = 4, dtype: str = "f32"):
        """dtype "f32": fp32 MFMA (1e-4 RMS parity path).  "f16": fp16 storage / fp32 accumulate on
        the fp16 matrix cores (BASELINE configs[4]; relative-RMS gate 5e-3)."""
        if dtype not in ("f32", "f16"):
            raise _lib.MbHipError(f"GanGenerator: dtype {dtype!r} (use 'f32' or 'f16')")
        if not torch.cuda.is_available():
            raise _lib.MbHipError("GAN vocoder: no MI355X visible; this build has no CPU path")
        self.cfg = weights.gan_config(config, kind, top_k)
        self.kind = kind
        self.dtype = dtype
        L = _lib.lib()
        ws = weights.gan_weight_list(state_dict, self.cfg)
        n = L.mb_gan_num_weights(C.byref(self.cfg))
        if n != len(ws):
            raise _lib.MbHipError(f"weight list has {len(ws)} tensors, ABI expects {n}")
        for i, w in enumerate(ws):
            want = L.mb_gan_weight_numel(C.byref(self.cfg), i)
            if w.numel() != want:
                raise _lib.MbHipError(f"weight {i}: {tuple(w.shape)} has {w.numel()} elements, expected {want}")
        arr = _lib.host_ptr_array(ws)
        h = C.c_void_p()
        _lib.check(L.mb_gan_create_ex(C.byref(self.cfg), arr, len(ws),
                                      _lib.MB_F16 if dtype == "f16" else _lib.MB_F32, C.byref(h)),
                   "mb_gan_create_ex")
        self._h = h
        self.hop = L.mb_gan_hop(h)
        self._ws = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mb_gan_destroy(h)
            except Exception:  # interpreter shutdown: module globals are already torn down
                pass
            self._h = None

    def forward(self, mel: torch.Tensor, chan_bias: torch.Tensor = None) -> torch.Tensor:
        if not mel.is_cuda:
            raise _lib.MbHipError("GanGenerator.forward needs a CUDA(HIP) tensor; there is no CPU path")
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        mel = mel.to(torch.float32).contiguous()
        B, M, F = mel.shape
        if M != self.cfg.num_mels:
            raise _lib.MbHipError(f"mel has {M} channels, model expects {self.cfg.num_mels}")
        if F == 0:
            return torch.empty(B, 1, 0, device=mel.device)
        L = _lib.lib()
        need = L.mb_gan_workspace_bytes(self._h, B, F)
        if self._ws is None or self._ws.numel() < need or self._ws.device != mel.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=mel.device)
        n_out = int(L.mb_gan_out_samples(self._h, F))  # F*hop, minus one per even-kernel stage of the 24 kHz variant
        if n_out <= 0:
            raise _lib.MbHipError(f"{F} mel frames are too few for this configuration")
        wav = torch.empty(B, 1, n_out, dtype=torch.float32, device=mel.device)
        if chan_bias is not None:  # [B, upsample_initial_channel] added to conv_pre's output (VITS cond)
            chan_bias = chan_bias.to(mel.device, torch.float32).contiguous()
            if tuple(chan_bias.shape) != (B, self.cfg.upsample_initial_channel):
                raise _lib.MbHipError(f"chan_bias must be {(B, self.cfg.upsample_initial_channel)}, got {tuple(chan_bias.shape)}")
        _lib.check(L.mb_gan_forward_ex(self._h, _lib.ptr(mel), B, F, _lib.ptr(wav), _lib.ptr(chan_bias), _lib.ptr(self._ws),
                                       self._ws.numel(), _lib.stream_ptr()), "mb_gan_forward_ex")
        return wav

    __call__ = forward

    def forward_ragged(self, mels) -> list:
        """A list of [80, F_i] mels of DIFFERENT lengths in ONE batched launch sequence (mb_gan_forward_ragged): item i
        equals forward(mels[i]) -- every conv treats positions beyond an item's length as its zero padding (the
        generators are not causal, so merely zero-padding the mel would change the tail).  Returns the list of
        [1, F_i * hop] device tensors."""
        if not mels:
            return []
        if self.cfg.interp_ups:  # 24 kHz variant: stage lengths are not multiples of the frame count
            return [self.forward(m.unsqueeze(0) if m.dim() == 2 else m)[0] for m in mels]
        dev = torch.device("cuda", torch.cuda.current_device())
        frames = [int(m.shape[-1]) for m in mels]
        if min(frames) <= 0:
            raise _lib.MbHipError("forward_ragged: empty mel")
        B, M, Fm = len(mels), self.cfg.num_mels, max(frames)
        if all(not (torch.is_tensor(m) and m.is_cuda) for m in mels):  # host mels: ONE padded array, one upload
            hb = np.zeros((B, M, Fm), dtype=np.float32)
            for i, m in enumerate(mels):
                m = np.asarray(m, dtype=np.float32)
                if m.shape[0] != M:
                    raise _lib.MbHipError(f"mel has {m.shape[0]} channels, model expects {M}")
                hb[i, :, :frames[i]] = m
            batch = torch.from_numpy(hb).to(dev)
        else:
            batch = torch.zeros(B, M, Fm, dtype=torch.float32, device=dev)
            for i, m in enumerate(mels):
                m = torch.as_tensor(m, dtype=torch.float32)
                if m.shape[0] != M:
                    raise _lib.MbHipError(f"mel has {m.shape[0]} channels, model expects {M}")
                batch[i, :, :frames[i]] = m.to(dev)
        d_frames = torch.tensor(frames, dtype=torch.int32, device=dev)
        L = _lib.lib()
        need = L.mb_gan_workspace_bytes(self._h, B, Fm)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        n_out = int(L.mb_gan_out_samples(self._h, Fm))
        wav = torch.empty(B, 1, n_out, dtype=torch.float32, device=dev)
        _lib.check(L.mb_gan_forward_ragged(self._h, _lib.ptr(batch), B, Fm, _lib.ptr(d_frames), _lib.ptr(wav), None,
                                           _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()), "mb_gan_forward_ragged")
        return [wav[i, :, :int(L.mb_gan_out_samples(self._h, f))] for i, f in enumerate(frames)]


class GanFacade:
    """Module-global-singleton semantics of the reference inference modules."""

    accepts_device_mels = True  # infer_waveform_batch takes device tensors as well as numpy mels (pipeline.gen_wavs keeps them in HBM)

    def __init__(self, kind: int, default_config: str, name: str):
        self.kind, self.default_config, self.name = kind, default_config, name
        self.generator = None
        self.output_sample_rate = None
        self._device = None

    def load_model(self, weights_fpath, config_fpath=None, verbose=True, dtype=None):
        """Reference signature (inference.py:22) plus an additive keyword: dtype "f32" | "f16"
        (default: env MBHIP_GAN_DTYPE, else "f32")."""
        import os
        dtype = dtype or os.environ.get("MBHIP_GAN_DTYPE", "f32")
        weights_fpath = Path(weights_fpath)
        if verbose:
            print(f"Building {self.name}")
        if config_fpath is None:  # inference.py:28-33
            found = list(weights_fpath.parent.rglob("*.json"))
            config_fpath = found[0] if found else self.default_config
        with open(config_fpath) as f:
            h = json.loads(f.read())
        self.output_sample_rate = h["sampling_rate"]
        torch.manual_seed(h["seed"])  # inference.py:39
        self._device = torch.device("cuda" if torch.cuda.is_available() else "cpu")  # inference.py:41-45
        if verbose:
            print(f"Loading '{weights_fpath}'")
        ckpt = torch.load(str(weights_fpath), map_location="cpu")
        self.generator = GanGenerator(h, ckpt["generator"], self.kind, dtype=dtype)
        if verbose:
            print("Complete.")

    def is_loaded(self):
        return self.generator is not None

    def infer_waveform(self, mel, progress_callback=None):
        if self.generator is None:
            raise Exception(f"Please load {self.name} in memory before using it")
        mel = torch.FloatTensor(mel).to(self._device)  # numpy or CPU tensor (run.py:90)
        y = self.generator(mel.unsqueeze(0))
        audio = y.squeeze().cpu().numpy()
        return audio, self.output_sample_rate

    def infer_waveform_batch(self, mels, progress_callback=None, normalize=None, pcm16=None, breaks=None,
                             break_hop=None, break_seconds=0.15, device_out=False, break_sample_rate=None):
        """Additive API: a list of (80, Fi) mels of any lengths -> list of waveforms, run as ragged
        batches (GanGenerator.forward_ragged: per-item lengths inside the kernels, each item equals its own run).

        The reference's host-side tail can run on the device, in gen_voice.py's order, before anything leaves HBM:
          breaks[i] = frames per sentence of item i -> cut at the sentence boundaries (frames * break_hop samples,
                      gen_voice.py:30-32) and put break_seconds of silence after every sentence (:33-34); the gap is
                      int(break_seconds * break_sample_rate) samples -- gen_voice.py:33 uses the SYNTHESIZER's
                      sample rate, which differs from this vocoder's for the 24 kHz generator (default: this
                      vocoder's output rate);
          normalize = 0.97 -> wav / abs(wav).max() * 0.97 (gen_voice.py:41);
          pcm16 = 'encode_16bits' | 'save_wav' (both pinned by goldens of the reference functions) | 'sndfile' (libsndfile's
                  PCM_16 of run.py:91, restated WITHOUT a pin: by name only) -> int16 PCM.
        device_out=True returns the device tensors themselves (for a device-to-device gather) instead of numpy."""
        if self.generator is None:
            raise Exception(f"Please load {self.name} in memory before using it")
        from . import wave
        out = [None] * len(mels)
        max_batch = 64  # items per launch sequence (workspace = batch x longest item)
        order = sorted(range(len(mels)), key=lambda i: -int(np.shape(mels[i])[1]))  # similar lengths share a batch
        for c0 in range(0, len(order), max_batch):
            idx = order[c0:c0 + max_batch]
            ys = self.generator.forward_ragged([mels[i] for i in idx])
            if pcm16 != "save_wav" and (breaks is not None or normalize is not None or pcm16 is not None) and len(idx) > 1:
                # the tails of the whole batch in two launches (wave.finish_batch); bit for bit the per-item calls below
                rs = wave.finish_batch(ys, [breaks[i] for i in idx] if breaks is not None else None, break_hop,
                                       break_sample_rate or self.output_sample_rate, break_seconds, normalize, pcm16)
                for k, i in enumerate(idx):
                    out[i] = rs[k] if device_out else rs[k].cpu().numpy()
                continue
            for k, i in enumerate(idx):
                r = ys[k].reshape(-1)
                if breaks is not None:
                    r = wave.insert_breaks(r, breaks[i], break_hop, break_sample_rate or self.output_sample_rate, break_seconds)
                if normalize is not None:
                    r = r.contiguous()
                    wave.peak_normalize_(r, normalize)
                if pcm16 is not None:
                    r = wave.pack_pcm16(r, pcm16)
                out[i] = r if device_out else r.cpu().numpy()
        return out, self.output_sample_rate
