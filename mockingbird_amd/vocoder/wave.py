"""Waveform wire format on device (include/mbhip.h section 7).

The reference finishes a generated waveform in numpy on the host: peak-normalise to 0.97
(gen_voice.py:41, control/toolbox/__init__.py:313) and convert to 16-bit PCM when writing
(run.py:91 sf.write(.., "PCM_16"); models/synthesizer/audio.py:12-15 save_wav;
models/vocoder/wavernn/audio.py:38-39 encode_16bits).  Here both run on the vocoder's output
while it is still in HBM, so the D2H copy (and the multi-GPU gather, pipeline.gen_wavs) moves
2 bytes per sample instead of 4 or 8.  fp32 and float64 tensors are processed in their own type,
like the numpy arrays they replace.  There is no CPU path."""
import ctypes as C

import torch

from .. import _lib

_MODES = {"sndfile": _lib.MB_PCM16_SNDFILE, "encode_16bits": _lib.MB_PCM16_ENCODE16, "save_wav": _lib.MB_PCM16_SAVE_WAV}


def _dtype(wav: torch.Tensor) -> int:
    if not wav.is_cuda:
        raise _lib.MbHipError("waveform post-processing needs a CUDA(HIP) tensor; there is no CPU path")
    if wav.dtype == torch.float32:
        return _lib.MB_F32
    if wav.dtype == torch.float64:
        return _lib.MB_F64
    raise _lib.MbHipError(f"waveform must be float32 or float64, got {wav.dtype}")


def _workspace(dev) -> torch.Tensor:
    return torch.empty(_lib.lib().mb_wave_workspace_bytes(), dtype=torch.uint8, device=dev)


def peak_normalize_(wav: torch.Tensor, target: float = 0.97) -> torch.Tensor:
    """In place: wav = wav / abs(wav).max() * target (gen_voice.py:41).  One waveform per call
    (the maximum is taken over the whole tensor)."""
    dt = _dtype(wav)
    if not wav.is_contiguous():
        raise _lib.MbHipError("peak_normalize_ needs a contiguous tensor")
    if wav.numel() == 0:
        raise ValueError("zero-size array to reduction operation maximum which has no identity")  # np.abs(wav).max()
    ws = _workspace(wav.device)
    _lib.check(_lib.lib().mb_wave_peak_normalize(_lib.ptr(wav), dt, wav.numel(), float(target), _lib.ptr(ws), ws.numel(),
                                                 _lib.stream_ptr()), "mb_wave_peak_normalize")
    return wav


def pack_pcm16(wav: torch.Tensor, mode: str = "encode_16bits") -> torch.Tensor:
    """float waveform -> int16 tensor of the same shape.  mode: 'encode_16bits' (wavernn/audio.py:38-39; the default),
    'save_wav' (synthesizer/audio.py:12-15, rescales the peak to 32767) -- both pinned bit-exactly by goldens from the
    reference functions -- or 'sndfile' (PCM_16 as libsndfile writes it, run.py:91): a restatement WITHOUT a pin
    (libsndfile is neither vendored by the reference nor installed here), so it is never a default and has to be asked
    for by name."""
    if mode not in _MODES:
        raise ValueError(f"mode must be one of {sorted(_MODES)}, got {mode!r}")
    dt = _dtype(wav)
    wav = wav.contiguous()
    out = torch.empty(wav.shape, dtype=torch.int16, device=wav.device)
    if wav.numel() == 0:
        if mode == "save_wav":
            raise ValueError("zero-size array to reduction operation maximum which has no identity")
        return out
    ws = _workspace(wav.device)
    _lib.check(_lib.lib().mb_wave_pack_pcm16(_lib.ptr(wav), dt, wav.numel(), _MODES[mode], _lib.ptr(out), _lib.ptr(ws),
                                             ws.numel(), _lib.stream_ptr()), "mb_wave_pack_pcm16")
    return out


def insert_breaks(wav: torch.Tensor, frames_per_sentence, hop_size: int, sample_rate: int, seconds: float = 0.15) -> torch.Tensor:
    """gen_voice.py:30-34 on the device: cut the vocoded waveform at the sentence boundaries (frames * hop_size
    samples -- slices past the end clip, as numpy's do: the reference cuts with the synthesizer's hop 256 while the
    16 kHz vocoders emit 200 samples per frame, SURVEY finding 5) and append `seconds` of zeros to every sentence.
    Pure device-memory plumbing: one output allocation, one copy per sentence; the waveform never visits the host."""
    if not wav.is_cuda:
        raise _lib.MbHipError("insert_breaks needs a CUDA(HIP) tensor; there is no CPU path")
    wav = wav.reshape(-1)
    n, gap = wav.numel(), int(seconds * sample_rate)
    ends, e = [], 0
    for f in frames_per_sentence:
        e += int(f) * int(hop_size)
        ends.append(e)
    starts = [0] + ends[:-1]
    pieces = [(min(s, n), min(t, n)) for s, t in zip(starts, ends)]
    total = sum(t - s for s, t in pieces) + gap * len(pieces)
    out = torch.zeros(total, dtype=wav.dtype, device=wav.device)
    o = 0
    for s, t in pieces:
        out[o:o + (t - s)] = wav[s:t]
        o += (t - s) + gap
    return out



def finish_batch(wavs, breaks, hop_size, sample_rate, seconds=0.15, normalize=None, pcm16=None):
    """insert_breaks -> peak_normalize_ -> pack_pcm16 for a LIST of fp32 device waveforms (views of one vocoder output are the usual
    case) in two launches (mb_wave_finish_batch) instead of half a dozen per item; every item's samples are bit for bit those of the
    one-waveform functions above.  breaks[i] = frames per sentence of item i (None: no breaks); normalize / pcm16 as there
    ('save_wav' is not offered here: it needs the peak of the normalised signal).  Returns the list of result tensors (views of one
    flat buffer): float32, or int16 with pcm16."""
    if pcm16 is not None and pcm16 not in ("encode_16bits", "sndfile"):
        raise ValueError(f"finish_batch: pcm16 must be 'encode_16bits' or 'sndfile', got {pcm16!r}")
    if not wavs:
        return []
    dev = wavs[0].device
    if any((not w.is_cuda) or w.dtype != torch.float32 for w in wavs):
        raise _lib.MbHipError("finish_batch needs fp32 CUDA(HIP) tensors; there is no CPU path")
    flat = [w.reshape(-1) for w in wavs]
    if any(not f.is_contiguous() for f in flat):
        flat = [f.contiguous() for f in flat]
    base = min(f.data_ptr() for f in flat)
    gap = int(seconds * sample_rate)
    table, spans, o = [], [], 0
    for i, f in enumerate(flat):
        n, src0 = f.numel(), (f.data_ptr() - base) // 4
        if breaks is None:
            pieces = [(0, n)]
        else:
            ends, e = [], 0
            for fr in breaks[i]:
                e += int(fr) * int(hop_size)
                ends.append(e)
            starts = [0] + ends[:-1]
            pieces = [(min(s, n), min(t, n)) for s, t in zip(starts, ends)]
        start = o
        for s, t in pieces:
            if t > s:
                table += [src0 + s, o, t - s, i]
            o += t - s
            if breaks is not None and gap > 0:  # the silence behind the sentence: a piece of zeros (normalised like the rest)
                table += [-1, o, gap, i]
                o += gap
        spans.append((start, o))
    out = torch.empty(o, dtype=torch.int16 if pcm16 is not None else torch.float32, device=dev)
    L = _lib.lib()
    n_pieces = len(table) // 4
    d_table = torch.tensor(table if table else [0, 0, 0, 0], dtype=torch.int64).to(dev)
    ws = torch.empty(max(256, L.mb_wave_finish_batch_workspace_bytes(len(flat))), dtype=torch.uint8, device=dev)
    _lib.check(L.mb_wave_finish_batch(C.c_void_p(base), _lib.ptr(d_table), n_pieces, len(flat), o,
                                      float(normalize) if normalize is not None else -1.0, -1 if pcm16 is None else _MODES[pcm16],
                                      _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "mb_wave_finish_batch")
    # (the views keep `out` alive; `flat` / the table must outlive the launches: same stream, released by the caching allocator in order)
    return [out[a:b] for a, b in spans]
