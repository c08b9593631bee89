"""End-to-end cloned-voice generation for a BATCH of requests, one process per GPU
(BASELINE.json configs[3]: Tacotron2 + HiFi-GAN, utterances sharded over the GPUs of a node).

A request = the reference's gen_one_wav(synthesizer, in_fpath, embed, texts, ...) call
(gen_voice.py:15-34): several sentences spoken with one speaker embedding ->
  specs = synthesizer.synthesize_spectrograms(texts, [embed]*len(texts), style_idx=-1, min_stop_token=4, steps=400)
  wav   = vocoder.infer_waveform(np.concatenate(specs, axis=1))
  wav   = sentences separated by 0.15 s of silence (gen_voice.py:30-34)
The trimming / peak normalisation / file writing that follow (gen_voice.py:40-47) use the speaker-encoder
package and soundfile and stay on the reference's CPU path (north_star).

Sharding: requests are dealt to ranks by sharding.shard_indices (length-sorted
round robin), each rank runs ITS requests through both models on its GPU, and the finished waveforms are
gathered once to rank 0 (sharding.gather_waveforms).  There is no collective inside a request."""
import inspect
from typing import List, Sequence, Tuple

import numpy as np
import torch.distributed as dist

from . import sharding

BREAK_SECONDS = 0.15  # gen_voice.py:33


def insert_breaks(wav: np.ndarray, frames_per_sentence: Sequence[int], hop_size: int, sample_rate: int) -> np.ndarray:
    """gen_voice.py:30-34: cut the vocoded waveform at the sentence boundaries (frames * hop_size samples)
    and put 0.15 s of zeros after every sentence."""
    b_ends = np.cumsum(np.array(frames_per_sentence) * hop_size)
    b_starts = np.concatenate(([0], b_ends[:-1]))
    wavs = [wav[start:end] for start, end in zip(b_starts, b_ends)]
    # (float waveforms: float64 zeros, so the result is float64 as in the reference; int16 PCM stays int16)
    gap = np.zeros(int(BREAK_SECONDS * sample_rate), dtype=np.int16 if wav.dtype == np.int16 else np.float64)
    return np.concatenate([piece for w in wavs for piece in (w, gap)])


def gen_wavs(synthesizer, vocoder, requests: List[Tuple[List[str], np.ndarray]], *, style_idx=-1, min_stop_token=4,
             steps=400, group=None, normalize=None, pcm16=None, dst=0, chunk_size=None, timings=None) -> List[np.ndarray]:
    """requests: [(texts, embed)] -> one waveform per request, in request order, on rank `dst` (every other rank
    returns []; dst=None: on every rank).

    The whole tail of gen_voice.py:30-41 that does not need the speaker-encoder package runs on the device inside
    the vocoder facade (vocoder/wave.py): the 0.15 s sentence breaks, then -- optionally -- normalize (gen_voice.py:41's
    peak normalisation) and pcm16 (the PCM_16 conversion of the file writer, run.py:91).  The finished waveforms
    stay in HBM until the gather, which moves them device-to-device to rank `dst` (int16: half the bytes).  The
    reference trims silence between break insertion and normalisation (gen_voice.py:40, webrtcvad on the CPU); zeros
    do not move the peak, so only a trim that removed the loudest sample would change the normalised result.

    synthesizer: object with synthesize_spectrograms / hparams.hop_size / sample_rate (the Synthesizer facade);
    vocoder: module or object with infer_waveform_batch(mels, normalize= | peak_normalize=, pcm16=, breaks=, break_hop=,
    break_sample_rate=, device_out=) -> (wavs, sample_rate): the hifigan / fregan facades (`normalize` = peak
    normalisation) or the wavernn facade (`peak_normalize`; its `normalize` keeps meaning the mel scaling).
    timings: optional dict (measurement only, bench.py): receives this rank's seconds in "synthesizer", "vocoder" (device-synchronised)
    and "gather"."""
    import time
    t_syn = t_voc = 0.0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lengths = [sum(len(t) for t in texts) for texts, _ in requests]
    mine = sharding.shard_indices(lengths, world, rank)
    local = []
    if mine:
        flat_texts, flat_embeds, owner = [], [], []
        for i in mine:
            texts, embed = requests[i]
            flat_texts += list(texts)
            flat_embeds += [embed] * len(texts)
            owner += [i] * len(texts)
        skw = {"chunk_size": chunk_size} if chunk_size else {}  # additive keyword of the Synthesizer facade (utterances per decoder loop)
        sparams = inspect.signature(synthesizer.synthesize_spectrograms).parameters
        on_device = "device_out" in sparams and getattr(vocoder, "accepts_device_mels", False)
        if on_device:  # the spectrograms stay in HBM between the two facades
            skw["device_out"] = True
        t0 = time.perf_counter()
        specs = synthesizer.synthesize_spectrograms(flat_texts, flat_embeds, style_idx=style_idx,
                                                    min_stop_token=min_stop_token, steps=steps, **skw)
        t_syn = time.perf_counter() - t0
        per_req = {i: [s for s, o in zip(specs, owner) if o == i] for i in mine}
        if on_device:
            import torch
            mels = [per_req[i][0] if len(per_req[i]) == 1 else torch.cat(per_req[i], dim=1) for i in mine]
        else:
            mels = [np.concatenate(per_req[i], axis=1) for i in mine]
        kw = dict(pcm16=pcm16, breaks=[[s.shape[1] for s in per_req[i]] for i in mine],
                  break_hop=synthesizer.hparams.hop_size, device_out=True)
        params = inspect.signature(vocoder.infer_waveform_batch).parameters
        if "break_sample_rate" in params:  # gen_voice.py:33: the gap is 0.15 s at the SYNTHESIZER's sample rate
            kw["break_sample_rate"] = synthesizer.sample_rate
        if "peak_normalize" in params:  # WaveRNN facade: its `normalize` is infer_waveform's mel scaling (inference.py:60-61)
            kw["peak_normalize"] = normalize
        else:
            kw["normalize"] = normalize
        t0 = time.perf_counter()
        local, _sr = vocoder.infer_waveform_batch(mels, **kw)
        if timings is not None:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t_voc = time.perf_counter() - t0
    t0 = time.perf_counter()
    gathered = sharding.gather_waveforms(local, group=group, dst=dst)
    if timings is not None:
        timings.update(synthesizer=t_syn, vocoder=t_voc, gather=time.perf_counter() - t0)
    if dst is not None and rank != dst:
        return []
    # gather_waveforms returns rank-major order; put the requests back in their own order
    order = [i for r in range(world) for i in sharding.shard_indices(lengths, world, r)]
    out: List[np.ndarray] = [None] * len(requests)
    for w, i in zip(gathered, order):
        out[i] = w
    return out
