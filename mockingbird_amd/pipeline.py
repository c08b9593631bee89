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
exchanged once (sharding.gather_waveforms).  There is no collective inside a request."""
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
             steps=400, group=None, normalize=None, pcm16=None) -> List[np.ndarray]:
    """requests: [(texts, embed)] -> one float waveform per request, in request order, on every rank.

    normalize / pcm16 (optional): run gen_voice.py:41's peak normalisation and the
    PCM_16 conversion of the file writer on the device (vocoder/wave.py) before the waveform leaves the GPU; the
    result is int16 and the gather moves half the bytes.  The reference normalises AFTER inserting the breaks
    and trimming silence (gen_voice.py:40-41); zeros do not move the peak, so only a trim that removed the
    loudest sample would differ.

    synthesizer: object with synthesize_spectrograms / hparams.hop_size / sample_rate (the Synthesizer facade);
    vocoder: module or object with infer_waveform_batch(mels) -> (wavs, sample_rate) (hifigan / fregan facade)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lengths = [sum(len(t) for t in texts) for texts, _ in requests]
    mine = sharding.shard_indices(lengths, world, rank)
    local: List[np.ndarray] = []
    if mine:
        flat_texts, flat_embeds, owner = [], [], []
        for i in mine:
            texts, embed = requests[i]
            flat_texts += list(texts)
            flat_embeds += [embed] * len(texts)
            owner += [i] * len(texts)
        specs = synthesizer.synthesize_spectrograms(flat_texts, flat_embeds, style_idx=style_idx,
                                                    min_stop_token=min_stop_token, steps=steps)
        per_req = {i: [s for s, o in zip(specs, owner) if o == i] for i in mine}
        mels = [np.concatenate(per_req[i], axis=1) for i in mine]
        if normalize is not None or pcm16 is not None:
            wavs, _sr = vocoder.infer_waveform_batch(mels, normalize=normalize, pcm16=pcm16)
        else:
            wavs, _sr = vocoder.infer_waveform_batch(mels)
        hop, sr = synthesizer.hparams.hop_size, synthesizer.sample_rate
        local = [insert_breaks(w, [s.shape[1] for s in per_req[i]], hop, sr) for w, i in zip(wavs, mine)]
    wire = np.int16 if pcm16 is not None else np.float32
    gathered = sharding.gather_waveforms([np.asarray(w, wire) for w in local], group=group)
    # gather_waveforms returns rank-major order; put the requests back in their own order
    order = [i for r in range(world) for i in sharding.shard_indices(lengths, world, r)]
    out: List[np.ndarray] = [None] * len(requests)
    for w, i in zip(gathered, order):
        out[i] = w
    return out
