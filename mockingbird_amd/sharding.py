"""Utterance-level data parallelism: one process per GPU, units =
utterances, no intra-utterance collective.  The only exchange is the gather of finished
waveforms (lengths first, then zero-padded float32 waveforms) over RCCL/xGMI -- tens of MB,
latency bound, so one all_gather each (a ring buys nothing at this size).

Reference precedent: data_parallel_workaround, models/synthesizer/utils/__init__.py:7-21
(batch-dim scatter/gather in one process); the reference has no multi-GPU inference."""
from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(lengths: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Length-sorted round-robin deal: rank r gets items r, r+W, r+2W, ... of the
    descending-length order, so padded work is balanced (the whole-batch stop rule,
    tacotron.py:275, makes a batch as slow as its longest item)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return order[rank::world_size]


def gather_waveforms(local: List[np.ndarray], device=None, group=None) -> List[np.ndarray]:
    """All ranks contribute a list of waveforms (float32, or int16 PCM: half the bytes on the wire);
    every rank returns the concatenated list in rank order.  Works with nccl(=RCCL) on GPUs and gloo on
    CPU.  The sample type is that of the first local waveform (int16 stays int16, anything else -> float32)
    and must be the same on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(local)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device(device) if device is not None else torch.device("cuda" if backend == "nccl" else "cpu")
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    max_items = int(max(int(c.item()) for c in counts))
    lens = torch.zeros(max_items, dtype=torch.int64, device=dev)
    for i, w in enumerate(local):
        lens[i] = len(w)
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens, group=group)
    max_len = int(max(int(l.max().item()) if l.numel() else 0 for l in all_lens))
    is_pcm = torch.tensor([int(bool(local) and np.asarray(local[0]).dtype == np.int16), int(bool(local))],
                          dtype=torch.int64, device=dev)
    kinds = [torch.zeros_like(is_pcm) for _ in range(world)]
    dist.all_gather(kinds, is_pcm, group=group)
    votes = {int(k[0].item()) for k in kinds if int(k[1].item())}  # ranks with no waveform do not vote
    if len(votes) > 1:
        raise ValueError("gather_waveforms: ranks disagree on the sample type (int16 vs float)")
    np_dt, t_dt = (np.int16, torch.int16) if votes == {1} else (np.float32, torch.float32)
    buf = torch.zeros(max_items, max_len, dtype=t_dt, device=dev)
    for i, w in enumerate(local):
        buf[i, :len(w)] = torch.from_numpy(np.ascontiguousarray(w, dtype=np_dt)).to(dev)
    # int16 travels as bytes: gloo has no int16 collectives, and the byte view costs nothing
    wire = buf.view(torch.uint8) if t_dt == torch.int16 else buf
    all_bufs = [torch.zeros_like(wire) for _ in range(world)]
    dist.all_gather(all_bufs, wire, group=group)
    out = []
    for r in range(world):
        b = all_bufs[r].view(t_dt).cpu().numpy()
        for i in range(int(counts[r].item())):
            out.append(b[i, :int(all_lens[r][i].item())].copy())
    return out
