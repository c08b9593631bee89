"""Utterance-level data parallelism: one process per GPU, units =
utterances, no intra-utterance collective.  The only exchange is the gather of finished
waveforms to rank 0 (a small header all_gather, then ONE flat buffer per rank gathered device-to-device) over
RCCL/xGMI -- tens of MB, latency bound: every peer sends its slice to rank 0 over its own point-to-point link.

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


def _as_tensor(w, dev, np_dt, t_dt):
    if isinstance(w, torch.Tensor):
        return w.detach().reshape(-1).to(dev, t_dt)
    return torch.from_numpy(np.ascontiguousarray(w, dtype=np_dt).reshape(-1)).to(dev)


def gather_waveforms(local, device=None, group=None, dst=0) -> List[np.ndarray]:
    """Every rank contributes a list of finished waveforms; rank `dst` returns the concatenated list in rank
    order (as numpy arrays), every other rank returns [] -- north_star: "RCCL over xGMI only to gather finished
    waveforms".  dst=None returns the full list on every rank (all_gather instead of gather).

    The items may be numpy arrays or torch tensors that already live on the GPU: they are packed into ONE flat
    device buffer per rank and travel device-to-device (nccl = RCCL on GPUs, gloo on CPU); only rank `dst` copies
    the result to the host, once.  Sample type = that of the first local waveform (int16 PCM stays int16 -- half
    the bytes on the wire --, float64 (WaveRNN) stays float64, anything else travels as float32) and must agree across ranks."""
    KINDS = {0: (np.float32, torch.float32), 1: (np.int16, torch.int16), 2: (np.float64, torch.float64)}

    def kind(w):  # wire type code: int16 PCM and float64 (WaveRNN.generate's dtype) travel as they are, anything else as float32
        dt = w.dtype if isinstance(w, torch.Tensor) else np.asarray(w).dtype
        return 1 if dt in (torch.int16, np.dtype(np.int16)) else 2 if dt in (torch.float64, np.dtype(np.float64)) else 0

    def host(w, np_dt):
        return w.detach().cpu().numpy().astype(np_dt, copy=False) if isinstance(w, torch.Tensor) else np.asarray(w, np_dt)

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [host(w, KINDS[kind(w)][0]) for w in local]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
        if backend == "nccl" else torch.device("cpu")
    # ---- header: item count, sample type, per-item lengths (one small all_gather: every rank sizes its buffer) ----
    MAXI = 1 + max(int(x) for x in _all_gather_ints([len(local)], dev, group))
    head = torch.zeros(2 + MAXI, dtype=torch.int64, device=dev)
    head[0] = len(local)
    head[1] = kind(local[0]) if local else -1
    for i, w in enumerate(local):
        head[2 + i] = int(w.numel() if isinstance(w, torch.Tensor) else np.asarray(w).size)
    heads = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    heads = [h.cpu() for h in heads]
    votes = {int(h[1]) for h in heads if int(h[1]) >= 0}  # ranks with no waveform do not vote
    if len(votes) > 1:
        raise ValueError("gather_waveforms: ranks disagree on the sample type (int16 / float32 / float64)")
    np_dt, t_dt = KINDS[votes.pop() if votes else 0]
    totals = [int(h[2:2 + int(h[0])].sum()) for h in heads]
    cap = max(max(totals), 1)
    cap += cap & 1  # even sample count: the int16 payload travels as bytes / stays 4-byte aligned
    # ---- payload: one flat device buffer per rank, gathered device-to-device ----
    buf = torch.zeros(cap, dtype=t_dt, device=dev)
    o = 0
    for w in local:
        t = _as_tensor(w, dev, np_dt, t_dt)
        buf[o:o + t.numel()] = t
        o += t.numel()
    wire = buf if t_dt == torch.float32 else buf.view(torch.uint8)  # gloo has no int16 collectives; bytes for float64 too
    if dst is None:
        bufs = [torch.zeros_like(wire) for _ in range(world)]
        dist.all_gather(bufs, wire, group=group)
    else:
        bufs = [torch.zeros_like(wire) for _ in range(world)] if rank == dst else None
        dist.gather(wire, bufs, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return []
    out = []
    for r in range(world):
        flat = bufs[r].view(t_dt).cpu().numpy()
        o = 0
        for i in range(int(heads[r][0])):
            n = int(heads[r][2 + i])
            out.append(flat[o:o + n].copy())
            o += n
    return out


def _all_gather_ints(vals, dev, group=None):
    t = torch.tensor(list(vals), dtype=torch.int64, device=dev)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t, group=group)
    return [int(x) for p in parts for x in p.cpu()]
