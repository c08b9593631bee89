"""N>1 path on CPU: world_size-2 gloo run of the waveform gather (the only exchange step)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mockingbird_amd.sharding import gather_waveforms, shard_indices
    lens = [700, 1200, 64, 999, 5]
    mine = shard_indices(lens, world, rank)
    local = [np.full(lens[i], float(i + 1), np.float32) for i in mine]
    out = gather_waveforms(local, device="cpu")
    order = [i for r in range(world) for i in shard_indices(lens, world, r)]
    ok = len(out) == len(lens) and all(len(w) == lens[i] and (w == i + 1).all() for w, i in zip(out, order))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gather_waveforms_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, True), (1, True)]
