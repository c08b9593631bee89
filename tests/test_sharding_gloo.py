"""N>1 path on CPU: world_size-2 gloo run of the waveform gather (the only exchange step)."""
import os
import socket
import sys

import pytest
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
    order = [i for r in range(world) for i in shard_indices(lens, world, r)]

    def complete(out):
        return len(out) == len(lens) and all(len(w) == lens[i] and (w == i + 1).all() for w, i in zip(out, order))

    out = gather_waveforms(local, device="cpu")                 # default: gather to rank 0 only
    ok = complete(out) if rank == 0 else out == []
    out1 = gather_waveforms([torch.from_numpy(w) for w in local], device="cpu", dst=1)  # tensors in, another root
    ok = ok and (complete(out1) if rank == 1 else out1 == [])
    ok = ok and complete(gather_waveforms(local, device="cpu", dst=None))  # all ranks
    pcm = gather_waveforms([np.full(lens[i], i + 1, np.int16) for i in mine], device="cpu", dst=None)
    ok = ok and complete(pcm) and all(w.dtype == np.int16 for w in pcm)
    f64 = gather_waveforms([np.full(lens[i], i + 1 + 2.0 ** -40, np.float64) for i in mine], device="cpu")  # WaveRNN's dtype survives
    if rank == 0:
        ok = ok and len(f64) == len(lens) and all(w.dtype == np.float64 and len(w) == lens[i] and (w == i + 1 + 2.0 ** -40).all()
                                                  for w, i in zip(f64, order))
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


def _idle_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mockingbird_amd.sharding import gather_waveforms, shard_indices
    lens = [300, 20, 77]  # fewer utterances than ranks: at least one rank has nothing to do and still has to take part in the gather
    mine = shard_indices(lens, world, rank)
    order = [i for r in range(world) for i in shard_indices(lens, world, r)]
    ok = sorted(order) == [0, 1, 2]
    local = [np.full(lens[i], float(i + 1), np.float32) for i in mine]
    out = gather_waveforms(local, device="cpu")
    if rank == 0:
        ok = ok and len(out) == 3 and all(len(w) == lens[i] and (w == i + 1).all() for w, i in zip(out, order))
    else:
        ok = ok and out == []
    allr = gather_waveforms([np.full(lens[i], i + 1, np.int16) for i in mine], device="cpu", dst=None)
    ok = ok and len(allr) == 3 and all(w.dtype == np.int16 and len(w) == lens[i] and (w == i + 1).all() for w, i in zip(allr, order))
    q.put((rank, ok, len(mine)))
    dist.destroy_process_group()


def test_gather_waveforms_world4_with_idle_ranks():
    """8 GPUs and 5 requests is an ordinary serving state: ranks without an utterance still enter every collective."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_idle_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted((r, ok) for r, ok, _ in res) == [(0, True), (1, True), (2, True), (3, True)]
    assert sorted(n for _, _, n in res) == [0, 1, 1, 1]


class _StubSynth:
    """Deterministic stand-in for the Synthesizer facade: sentence -> (80, 2*len(text)) spectrogram."""
    sample_rate = 16000

    class hparams:
        hop_size = 200

    def synthesize_spectrograms(self, texts, embeddings, style_idx=0, min_stop_token=5, steps=2000):
        return [np.full((80, 2 * len(t)), float(len(t)) + float(e[0]), np.float32) for t, e in zip(texts, embeddings)]


class _StubVocoder:
    def infer_waveform_batch(self, mels, normalize=None, pcm16=None, breaks=None, break_hop=None, device_out=False, break_sample_rate=None):
        from mockingbird_amd import pipeline
        wavs = [np.sin(np.arange(m.shape[1] * 200) * 0.01 * m[0, 0]).astype(np.float32) * 0.5 for m in mels]
        if breaks is not None:  # CPU stand-in for vocoder/wave.py insert_breaks
            wavs = [pipeline.insert_breaks(w, b, break_hop, break_sample_rate or 16000).astype(np.float32) for w, b in zip(wavs, breaks)]
        if normalize is not None:  # CPU stand-in for vocoder/wave.py (the oracle is the checker here)
            from oracle import wave as owv
            wavs = [owv.peak_normalize(w, normalize) for w in wavs]
        if pcm16 is not None:
            from oracle import wave as owv
            wavs = [owv.save_wav_pcm(w) for w in wavs]
        return wavs, 16000


def _requests():
    rng = np.random.default_rng(0)
    reqs = []
    for i in range(5):
        texts = ["x" * int(rng.integers(3, 30)) for _ in range(1 + i % 3)]
        reqs.append((texts, np.full(256, float(i), np.float32)))
    return reqs


def _pipeline_worker(rank, world, port, q, kw):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mockingbird_amd import pipeline
    out = pipeline.gen_wavs(_StubSynth(), _StubVocoder(), _requests(), **kw)
    q.put((rank, [(str(w.dtype), w.tolist()) for w in out]))
    dist.destroy_process_group()


@pytest.mark.parametrize("kw", [{}, {"normalize": 0.97, "pcm16": "save_wav"}], ids=["float32", "pcm16"])
def test_pipeline_gen_wavs_world2_matches_single_process(kw):
    """configs[3] plumbing: requests sharded over 2 ranks come back complete and in request order, equal to
    the single-process result, with gen_voice.py's 0.15 s breaks after every sentence -- as float32, and as
    int16 PCM on the wire (half the gather bytes)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mockingbird_amd import pipeline
    ref = pipeline.gen_wavs(_StubSynth(), _StubVocoder(), _requests(), **kw)
    want_dtype = np.int16 if kw else np.float32
    reqs = _requests()
    for w, (texts, emb) in zip(ref, reqs):
        assert len(w) == sum(2 * len(t) * 200 for t in texts) + len(texts) * int(0.15 * 16000)
        assert (w[-int(0.15 * 16000):] == 0).all()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q, kw)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    assert res[1] == []  # finished waveforms are gathered to rank 0 only
    assert len(res[0]) == len(ref)
    for (dt, a), b in zip(res[0], ref):
        assert dt == np.dtype(want_dtype).name and b.dtype == want_dtype
        assert np.array_equal(np.asarray(a, want_dtype), b)
    if kw:
        assert max(int(np.abs(b).max()) for b in ref) in (32766, 32767)  # save_wav rescales the peak to 32767 and truncates (synthesizer/audio.py:13-15)


def _pipeline_worker_few(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mockingbird_amd import pipeline
    out = pipeline.gen_wavs(_StubSynth(), _StubVocoder(), _requests()[:1], pcm16="save_wav", normalize=0.97)  # ONE request, three sentences at most
    q.put((rank, [w.tolist() for w in out]))
    dist.destroy_process_group()


def test_pipeline_gen_wavs_more_ranks_than_sentences():
    """One request on four ranks: ranks without a sentence synthesise nothing, take part in the gather, and rank 0 still gets the
    request's waveform equal to the single-process result."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mockingbird_amd import pipeline
    ref = pipeline.gen_wavs(_StubSynth(), _StubVocoder(), _requests()[:1], pcm16="save_wav", normalize=0.97)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker_few, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(res[r] == [] for r in (1, 2, 3))
    assert len(res[0]) == 1 and np.array_equal(np.asarray(res[0][0], np.int16), ref[0])
