#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native MockingBird hot path.

Metric (BASELINE.json): audio samples/s (16 kHz) and xRT.  Workload at N=1 = BASELINE
configs[1]: WaveRNN 9-bit mu-law, batch 1, (80, 1000) mel, batched generation (target 8000 /
overlap 800 -> 23 folds x 9600 steps).  One "step" = one full infer_waveform-equivalent pass
(conditioning networks + sample loop + device->host + float64 post-processing) over one
utterance per GPU, mel already resident in HBM.  N>1: every rank vocodes its own utterance
(weak scaling, no data-path collective) and the finished waveforms are gathered device-to-device to
rank 0 with RCCL, inside the timed region.  `python bench.py --gpus N` without a launcher starts its
own N ranks (torch.distributed.run); the line carries ranks_seen = dist.get_world_size().

Also reports (same JSON line): roofline of the dominant loop kernel (HBM bound on the fp32
weights), the CPU baseline (oracle = the reference's ATen-CPU arithmetic) on a bounded sample,
and a secondary HiFi-GAN line (batch 32 x (80,200), MFMA bound).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: fp16/bf16 MFMA dense peak
HIFIGAN_MFLOP_PER_FRAME = 352.1
FREGAN_MFLOP_PER_FRAME = 385.1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=1000, help="mel frames per utterance (BASELINE configs[1]: 1000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hifigan", action="store_true")
    ap.add_argument("--no-tacotron", action="store_true")
    ap.add_argument("--no-ppg2mel", action="store_true")
    ap.add_argument("--no-wavernn-batch", action="store_true")
    ap.add_argument("--no-wavernn-unbatched", action="store_true")
    ap.add_argument("--no-wavernn-mol", action="store_true",
                    help="skip the MOL object (the rocprofv3 pass behind frac_rocprof: its launches share the headline's kernel name)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the configs[3] / configs[4] sharded objects")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` with no launcher around it starts its own N ranks (one process per GPU) by
    re-executing itself under torch.distributed.run -- the same command line the driver uses.  Under a launcher
    (WORLD_SIZE set) this is a no-op; a WORLD_SIZE that disagrees with --gpus is an error, not a silent fallback."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"[bench] launched with WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a "
                             f"line whose n_gpus is not the number of ranks that ran")
        return
    if args.gpus <= 1:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


class _StubModel:
    """MBHIP_BENCH_STUB=1 (tests/test_bench_spawn.py, CPU + gloo): stands in for the WaveRNN handle so that the
    launcher / barrier / max-over-ranks / gather / JSON plumbing can be exercised without a GPU.  The line it
    produces says data = "stub"; it is never a measurement."""
    hop_length, last_loop_ms = 256, 1.0

    class _P:
        n_folds, seq_len = 23, 9600
    last_plan = _P()

    def generate_samples(self, mel, batched, target, overlap, seed=0):
        time.sleep(0.01)
        return torch.zeros(23, 16)

    def finish(self, samples, batched, overlap, mu_law, wave_len):
        return np.zeros(4096, np.float64)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


KERNEL_STATS_CSV = "r06_bench_kernel_stats.csv"   # rocprofv3 --kernel-trace --stats summary of this command, committed per round
CPU_BASELINE_THREADS = 8   # one policy for every cpu_baseline leg: min(host threads, 8) -- the reference's eager ops on small batches get
                           # SLOWER beyond that (WaveRNN's 23-row GEMVs 3x at 128 threads, HiFi-GAN 1 x (80,200) 4.5x); `cores` states it


class _cpu_threads:
    """with _cpu_threads() as n: torch runs on n = min(default, CPU_BASELINE_THREADS) threads inside, the default again outside"""
    def __enter__(self):
        self.default = torch.get_num_threads()
        self.n = min(self.default, CPU_BASELINE_THREADS)
        torch.set_num_threads(self.n)
        return self.n

    def __exit__(self, *exc):
        torch.set_num_threads(self.default)
        return False


def _reference_available():
    try:
        import refimport
        return refimport.available()
    except Exception:
        return False


def cpu_baseline_wavernn(state, F, target, overlap, n_useful, plan, cpu_seconds):
    """The reference's WaveRNN.generate (fatchord_version.py:153-257) on the host cores, same weights and mel.
    kind "reference": the real nn.Module from /root/reference, ONE full generate (it cannot be bounded: 23 folds x
    9600 steps).  kind "port": oracle/wavernn.py (the same ATen-CPU ops in the same order) for ~cpu_seconds of the
    same workload.  Threads: min(torch's default, 8) -- the loop is 512-wide GEMVs on 23 rows, which 128 threads make three times
    SLOWER than 8 (1 909 samples/s on 128 threads against 5 817 on 8); `cores` says what was used."""
    ct = _cpu_threads()
    with ct as threads:
        return _cpu_baseline_wavernn(state, F, target, overlap, n_useful, plan, cpu_seconds, threads, ct.default)


def _cpu_baseline_wavernn(state, F, target, overlap, n_useful, plan, cpu_seconds, threads, default_threads):
    import synth
    mel = synth.wavernn_mel(F, seed=1)
    if _reference_available():
        try:
            import refimport
            refimport.setup()
            from models.vocoder.wavernn.models.fatchord_version import WaveRNN
            from oracle import wavernn as ow
            hp = ow.HP
            m = WaveRNN(rnn_dims=hp["rnn_dims"], fc_dims=hp["fc_dims"], bits=hp["bits"], pad=hp["pad"],
                        upsample_factors=hp["upsample_factors"], feat_dims=hp["feat_dims"], compute_dims=hp["compute_dims"],
                        res_out_dims=hp["res_out_dims"], res_blocks=hp["res_blocks"], hop_length=hp["hop_length"],
                        sample_rate=hp["sample_rate"], mode=hp["mode"])
            m.load_state_dict(state, strict=False)
            m.eval()
            import contextlib
            import io
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):  # the reference draws a progress bar on stdout
                out = m.generate(torch.from_numpy(mel[None] / 4.0), True, target, overlap, hp["mu_law"], None)
            tc = time.perf_counter() - t0
            return {"value": len(out) / tc, "unit": "samples/s", "cores": threads, "kind": "reference",
                    "sample": f"the reference's WaveRNN.generate (imported from the checkout) on the same mel 80x{F}, one "
                              f"full pass: {len(out)} samples in {tc:.1f} s"}
        except Exception as e:  # fall through to the port, and say why
            note = f" (reference import failed: {type(e).__name__}: {e})"
    else:
        note = " (no reference checkout on this host)"
    from oracle import wavernn as ow
    w = dict(state)
    with torch.no_grad():
        mels, aux = ow.conditioning(w, ow.HP, torch.from_numpy(mel[None, :, :] / 4.0), True, target, overlap)
        nf = mels.shape[0]
        ow.sample_loop(w, ow.HP, mels[:, :2], aux[:, :2])  # warm-up
        t0c = time.perf_counter()
        steps_done, chunk = 0, 10
        while time.perf_counter() - t0c < cpu_seconds and steps_done + chunk <= mels.shape[1]:
            ow.sample_loop(w, ow.HP, mels[:, steps_done:steps_done + chunk], aux[:, steps_done:steps_done + chunk])
            steps_done += chunk
        tc = time.perf_counter() - t0c
    raw_rate = nf * steps_done / tc
    useful = n_useful / float(plan.n_folds * plan.seq_len)
    return {"value": raw_rate * useful, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"oracle sample loop (the reference's ATen-CPU ops, oracle/wavernn.py){note}: {nf} folds x {steps_done} "
                      f"steps of the same workload in {tc:.1f} s on {threads} threads (host default {default_threads}); raw "
                      f"{raw_rate:.0f} fold-steps/s scaled by the useful-sample fraction {useful:.3f}"}


def cpu_baseline_hifigan():
    """BASELINE configs[0]: HiFi-GAN infer_waveform on 1 x (80, 200), CPU, median of 3."""
    import synth
    from oracle import gan as og
    h = synth.HIFIGAN_16K
    st = synth.gan_state(h, "hifigan", seed=3)["generator"]
    mel = torch.from_numpy(synth.mel_input(200, 1, seed=0))
    kind, fwd = "port", None
    if _reference_available():
        try:
            import refimport
            refimport.setup()
            from models.vocoder.hifigan.models import Generator
            from utils.util import AttrDict
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                g = Generator(AttrDict(h))
                g.load_state_dict(st)
                g.eval()
                g.remove_weight_norm()
            kind, fwd = "reference", (lambda: g(mel))
        except Exception:
            fwd = None
    if fwd is None:
        w = og.fold_weight_norm_state(st)
        fwd = lambda: og.hifigan_forward(w, h, mel)  # noqa: E731
    ts = []
    with torch.no_grad(), _cpu_threads() as threads:
        fwd()
        for _ in range(3):
            t0 = time.perf_counter()
            y = fwd()
            ts.append(time.perf_counter() - t0)
    t = _median(ts)
    return {"value": y.numel() / t, "unit": "samples/s", "cores": threads, "kind": kind,
            "sample": f"BASELINE configs[0]: generator forward on 1 x (80,200), median of 3 = {t * 1e3:.1f} ms on {threads} threads "
                      f"({'the reference Generator' if kind == 'reference' else 'oracle/gan.py, the same ATen-CPU ops'})"}


def _ref_tacotron(Tacotron, rhp, symbols):
    # models/synthesizer/inference.py:54-67
    return Tacotron(embed_dims=rhp.tts_embed_dims, num_chars=len(symbols), encoder_dims=rhp.tts_encoder_dims,
                    decoder_dims=rhp.tts_decoder_dims, n_mels=rhp.num_mels, fft_bins=rhp.num_mels,
                    postnet_dims=rhp.tts_postnet_dims, encoder_K=rhp.tts_encoder_K, lstm_dims=rhp.tts_lstm_dims,
                    postnet_K=rhp.tts_postnet_K, num_highways=rhp.tts_num_highways, dropout=rhp.tts_dropout,
                    stop_threshold=rhp.tts_stop_threshold, speaker_embedding_size=rhp.speaker_embedding_size)


def cpu_baseline_tacotron():
    """BASELINE configs[2]: Tacotron.generate, B = 32, ~100 tokens, 400 frames forced, CPU, median of 3."""
    import synth
    from oracle import tacotron as ot
    tst = synth.tacotron_state(seed=3)["model_state"]
    seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
    Tt = max(len(q) for q in seqs)
    chars = torch.tensor(np.stack([np.pad(q, (0, Tt - len(q))) for q in seqs])).long()
    spk = torch.tensor(np.stack(emb))
    kind, fwd = "port", None
    if _reference_available():
        try:
            import refimport
            refimport.setup()
            from models.synthesizer.models.tacotron import Tacotron
            from models.synthesizer.hparams import hparams as rhp
            from models.synthesizer.utils.symbols import symbols
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                m = _ref_tacotron(Tacotron, rhp, symbols)
            m.load_state_dict(tst, strict=False)
            m.eval()
            kind, fwd = "reference", (lambda: m.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11))
        except Exception:
            fwd = None
    if fwd is None:
        fwd = lambda: ot.generate(tst, ot.HP, 2, chars, spk, steps=400, style_idx=-1, min_stop_token=11)  # noqa: E731
    ts = []
    with torch.no_grad(), _cpu_threads() as threads:
        for _ in range(3):
            t0 = time.perf_counter()
            out = fwd()
            ts.append(time.perf_counter() - t0)
    t = _median(ts)
    frames = int(out[1].shape[-1]) * 32
    return {"value": frames / t, "unit": "mel frames/s", "cores": threads, "kind": kind,
            "sample": f"BASELINE configs[2]: generate, B=32, T={Tt}, {frames // 32} frames forced, median of 3 = {t:.2f} s on {threads} threads "
                      f"({'the reference Tacotron' if kind == 'reference' else 'oracle/tacotron.py, the same ATen-CPU ops'})"}


def sharded_objects(args, rank, world, dev, use_dist, timed, total_over_ranks):
    """The two BASELINE configs that are DEFINED on 8 GPUs, as weak-scaling objects (every rank runs its per-GPU share,
    finished waveforms gathered device-to-device to rank 0 inside the timed region):
      configs[3]: end-to-end gen_voice.py shape, Tacotron2 + HiFi-GAN, 256 requests / 8 GPUs = 32 per rank
      configs[4]: Fre-GAN fp16, batch 64 x (80,3000) / 8 GPUs = 8 per rank."""
    import tempfile
    import synth
    from mockingbird_amd import pipeline
    from mockingbird_amd.synthesizer.inference import Synthesizer
    from mockingbird_amd.vocoder import gan as gan_mod
    out = {}
    tmp = tempfile.mkdtemp(prefix=f"mbhip_bench_r{rank}_")
    # ---- configs[3] ----
    torch.save(synth.tacotron_state(seed=3), os.path.join(tmp, "taco.pt"))
    syn = Synthesizer(os.path.join(tmp, "taco.pt"), verbose=False)
    voc = gan_mod.GanFacade(gan_mod.KIND_HIFIGAN, "", "hifigan")
    os.makedirs(os.path.join(tmp, "voc"))
    torch.save(synth.gan_state(synth.HIFIGAN_16K, "hifigan", seed=3), os.path.join(tmp, "voc", "g.pt"))
    json.dump(synth.HIFIGAN_16K, open(os.path.join(tmp, "voc", "config.json"), "w"))
    voc.load_model(os.path.join(tmp, "voc", "g.pt"), verbose=False)
    rng = np.random.default_rng(7)  # every rank draws the same request list: ~100-character ASCII prompts
    n_req = 32 * world
    alphabet = np.array(list("abcdefghijklmnopqrstuvwxyz     "))
    requests = []
    for i in range(n_req):
        text = "".join(rng.choice(alphabet, int(rng.integers(90, 111)))).strip() or "a"
        e = rng.standard_normal(256).astype(np.float32)
        requests.append(([text], e / np.linalg.norm(e)))
    import contextlib
    import io

    e2e_t = {}

    def e2e(_):
        with contextlib.redirect_stdout(io.StringIO()):  # the facade prints the prompts, like the reference
            wavs = pipeline.gen_wavs(syn, voc, requests, steps=400, min_stop_token=11, normalize=0.97, pcm16="save_wav", chunk_size=32,
                                     timings=e2e_t)
        return sum(len(w) for w in wavs)  # rank 0 holds everything after the gather; other ranks 0

    el, res = timed(e2e, 1, 1)
    n_samples = total_over_ranks(res[0])
    out["e2e_configs3"] = {
        "workload": f"BASELINE configs[3]: {n_req} cloned-voice requests (~100 tokens, one sentence) sharded over {world} GPU(s) "
                    "= 32 per rank: Synthesizer.synthesize_spectrograms (ONE decoder loop per rank: additive chunk_size=32 instead of the default 2 x 16; 400 frames forced) -> HiFi-GAN V1 fp32 -> "
                    "0.15 s breaks, peak normalise, int16 PCM on the device -> device-to-device gather to rank 0",
        "value": n_samples / el, "unit": "samples/s", "x_realtime": n_samples / el / 16000.0, "s_total": el,
        "requests": n_req, "n_gpus": world, "scaling": "weak",
        "shares_rank0": {"synthesizer_s": e2e_t.get("synthesizer"), "vocoder_s": e2e_t.get("vocoder"), "gather_s": e2e_t.get("gather"),
                         "vocoder_share": (e2e_t["vocoder"] / el) if e2e_t.get("vocoder") is not None else None,
                         "what": "this rank's seconds of the timed pass: Synthesizer facade (Tacotron encoder + decoder loop + postnet + "
                                 "D2H of the mels + trimming), vocoder facade (H2D of the mels, HiFi-GAN fp32-result forward, breaks, peak "
                                 "normalisation, PCM16; device-synchronised), gather"}}
    del syn, voc
    # ---- configs[4] ----
    hf = synth.FREGAN_16K
    gen = gan_mod.GanFacade(gan_mod.KIND_FREGAN, "", "fregan")
    os.makedirs(os.path.join(tmp, "fre"))
    torch.save(synth.gan_state(hf, "fregan", seed=4), os.path.join(tmp, "fre", "g.pt"))
    json.dump(hf, open(os.path.join(tmp, "fre", "config.json"), "w"))
    gen.load_model(os.path.join(tmp, "fre", "g.pt"), verbose=False, dtype="f16")
    mels = [m for m in synth.mel_input(3000, 8, seed=1 + rank)]
    from mockingbird_amd import sharding

    def fre(_):
        wavs, _sr = gen.infer_waveform_batch(mels, pcm16="encode_16bits", device_out=True)
        got = sharding.gather_waveforms(wavs, dev, dst=0)
        return sum(len(w) for w in got)

    el, res = timed(fre, 2, 1)
    n_samples = total_over_ranks(sum(res))
    fl = FREGAN_MFLOP_PER_FRAME * 1e6 * 3000 * 8 * world * 2
    out["fregan_configs4"] = {
        "workload": f"BASELINE configs[4]: Fre-GAN fp16 MFMA, batch {8 * world} x mel (80,3000) = 8 per rank over {world} GPU(s): "
                    "H2D of the mels, generator forward, int16 PCM on the device, gather to rank 0 (2 timed passes)",
        "value": n_samples / el, "unit": "samples/s", "x_realtime": n_samples / el / 16000.0, "s_total": el,
        "n_gpus": world, "scaling": "weak", "aggregate_TFLOPs": fl / el / 1e12}
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    torch.cuda.empty_cache()
    return out


def gan_floor_bytes(h, batch, frames, fregan=False, top_k=4):
    """Read-x + write-y floor of one fp16 generator forward AS LAUNCHED (every launch reads its input and writes its output
    once, 2 B per element): conv_pre, per stage the upsampler and either num_kernels x len(dilations) fused pair units (the mean
    over the parallel ResBlocks adds a read of the accumulator for all but the first) or, for the narrow stages (<= 32 channels),
    ONE launch for the whole ResBlock group (resblock_stage_f16.hip: one read, one write); at 64 / 128 channels the k = 3 ResBlocks
    are one launch each; conv_post.  fregan=True adds that generator's own launches: the cond_up chain, `x += mel`, res_output and
    `output + x` (fused into the res_output launch except at the first level)."""
    B, T, C = batch, frames, h["upsample_initial_channel"]
    b = B * T * (h["num_mels"] + C) * 2.0
    nk, nd = len(h["resblock_kernel_sizes"]), len(h["resblock_dilation_sizes"][0])
    lvl = len(h["upsample_rates"]) - top_k if fregan else 1 << 30
    mel_c = h["num_mels"]
    for i, u in enumerate(h["upsample_rates"]):
        if i >= lvl:      # Fre-GAN: mel = cond_up(mel) (read + write), x += mel (two reads, one write)  generator.py:142-144
            b += B * (T // h["upsample_rates"][i - 1]) * mel_c * 2.0 + B * T * C * 2.0 + 3 * B * T * C * 2.0
            mel_c = C
        if i > lvl:       # output = res_output(x or output) + x (generator.py:145-159): the source at 1/u of the rows, the result
            dst = B * T * u * (C // 2) * 2.0                                     # written once, x read once as the residual; at the
            b += B * T * C * 2.0 + (4 if i == lvl + 1 else 2) * dst              # first level "+ x" is a launch of its own
        b += B * T * C * 2.0
        T, C = T * u, C // 2
        b += B * T * C * 2.0
        if C <= 32:
            b += 2 * B * T * C * 2.0
        else:
            for j, ks in enumerate(h["resblock_kernel_sizes"]):
                # 64 / 128 channels: a ResBlock with k = 3 is one launch (gan.hip: mb_resblock_stage_f16_efficiency >= 0.65 for k = 3, which
                # holds for both benchmarked dilation sets), the others one per unit
                chain = C in (64, 128) and ks == 3
                b += (2 if chain else nd * 2) * B * T * C * 2.0 + (B * T * C * 2.0 if j > 0 else 0.0)
    return b + B * T * C * 2.0 + B * T * 4.0


def pmc_traffic(what, keys=None):
    """HBM bytes from the committed rocprofv3 PMC passes of the benchmarked configuration (profiles/r0N_pmc_<what>.json,
    tools/pmc_r02.sh: 2 x FETCH_SIZE + WRITE_SIZE per launch; counters cannot be read in-process).  keys = kernels to
    sum per launch; None = all bytes of the profiled command.  Returns (bytes, source) or (None, None)."""
    pm = name = None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):  # the newest committed pass of this object
        try:
            name = f"profiles/{rnd}_pmc_{what}.json"
            pm = json.load(open(os.path.join(ROOT, name)))
            break
        except Exception:
            pm = None
    if pm is None:
        return None, None
    if keys is None:
        tot = sum(v for k, v in pm.items() if k.endswith("_hbm_bytes_total")) / max(pm.get("forwards", 1), 1)
    else:
        vals = [pm.get(k + "_hbm_bytes_per_launch") for k in keys]
        if any(v is None for v in vals):
            return None, None
        tot = sum(vals)
    return tot, f"{name}: {pm.get('source', '')[:200]}"


def main():
    args = parse()
    spawn_ranks_if_needed(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = os.environ.get("MBHIP_BENCH_STUB") == "1"
    import torch.distributed as dist
    use_dist = world > 1
    if use_dist:
        # N > 1: the line is the sharded headline workload + its roofline and the sharded configs[3]/[4] objects; the
        # single-GPU secondary objects and the CPU baseline are N = 1 material (rank 0 would keep the others waiting)
        args.no_hifigan = args.no_tacotron = args.no_ppg2mel = args.no_cpu_baseline = args.no_wavernn_batch = True
        args.no_wavernn_unbatched = True
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ranks_seen = dist.get_world_size() if use_dist else 1

    def sync():
        if not stub:
            torch.cuda.synchronize()

    import synth
    from mockingbird_amd import sharding
    if stub:
        model, state, _lib = _StubModel(), None, None
    else:
        from mockingbird_amd import build, _lib
        if rank == 0:
            build.build(verbose=False)
        if use_dist:
            dist.barrier()
        from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
        state = synth.wavernn_state(seed=5)["model_state"]
        model = WaveRNNDevice(state)
    F = args.frames
    target, overlap = 8000, 800
    # each rank owns one utterance (seed differs per rank), already resident in HBM
    mel = torch.from_numpy(synth.wavernn_mel(F, seed=1 + rank) / 4.0).to(dev)
    wave_len = (F - 1) * model.hop_length

    phase_s = []  # (compute, gather) seconds of every pass of this rank: a 1 -> 8 run must show which of the two grows

    def one_pass(seed, mel_in=None):
        tc0 = time.perf_counter()
        samples = model.generate_samples(mel if mel_in is None else mel_in, True, target, overlap, seed=seed)
        if use_dist and not stub:
            # float64 tail on the device; the finished waveform goes device-to-device to rank 0 (the only exchange
            # of the whole path), which copies the gathered set to the host once
            wav = model.finish(samples, True, overlap, True, wave_len, device_out=True)
            sync()
            tc1 = time.perf_counter()
            sharding.gather_waveforms([wav.to(torch.float32)], dev, dst=0)
        else:
            wav = model.finish(samples, True, overlap, True, wave_len)  # float64 tail on the device, D2H of the waveform
            tc1 = time.perf_counter()
            if use_dist:
                sharding.gather_waveforms([wav.astype(np.float32)], dev, dst=0)
        phase_s.append((tc1 - tc0, time.perf_counter() - tc1))
        return wav, len(wav)

    def timed(fn, steps, warmup):
        """warmup untimed calls, then exactly `steps` calls bracketed by barrier + synchronize on both sides;
        returns (max-over-ranks seconds, per-rank results of the timed calls)."""
        for i in range(warmup):
            fn(1000 + i)
        sync()
        if use_dist:
            dist.barrier()
        sync()
        t0_ = time.perf_counter()
        res = [fn(i) for i in range(steps)]
        sync()
        if use_dist:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0_
        if use_dist:
            te = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        return el, res

    def total_over_ranks(n):
        if not use_dist:
            return n
        t = torch.tensor([n], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    loop_ms = []

    def headline(i):
        wav_, n_ = one_pass(i)
        loop_ms.append(model.last_loop_ms)
        return wav_, n_

    elapsed, res = timed(headline, args.steps, args.warmup)
    loop_ms = loop_ms[args.warmup:]
    side = {}
    if not stub and not use_dist:
        # ---- the same pass with the mel handed over as a HOST buffer (H2D of the mel inside the timed region, as the
        #      facade's callers do).  Reported beside `value`, never as `value` (the task's contract: inputs resident in HBM).
        mel_host = torch.from_numpy((synth.wavernn_mel(F, seed=1 + rank) / 4.0).astype(np.float32)).pin_memory()
        el_h, res_h = timed(lambda i: one_pass(i, mel_host.to(dev, non_blocking=True)), args.steps, 1)
        n_h = sum(n for _, n in res_h)
        side["pcie_inclusive"] = {"value": n_h / el_h, "unit": "samples/s", "ms_per_step": el_h / args.steps * 1e3,
                                  "what": f"the headline pass with the {mel_host.numel() * 4} B mel uploaded from pinned host memory inside "
                                          "the timed region (the D2H of the waveform is inside both)"}
        # ---- the strict-fp32 kernel in the same run: MBHIP_WAVERNN_RESIDENT=exact = wf_pipe_kernel, fp32-input MFMA,
        #      bit-identical to the launch chain
        prev = os.environ.get("MBHIP_WAVERNN_RESIDENT")
        os.environ["MBHIP_WAVERNN_RESIDENT"] = "exact"
        try:
            x_ms = []

            def exact_pass(i):
                r_ = one_pass(i)
                x_ms.append(model.last_loop_ms)
                return r_
            el_x, res_x = timed(exact_pass, args.steps, 1)
            n_x = sum(n for _, n in res_x)
            side["exact_f32"] = {"value": n_x / el_x, "unit": "samples/s", "x_realtime": n_x / el_x / 16000.0, "ms_per_step": el_x / args.steps * 1e3,
                                 "dtype": "f32 (fp32-input MFMA v_mfma_f32_16x16x4_f32, fp32 everywhere)",
                                 "sample_loop_ms": float(np.median(x_ms[1:])), "us_per_time_step": float(np.median(x_ms[1:])) * 1e3 / model.last_plan.seq_len,
                                 "loop_launches": model.last_loop_launches, "path": model.last_path,
                                 "kernel": "mb::wf_pipe_kernel (wavernn_pipe.h): the same roles and hand-offs on 8-byte {fp32 value, tag} granules; "
                                           "sample stream bit-identical to the 5-launch chain"}
        finally:
            if prev is None:
                os.environ.pop("MBHIP_WAVERNN_RESIDENT", None)
            else:
                os.environ["MBHIP_WAVERNN_RESIDENT"] = prev
        model.generate_samples(mel, True, target, overlap, seed=0)  # (last_plan / last_loop_launches of the default path again)
        torch.cuda.synchronize()
    my_phase = [float(np.median([p[k] for p in phase_s[args.warmup:]])) * 1e3 for k in (0, 1)]
    my_phase.append(float(local_rank if stub else torch.cuda.current_device()))  # the device this rank ran on (LOCAL_RANK -> cuda:LOCAL_RANK)
    if use_dist:  # per-rank medians, in rank order
        tp = torch.tensor(my_phase, device=dev, dtype=torch.float64)
        parts = [torch.zeros_like(tp) for _ in range(world)]
        dist.all_gather(parts, tp)
        per_rank = [[float(x) for x in p.cpu()] for p in parts]
    else:
        per_rank = [my_phase]
    wav = res[-1][0]
    total_samples = total_over_ranks(sum(n for _, n in res))  # every rank vocoded its own utterance
    plan = model.last_plan
    value = total_samples / elapsed
    result = {
        "metric": "audio samples/sec (16 kHz), WaveRNN vocoder.infer_waveform",
        "value": value, "unit": "samples/s", "x_realtime": value / 16000.0,
        "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1000.0, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "split-f16 products (two fp16 halves per operand, 21-22 bits), f32 accumulate / state / epilogues",
        "data": "stub" if stub else "synthetic",
        "dtype_note": ("what MULTIPLIES on the headline path is the fp16 matrix pipe: w 2^s = wh + wl, x = xh + 2^-11 xl (scaled residual since "
                       "round 5), products wl.xh + wh.xh + 2^-11 wh.xl in fp32 accumulators (wavernn_pipe16.h); weights, GRU state, sums, gate "
                       "functions, sampler and samples are fp32.  Parity = the reference loop body replayed on the device's own sample "
                       "history with the exported noise over ALL 9600 steps x 23 folds (tests/test_wavernn_gpu.py::"
                       "test_production_default_full_length_vs_oracle: every pick equal except provable near-ties).  The strict-fp32 "
                       "kernel (MBHIP_WAVERNN_RESIDENT=exact: fp32-input MFMA, bit-identical to the launch chain) is timed in the same run: "
                       "`exact_f32` below"),
        "config": {"workload": "BASELINE configs[1]: WaveRNN 9-bit mu-law RAW, batch=1 utterance/GPU, "
                               f"mel 80x{F}, batched target=8000 overlap=800 -> {plan.n_folds} folds x "
                               f"{plan.seq_len} steps, Philox sampling, fp32 weights (synthetic, seeded)",
                   "samples_per_utterance": int(len(wav)), "utterances": world,
                   "sample_loop_ms": float(np.median(loop_ms)),
                   "us_per_time_step": float(np.median(loop_ms)) * 1000.0 / plan.seq_len},
        "per_rank_ms": {"compute": [p[0] for p in per_rank], "gather": [p[1] for p in per_rank], "device": [int(p[2]) for p in per_rank],
                        "what": "median over the timed passes of each rank: compute = conditioning + sample loop + float64 tail "
                                "(+ D2H at 1 GPU); gather = the device-to-device gather of the finished waveforms to rank 0 "
                                "(0 at 1 GPU)"},
    }

    result.update(side)
    if not stub:
        result["config"]["loop_path"] = {"path": model.last_path, "fallback": model.last_fallback}
    if "exact_f32" in side:  # the strict-fp32 leg where the driver's `parsed` keeps it whole
        xf = side["exact_f32"]
        result["config"].update({"exact_f32_samples_per_s": xf["value"], "exact_f32_us_per_step": xf["us_per_time_step"],
                                 "exact_f32_x_realtime": xf["x_realtime"]})
    if stub:
        if rank == 0:
            print(json.dumps(_ordered(result)))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not args.no_e2e:
        result.update(sharded_objects(args, rank, world, dev, use_dist, timed, total_over_ranks))
    if rank == 0:
        # ---- roofline of the dominant loop kernel
        L = _lib.lib()
        ws = model._ws
        smp = torch.empty(plan.n_folds, plan.seq_len, device=dev)
        resident = model.last_loop_launches == 1  # wavernn_pipe.h: the whole sample loop is ONE resident launch
        step_bytes = 16.3e6 + 452.0 * plan.n_folds  # fp32 weights + conditioning per step per fold-batch
        per_kernel = {}
        split = "classic" not in os.environ.get("MBHIP_WAVERNN_CHAIN", "").split(",")
        names = (("gru1_finish", "rnn2_input_half", "fc1+hh1", "fc2+hh2", "fc3_sampler") if split
                 else ("rnn1_gru", "rnn2_gru", "fc1", "fc2", "fc3_sampler"))
        for which, name in enumerate(names):  # (mb_wavernn_bench_kernel always times the launch chain)
            us, ab = C.c_float(), C.c_double()
            _lib.check(L.mb_wavernn_bench_kernel(model._h, C.byref(plan), _lib.ptr(mel), _lib.ptr(smp),
                                                 _lib.ptr(ws), ws.numel(), which, 0, C.byref(us), C.byref(ab),
                                                 _lib.stream_ptr()), "mb_wavernn_bench_kernel")
            torch.cuda.synchronize()
            per_kernel[name] = {"avg_us": us.value, "algorithmic_bytes": ab.value,
                                "GBps": ab.value / (us.value * 1e-6) / 1e9}
        dom_name = "fc1+hh1" if split else "rnn2_gru"
        dom = per_kernel[dom_name]

        def committed(key, files):
            """HBM traffic from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
            correction + WRITE_SIZE); counters cannot be read in-process."""
            for cand in files:
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", cand)))
                    if pm.get(key) is not None:
                        return pm[key], f"profiles/{cand}: {pm.get('source', '')[:200]}"
                except Exception:
                    pass
            return None, None

        def rocprof_avg_us(kernel_substr):
            """average duration of a kernel in the committed rocprofv3 --kernel-trace --stats summary of this command"""
            try:
                import csv
                for row in csv.DictReader(open(os.path.join(ROOT, "profiles", KERNEL_STATS_CSV))):
                    if kernel_substr in row.get("Name", ""):
                        return float(row["AverageNs"]) / 1e3
            except Exception:
                pass
            return None

        if resident:
            launch_us = float(np.median(loop_ms)) * 1000.0
            launch_bytes = step_bytes * plan.seq_len
            gbps = launch_bytes / (launch_us * 1e-6) / 1e9
            q16 = os.environ.get("MBHIP_WAVERNN_RESIDENT", "auto") != "exact"  # wavernn_pipe16.h (default) or the exact wavernn_pipe.h kernel
            traffic, traffic_src = committed("pipe_hbm_bytes_per_launch", ("r05_pmc_wavernn.json", "r04_pmc_wavernn.json") if q16 else ("r03_pmc_wavernn.json",))
            resident_env = os.environ.get("MBHIP_WAVERNN_RESIDENT")
            os.environ["MBHIP_WAVERNN_RESIDENT"] = "0"  # the launch chain on the same utterance, for reference
            model.generate_samples(mel, True, target, overlap, seed=0)
            chain_us = model.last_loop_ms * 1e3 / plan.seq_len
            if resident_env is None:
                os.environ.pop("MBHIP_WAVERNN_RESIDENT")
            else:
                os.environ["MBHIP_WAVERNN_RESIDENT"] = resident_env
            rp = rocprof_avg_us("wf_pipe16_kernel" if q16 else "wf_pipe_kernel")
            result["roofline"] = {
                "kernel": ("mb::wf_pipe16_kernel (wavernn_pipe16.h): the whole sample loop of the utterance as ONE resident launch -- 224 "
                           "role-specialised workgroups, two fold-column groups in flight; round 4: exchange vectors as fp16 hi / lo pairs "
                           "(two features per 8-byte granule, 2-bit tags: half the sweep bytes), products as error-compensated "
                           "v_mfma_f32_16x16x32_f16 on weight fragments held in registers; samples held to the oracle, "
                           "tests/test_wavernn_gpu.py::test_production_*" if q16 else
                           "mb::wf_pipe_kernel (wavernn_pipe.h, MBHIP_WAVERNN_RESIDENT=exact): the exact fp32 resident kernel -- 224 "
                           "role-specialised workgroups, weight tiles in LDS, two fold-column groups in flight, granule hand-offs"),
                "bound": "latency",
                "bound_note": "nominal roofline = HBM on the weights AS IF streamed every step (SURVEY 8(d)): `achieved` / `frac` price the launch "
                              "that way.  The kernel keeps every weight in registers and reads it once per utterance; what bounds a step is "
                              "the latency of its five dependent all-to-all hand-offs (MI355X_MICROARCH.md handoff-1to1: 0.8-1.0 us each) plus "
                              "the stages between them -- `counter_GBps` is the HBM rate the PMC counters actually see",
                "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
                "frac_rocprof": (launch_bytes / (rp * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp else None, "rocprof_avg_launch_us": rp,
                "traffic": traffic, "traffic_source": traffic_src,
                "counter_GBps": (traffic / (launch_us * 1e-6) / 1e9) if traffic else None,
                "algorithmic_bytes_per_launch": launch_bytes,
                "algorithmic_bytes_per_step": step_bytes, "steps_per_launch": plan.seq_len,
                "avg_launch_us": launch_us,
                "avg_launch_us_method": "HIP events recorded on the loop's own stream right before and after the launch "
                                        "(mb_wavernn_last_loop_ms), median over the timed passes; frac_rocprof uses the average "
                                        "duration of the same kernel in profiles/" + KERNEL_STATS_CSV,
                "note": "algorithmic bytes follow SURVEY 8(d) (16.3 MB of fp32 weights per step as if streamed); the resident kernel "
                        "reads each weight ONCE per utterance, so `traffic` is far below them -- the bound that matters is the "
                        "hand-off latency of the 5 all-to-all edges per step, DESIGN.md section 4e",
                "whole_step": {"algorithmic_bytes": step_bytes, "us": result["config"]["us_per_time_step"],
                               "GBps": step_bytes / (result["config"]["us_per_time_step"] * 1e-6) / 1e9},
                "chain_reference": {"us_per_time_step": chain_us, "loop": "5-launch chain (wavernn_fast.h), hipGraph replays, same utterance",
                                    "dominant_launch": dom_name, "per_kernel_marginal": per_kernel},
            }
        else:
            traffic, traffic_src = committed(("fc1_hh1" if split else "rnn2_gru") + "_hbm_bytes_per_launch",
                                             ("r02_pmc_wavernn.json", "r01_pmc_wavernn.json"))
            launch_us = result["config"]["us_per_time_step"] / 5.0  # every launch of the chain, back to back
            dom_gbps = dom["algorithmic_bytes"] / (launch_us * 1e-6) / 1e9
            rp = rocprof_avg_us("wf_fc_hh_kernel")
            result["roofline"] = {
                "kernel": ("mb::wf_fc_hh_kernel<NT> (wavernn_fast.h: WaveRNN fc1 beside the hidden half of the next step's "
                           "rnn1; the launch with the most bytes of the chain)" if split else
                           "mb::rnn_rowtile_kernel<EPI_GRU, 1, 8, ...> (WaveRNN rnn2 instance, in-situ marginal duration)"),
                "chain": "split-hidden" if split else "classic",
                "bound": "hbm", "achieved": dom_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom_gbps / HBM_PEAK_GBS,
                "frac_rocprof": (dom["algorithmic_bytes"] / (rp * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp else None, "rocprof_avg_launch_us": rp,
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "avg_launch_us": launch_us,
                "avg_launch_us_method": "HIP events on the loop's stream around the whole sample loop / launches: the mean "
                                        "launch-to-launch period of the 5-launch chain; frac_rocprof uses the kernel's average in the "
                                        "committed rocprofv3 summary; per_kernel[*].avg_us are in-situ MARGINAL times (loop timed with "
                                        "and without that launch)",
                "marginal_us": dom["avg_us"], "marginal_GBps": dom["GBps"],
                "whole_step": {"algorithmic_bytes": step_bytes, "us": result["config"]["us_per_time_step"],
                               "GBps": step_bytes / (result["config"]["us_per_time_step"] * 1e-6) / 1e9},
                "per_kernel": per_kernel,
            }
        # ---- secondary: the same utterance with batched=False (SURVEY.md section 8d config 1 "also report"): ONE
        # fold, one column per launch, every sample a dependent step -- the pure latency floor of the chain
        if not args.no_wavernn_unbatched:
            model.generate_samples(mel[:, :40], False, target, overlap, seed=3)  # untimed: graph capture / first touch
            torch.cuda.synchronize()
            t0u = time.perf_counter()
            su = model.generate_samples(mel, False, target, overlap, seed=3)
            wu = model.finish(su, False, overlap, True, wave_len)
            torch.cuda.synchronize()
            tu = time.perf_counter() - t0u
            pu = model.last_plan
            result["wavernn_unbatched"] = {
                "workload": f"mel 80x{F}, batched=False: {pu.n_folds} fold x {pu.seq_len} dependent steps, Philox sampling, fp32",
                "value": len(wu) / tu, "unit": "samples/s", "x_realtime": len(wu) / tu / 16000.0, "s_total": tu,
                "sample_loop_ms": model.last_loop_ms, "us_per_time_step": model.last_loop_ms * 1e3 / pu.seq_len,
                "loop_launches": model.last_loop_launches,
                "loop": ("ONE persistent launch, weight tiles resident in LDS, tagged-granule hand-offs between the layers "
                         "(wavernn_persist.h); A/B against the 5-launch chain with identical sample streams: "
                         "profiles/r02_wavernn_persistent_ab.json (10.0 vs 16.4 us per step)"
                         if model.last_loop_launches == 1 else "5-launch chain, hipGraph replays"),
            }
            del su, wu
        # ---- secondary: MOL mode (hparams voc_mode = 'MOL': 30 fc3 outputs, mixture-of-logistics sampler) on the production path
        if not args.no_wavernn_unbatched and not args.no_wavernn_mol:
            import types
            from mockingbird_amd.vocoder.wavernn import hparams as whp
            hpm = types.SimpleNamespace(**{k_: getattr(whp, k_) for k_ in dir(whp) if not k_.startswith("_")})
            hpm.voc_mode = "MOL"
            mol = WaveRNNDevice(synth.wavernn_state(synth.WAVERNN_HP_MOL, seed=6)["model_state"], hpm)
            mol.generate_samples(mel[:, :60], True, 2000, 200, seed=1)
            sm = mol.generate_samples(mel, True, target, overlap, seed=2)
            torch.cuda.synchronize()
            resident_m = mol.last_loop_launches == 1
            result["wavernn_mol"] = {
                "workload": f"MOL-mode WaveRNN, mel 80x{F}, batched: {mol.last_plan.n_folds} folds x {mol.last_plan.seq_len} steps; "
                            + ("ONE resident launch (wavernn_pipe16.h since round 4 -- operand pairs on the fp16 pipe, as the RAW "
                               "headline: the F3 role is one workgroup that computes the 30 mixture parameters and samples "
                               "sample_from_discretized_mix_logistic itself; MBHIP_WAVERNN_RESIDENT=exact: wavernn_pipe.h, bit-identical to the chain)"
                               if resident_m else
                               "5-launch chain with fc3 + sample_from_discretized_mix_logistic fused in one launch (wf_fc3_mol_kernel)"),
                "sample_loop_ms": mol.last_loop_ms, "us_per_time_step": mol.last_loop_ms * 1e3 / mol.last_plan.seq_len,
                "loop_launches": mol.last_loop_launches,
                "value": sm.numel() / (mol.last_loop_ms * 1e-3), "unit": "fold samples/s (loop only)"}
            if resident_m:  # A/B partner: the launch chain, same utterance and seed -- bit-identical stream
                os.environ["MBHIP_WAVERNN_RESIDENT"] = "0"
                try:
                    sc = mol.generate_samples(mel, True, target, overlap, seed=2)
                    torch.cuda.synchronize()
                    result["wavernn_mol"]["chain_reference"] = {
                        "us_per_time_step": mol.last_loop_ms * 1e3 / mol.last_plan.seq_len, "loop_launches": mol.last_loop_launches,
                        "identical_stream": bool(torch.equal(sc, sm)),
                        "max_abs_diff_first_40_steps": float((sc[:, :40] - sm[:, :40]).abs().max())}
                    del sc
                finally:
                    os.environ.pop("MBHIP_WAVERNN_RESIDENT", None)
            del mol, sm
        # ---- secondary: WaveRNN throughput mode -- north_star's "batch-32 synthetic input": 32 utterances of
        # mel 80x{F} share ONE sample loop (736 fold columns instead of 23 per launch)
        if not args.no_wavernn_batch:
            nb = 32
            bm = [torch.from_numpy(synth.wavernn_mel(F, seed=100 + u) / 4.0).to(dev) for u in range(nb)]
            model.generate_samples_batch(bm, target, overlap, list(range(nb)))  # untimed: allocates the workspace
            torch.cuda.synchronize()
            t0b = time.perf_counter()
            outs = model.generate_samples_batch(bm, target, overlap, list(range(nb)))
            bw = [model.finish(o, True, overlap, True, wave_len) for o in outs]
            torch.cuda.synchronize()
            tb = time.perf_counter() - t0b
            bp = model.last_batch_plan
            nsmp = sum(len(x) for x in bw)
            result["wavernn_batch32"] = {
                "workload": f"{nb} utterances x mel 80x{F} in one sample loop: {bp.n_folds} folds x {bp.seq_len} steps, "
                            "conditioning + loop + float64 tail + D2H, Philox sampling, fp32 state / results "
                            "(products on the fp16 matrix pipe with error compensation since round 4)",
                "value": nsmp / tb, "unit": "samples/s", "x_realtime": nsmp / tb / 16000.0, "s_total": tb,
                "sample_loop_ms": model.last_loop_ms, "us_per_time_step": model.last_loop_ms * 1e3 / bp.seq_len,
                "us_per_fold_step": model.last_loop_ms * 1e3 / bp.seq_len / bp.n_folds,
                "workspace_GB": bp.workspace_bytes / 1e9,
            }
            # at this width the loop is MFMA-bound, not latency-bound: price it against the fp32 matrix peak.
            # Algorithmic flops per fold and step = the six products of the split chain (table-folded inputs and
            # the GRU tiles' dead fourth row not counted): rnn2 input half, hh1, hh2 (3R x R each), fc1, fc2, fc3
            R_, FC_, C_ = model.cfg.rnn_dims, model.cfg.fc_dims, model.n_classes
            fl = 2.0 * (3 * (3 * R_ * R_) + FC_ * R_ + FC_ * FC_ + C_ * FC_) * bp.n_folds
            tf = fl / (result["wavernn_batch32"]["us_per_time_step"] * 1e-6) / 1e12
            ts3 = os.environ.get("MBHIP_RNN_WIDE", "ts3").startswith("ts3")  # rnn_ts3_body.h (default since round 4) or the fp32 form
            peak = (2500.0 / 3.0) if ts3 else MFMA_F32_PEAK_TFLOPS
            result["wavernn_batch32"]["roofline"] = {
                "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                "frac": tf / peak, "frac_of_fp32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS, "frac_of_fp16_mfma_peak": tf / 2500.0, "traffic": None,
                "kernel": ("mb::rnn_ts3_kernel / rnn_dual_linear_ts3_kernel (rnn_ts3_body.h; whole step: 4 GEMM launches + finish): the "
                           "recurrent GEMMs as error-compensated fp16 MFMA products (three per algorithmic product: ceiling 2500 / 3 TFLOP/s), "
                           "activations split once per workgroup in LDS; samples held to the oracle (test_production_batch*)" if ts3 else
                           "mb::rnn_ts2_kernel / rnn_dual_linear_ts2_kernel (whole step: 4 GEMM launches + finish), fp32 MFMA"),
                "note": "MBHIP_DIAG=ts3_dbg diagnostics (tools/wrn_batch32_dbg.py): of 65 us per step the k loops are 27 (L2 -> L1 operand traffic: "
                        "88 MB per big launch), the epilogues 10, launch ramps + prologues + the elementwise rnn1 launch 28",
                "algorithmic_flops_per_step": fl}
            del outs, bw, bm
            model._ws = None
            torch.cuda.empty_cache()
        # ---- secondary: GAN vocoders.  HiFi-GAN batch 32 x (80,200) (north_star "batch-32 synthetic input"),
        # fp32 MFMA (parity path) and fp16 MFMA (throughput path); Fre-GAN fp16 8 x (80,3000) = the per-GPU
        # share of BASELINE configs[4] (batch 64 over 8 GPUs).
        if not args.no_hifigan:
            from mockingbird_amd.vocoder.gan import GanGenerator

            def time_gan(gen, gm, reps):
                for _ in range(2):
                    gen(gm)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    y = gen(gm)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps, y

            h = synth.HIFIGAN_16K
            st = synth.gan_state(h, "hifigan", seed=3)["generator"]
            gm = torch.from_numpy(synth.mel_input(200, 32, seed=0)).to(dev)
            flops = HIFIGAN_MFLOP_PER_FRAME * 1e6 * 200 * 32
            y32 = None
            # the fp32 path's convs run on the fp16 matrix pipe with error compensation (conv1d.hip: three f16 MFMA products per
            # algorithmic product, fp32-grade results) unless MBHIP_CONV_SPLIT=0: its matrix-pipe ceiling is a third of the f16 peak
            split = os.environ.get("MBHIP_CONV_SPLIT", "1") != "0"
            for dt, peak, key in (("f32", MFMA_F16_PEAK_TFLOPS / 3.0 if split else MFMA_F32_PEAK_TFLOPS, "hifigan"),
                                  ("f16", MFMA_F16_PEAK_TFLOPS, "hifigan_f16")):
                gen = GanGenerator(h, st, 0, dtype=dt)
                ms, y = time_gan(gen, gm, 5 if dt == "f32" else 20)
                entry = {
                    "workload": f"HiFi-GAN V1 16k generator forward, batch 32 x mel (80,200), {dt}"
                                + (" (fp16 storage, fp16 MFMA, fp32 accumulate)" if dt == "f16" else
                                   " storage, fp32-grade results: error-compensated fp16 MFMA (x = xh + 2^-11 xl, w = wh + wl, 3 products), "
                                   "peak = 2500 / 3 TFLOP/s; round 6: the whole generator on time-major tensors (resblock_pair_split.hip: one "
                                   "ResBlock unit per launch, 4 MMA + 4-8 support waves; conv_split_tm.hip: every other conv)"
                                   if split else " storage, fp32-input MFMA"),
                    "dtype": dt, "value": 32 * 200 * 200 / (ms * 1e-3), "unit": "samples/s",
                    "x_realtime": 32 * 200 * 200 / (ms * 1e-3) / 16000.0, "ms_per_batch": ms,
                    "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": peak,
                                 "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / peak, "traffic": None},
                }
                if dt == "f16":  # PMC pass of this object: HBM bytes of one forward against the read-x + write-y floor
                    tr, src = pmc_traffic("hifigan")
                    entry["roofline"].update({"traffic": tr, "traffic_source": src, "traffic_floor_bytes": gan_floor_bytes(h, 32, 200)})
                elif split:  # the fp32-result object's own pass (round 4: profiles/r04_pmc_hifigan_f32.json), when it has been committed
                    tr, src = pmc_traffic("hifigan_f32")
                    if tr is not None:
                        entry["roofline"].update({"traffic": tr, "traffic_source": src})
                if dt == "f32":
                    y32 = y
                else:
                    d = (y.double() - y32.double())
                    entry["vs_f32_path"] = {"rms": float(d.pow(2).mean().sqrt()),
                                            "rel_rms": float(d.pow(2).mean().sqrt() / y32.double().pow(2).mean().sqrt())}
                result[key] = entry
                del gen
            hf = synth.FREGAN_16K
            stf = synth.gan_state(hf, "fregan", seed=4)["generator"]
            gmf = torch.from_numpy(synth.mel_input(3000, 8, seed=1)).to(dev)
            gen = GanGenerator(hf, stf, 1, dtype="f16")
            ms, y = time_gan(gen, gmf, 5)
            fl = FREGAN_MFLOP_PER_FRAME * 1e6 * 3000 * 8
            result["fregan_f16"] = {
                "workload": "BASELINE configs[4] per-GPU share: Fre-GAN generator forward fp16 MFMA, batch 8 x mel (80,3000)",
                "dtype": "f16", "value": 8 * 3000 * 200 / (ms * 1e-3), "unit": "samples/s",
                "x_realtime": 8 * 3000 * 200 / (ms * 1e-3) / 16000.0, "ms_per_batch": ms,
                "roofline": {"bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, "traffic": None},
            }
            tr, src = pmc_traffic("fregan")
            result["fregan_f16"]["roofline"].update({"traffic": tr, "traffic_source": src,
                                                     "traffic_floor_bytes": gan_floor_bytes(hf, 8, 3000, fregan=True)})
            del gen, y, gmf
        # ---- secondary: Tacotron synthesize (BASELINE configs[2]): B=32, ~100 tokens, r=2, 400 frames forced
        if not args.no_tacotron:
            from mockingbird_amd.synthesizer.inference import TacotronDevice
            tst = synth.tacotron_state(seed=3)["model_state"]
            tdev = TacotronDevice(tst, dev)
            seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
            Tt = max(len(q) for q in seqs)
            chars = torch.tensor(np.stack([np.pad(q, (0, Tt - len(q))) for q in seqs])).long().to(dev)
            spk = torch.tensor(np.stack(emb)).to(dev)
            tdev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=1)
            torch.cuda.synchronize()
            t0t = time.perf_counter()
            reps = 3
            for i in range(reps):
                tdev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=2 + i)
            torch.cuda.synchronize()
            tt = (time.perf_counter() - t0t) / reps
            mem, memp = tdev.encode(chars, spk, -1, None, 1)
            torch.cuda.synchronize()
            t0t = time.perf_counter()
            for i in range(reps):
                tdev.decode(mem, memp, chars, 400, 11, seed=2 + i)
            torch.cuda.synchronize()
            td = (time.perf_counter() - t0t) / reps
            loop_ms = getattr(tdev, "last_loop_ms", None)  # HIP events around the decoder loop alone (mb_taco_last_loop_ms)
            post_ms = getattr(tdev, "last_postnet_ms", None)  # HIP events around CBHG postnet + post_proj (mb_taco_last_postnet_ms)
            torch.cuda.synchronize()
            t0e = time.perf_counter()
            for i in range(reps):
                tdev.encode(chars, spk, -1, None, 2 + i)
            torch.cuda.synchronize()
            enc_ms = (time.perf_counter() - t0e) / reps * 1e3
            iters = getattr(tdev, "last_loop_iterations", 200) or 200
            it_us = loop_ms * 1e3 / iters if loop_ms else td * 1e6 / 200
            bytes_it = 81.06e6 + 32 * Tt * (1024 + 128) * 4  # decoder weights + attention memory, per iteration
            # PMC pass of this configuration: the launches of one iteration (lstm runs twice)
            t_launches = getattr(tdev, "last_loop_launches_per_iteration", 7)
            t_f16 = bool(getattr(tdev, "last_loop_f16_products", False))
            # the exact fp32 loop (7 launches, every product on the fp32 pipe) timed in the same run
            _old_diag = os.environ.get("MBHIP_DIAG")
            os.environ["MBHIP_DIAG"] = (_old_diag + "," if _old_diag else "") + "taco_front=0"
            try:
                tdev.decode(mem, memp, chars, 400, 11, seed=1)
                tdev.decode(mem, memp, chars, 400, 11, seed=2)
                torch.cuda.synchronize()
                x_loop_ms, x_launches = getattr(tdev, "last_loop_ms", None), getattr(tdev, "last_loop_launches_per_iteration", None)
            finally:
                if _old_diag is None:
                    os.environ.pop("MBHIP_DIAG", None)
                else:
                    os.environ["MBHIP_DIAG"] = _old_diag
            t_traffic, t_src = (pmc_traffic("tacotron", ["front", "lstm", "lstm", "mel_proj"]) if t_launches == 4 else
                                pmc_traffic("tacotron", ["front", "rnn_input", "lstm", "lstm", "mel_proj"]) if t_launches == 5 else (None, None))
            if t_traffic is None:  # the seven-launch loop (or a PMC pass older than the fused front: same bytes, other kernel names)
                t_traffic, t_src = pmc_traffic("tacotron", ["prenet_fc2", "attn_gru", "lsa", "rnn_input", "lstm", "lstm", "mel_proj"])
            result["tacotron"] = {
                "workload": "Tacotron generate (text encoder + GST + 200 decoder iterations r=2 + CBHG postnet), "
                            f"batch 32 x ~100 tokens (T={Tt}), 400 mel frames forced, fp32 state / accumulate / epilogues, on-device dropout RNG",
                "dtype": ("split-f16 products (22-bit operands: w 2^s = wh + wl, x = xh + 2^-11 xl; three v_mfma_f32_16x16x32_f16 per 32 k) on the K >= 1024 "
                          "tiles of the decoder loop, f32 accumulate / state / epilogues; fc2, attention GRU and attention on the f32 pipe" if t_f16 else
                          "f32 (every product of the decoder loop on the fp32 matrix pipe)"),
                "exact_f32": {"decoder_loop_ms": x_loop_ms, "us_per_decoder_iteration": (x_loop_ms * 1e3 / iters) if x_loop_ms else None,
                              "launches_per_iteration": x_launches, "how": "MBHIP_DIAG=taco_front=0: the 7-launch loop, fp32 MFMA only (also the loop a range event or a lost hand-off falls back to)"},
                "value": 32 * 400 / tt, "unit": "mel frames/s", "x_realtime_at_200_samples_per_frame": 32 * 400 * 200 / tt / 16000.0,
                "ms_per_batch": tt * 1e3, "decode_plus_postnet_ms": td * 1e3, "decoder_loop_ms": loop_ms,
                "us_per_decoder_iteration": it_us, "launches_per_iteration": t_launches,
                "encoder_ms": enc_ms, "postnet_ms": post_ms,
                # SURVEY 8(d) "Tacotron CBHG postnet": MFMA-bound, 16.1 MFLOP per frame; its convs run the error-compensated fp16 products
                # (three MFMA products per algorithmic one: ceiling 2500 / 3 = 833 TFLOP/s)
                "postnet_roofline": ({"bound": "mfma", "kernel": "CBHG postnet + post_proj (tacotron.hip cbhg_forward_tm on conv_split_tm.hip: conv bank as one conv, two projections, 4 highways, "
                                                                  "bidirectional GRU scan gru_scan.h, post_proj) over 32 x 400 frames",
                                      "achieved": 16.1e6 * 400 * 32 / (post_ms * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s",
                                      "frac": 16.1e6 * 400 * 32 / (post_ms * 1e-3) / 1e12 / (MFMA_F16_PEAK_TFLOPS / 3.0),
                                      "frac_of_2500": 16.1e6 * 400 * 32 / (post_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                                      "algorithmic_flops": 16.1e6 * 400 * 32, "ms": post_ms,
                                      "timing": "HIP events on the call's stream around the postnet (mb_taco_last_postnet_ms)", "traffic": None}
                                     if post_ms else None),
                "roofline": {"bound": "hbm", "kernel": f"decoder iteration (taco_fast.h: {t_launches} launches per iteration"
                                                       + (" -- prenet fc2, attention GRU and attention are roles of one launch with tagged-granule "
                                                          "hand-offs, taco_front_kernel" if t_launches in (4, 5) else "") +
                                                       ("; rnn_input folded into the attention role through a memory projected once per call" if t_launches == 4 else "") +
                                                       ("; LSTM / mel launches and the hidden-half riders on fm_gemm16" if t_f16 else "") +
                                                       ", hipGraph replays; 81.06 MB fp32 weights + attention memory per iteration)",
                             "achieved": bytes_it / (it_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": bytes_it / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": t_traffic,
                             "traffic_source": t_src, "algorithmic_bytes_per_iteration": bytes_it,
                             "timing": "HIP events on the loop's stream around the 200 iterations" if loop_ms else
                                       "decode + postnet wall time / 200 (upper bound on the iteration time)"},
            }
        # ---- secondary: ppg2mel voice-conversion decoder: one utterance of 200 encoder
        # frames (800 ppg frames, 8 s) -> 400 forced decoder steps of 2 mel frames; and a batch of 32
        if not args.no_ppg2mel:
            from mockingbird_amd.ppg2mel import Ppg2MelDecoder
            pdec = Ppg2MelDecoder(synth.ppg2mel_decoder_state(synth.PPG2MEL_HP, seed=3, stop_bias=-6.0), synth.PPG2MEL_HP)
            entry = {}
            for pb in (1, 32):
                pmem = torch.from_numpy(synth.ppg2mel_memory(pb, 200, seed=1)).to(dev)
                pdec.decode(pmem, seed=1)
                torch.cuda.synchronize()
                t0p = time.perf_counter()
                for i in range(3):
                    pm, _, _ = pdec.decode(pmem, seed=2 + i)
                torch.cuda.synchronize()
                tp = (time.perf_counter() - t0p) / 3
                steps_p = pm.shape[1]
                lm = getattr(pdec, "last_loop_ms", None)  # HIP events around the loop (mb_ppg2mel_last_loop_ms)
                entry[f"batch{pb}"] = {"ms": tp * 1e3, "steps": int(steps_p), "us_per_step": (lm * 1e3 if lm else tp * 1e6) / steps_p,
                                       "wall_us_per_step": tp * 1e6 / steps_p, "mel_frames_per_s": pb * steps_p * 2 / tp,
                                       "loop_launches": getattr(pdec, "last_loop_launches", None)}
                if getattr(pdec, "last_loop_launches", 0) == 1:
                    entry[f"batch{pb}"]["loop"] = ("ONE resident launch, fp32 fmaf chains on LDS-resident weights (ppg_resident.h)" if pb == 1 else
                                                   "ONE resident launch (ppg_batch.h, round 5): 160 role workgroups + one attention workgroup per utterance, "
                                                   "weights as split fp16 fragments in registers, error-compensated v_mfma_f32_16x16x32_f16, pair-granule "
                                                   "hand-offs, two column groups of 16 in flight")
                _ppg_env = os.environ.get("MBHIP_PPG_RESIDENT")  # (a value the caller set is put back behind the A/B legs)

                def _ppg_restore():
                    if _ppg_env is None:
                        os.environ.pop("MBHIP_PPG_RESIDENT", None)
                    else:
                        os.environ["MBHIP_PPG_RESIDENT"] = _ppg_env
                if pb == 32 and getattr(pdec, "last_loop_launches", 0) == 1:  # A/B partner: the 6-launch chain on the same batch
                    os.environ["MBHIP_PPG_RESIDENT"] = "0"
                    try:
                        pdec.decode(pmem, seed=1)
                        cm, _, _ = pdec.decode(pmem, seed=4)
                        torch.cuda.synchronize()
                        entry["batch32"]["chain_reference"] = {"us_per_step": pdec.last_loop_ms * 1e3 / cm.shape[1], "loop_launches": pdec.last_loop_launches,
                                                               "loop": "6-launch step (ppg_fast.h), hipGraph replays, same batch",
                                                               "mel_max_abs_diff_vs_resident": float((cm - pm).abs().max())}
                    finally:
                        _ppg_restore()
                if pb == 1 and getattr(pdec, "last_loop_launches", 0) == 1:  # A/B partner of the resident loop: the 6-launch chain
                    os.environ["MBHIP_PPG_RESIDENT"] = "0"
                    try:
                        pdec.decode(pmem, seed=1)
                        cm, _, _ = pdec.decode(pmem, seed=4)  # the seed of the last timed resident pass
                        torch.cuda.synchronize()
                        entry["batch1"]["chain_reference"] = {"us_per_step": pdec.last_loop_ms * 1e3 / cm.shape[1], "loop_launches": pdec.last_loop_launches,
                                                              "loop": "6-launch step (ppg_fast.h), hipGraph replays, same utterance",
                                                              "mel_max_abs_diff_vs_resident": float((cm - pm).abs().max())}
                    finally:
                        _ppg_restore()
            # 19.1 MB of fp32 weights are touched once per step (attention LSTM 7.3 MB, decoder LSTM 10.5 MB, rest 1.3 MB)
            wbytes = 4.0 * (256 * 80 + 128 * 256 + 2048 * (384 + 512) + 256 * 512 + 15 * 256 + 2048 * (768 + 512) + 161 * 768)
            entry["workload"] = ("ppg2mel Decoder.inference loop (prenet, attention LSTMCell, MoL attention, decoder LSTMCell, "
                                 "projection + stop), T_enc = 200, 400 steps forced, fp32, on-device dropout RNG")
            resident_p = entry["batch1"].get("loop_launches") == 1
            entry["roofline"] = {"bound": "hbm", "kernel": ("mb::ppg_resident_kernel (ppg_resident.h): the whole loop of the utterance as ONE resident launch -- 217 "
                                                            "role-specialised workgroups, weights in LDS, 5 granule hand-offs per step; algorithmic bytes as if the "
                                                            "weights were streamed once per step (the kernel reads them once per utterance)") if resident_p else
                                 "decoder step (ppg_fast.h: 6 launches per step, hipGraph replays), weights streamed once per step",
                                 "achieved": wbytes / (entry["batch1"]["us_per_step"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": wbytes / (entry["batch1"]["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": None, "algorithmic_bytes_per_step": wbytes}
            try:  # HBM bytes of one step (all six launches) from the committed PMC passes (tools/pmc_r03_ppg.sh)
                ppg_pmc = "r05_pmc_ppg2mel.json" if os.path.exists(os.path.join(ROOT, "profiles", "r05_pmc_ppg2mel.json")) else "r03_pmc_ppg2mel.json"
                pm = json.load(open(os.path.join(ROOT, "profiles", ppg_pmc)))
                pmc_resident = any("ppg_resident_kernel" in k for k in pm.get("kernels", {}))  # round 5 profiled the resident launch, round 3 the chain
                entry["roofline"]["traffic_resident_step" if pmc_resident else "traffic_chain_step"] = pm["step_hbm_bytes_per_launch"]
                entry["roofline"]["traffic_source"] = f"profiles/{ppg_pmc}: " + pm.get("source", "")[:200]
                if resident_p == pmc_resident:  # the counters describe the form that was timed
                    entry["roofline"]["traffic"] = pm["step_hbm_bytes_per_launch"]
            except Exception:
                pass
            result["ppg2mel"] = entry
        # ---- CPU baselines: the reference itself when its checkout is importable (this container), else the oracle
        # port (the GPU box has no /root/reference); torch's default thread count = what the reference would use
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_wavernn(state, F, target, overlap, len(wav), plan, args.cpu_seconds)
            if "hifigan" in result:
                result["hifigan"]["cpu_baseline"] = cpu_baseline_hifigan()
            if "tacotron" in result:
                result["tacotron"]["cpu_baseline"] = cpu_baseline_tacotron()
        if "exact_f32" in result and "roofline" in result and result["roofline"].get("algorithmic_bytes_per_launch"):
            xf = result["exact_f32"]
            xg = result["roofline"]["algorithmic_bytes_per_launch"] / (xf["sample_loop_ms"] * 1e-3) / 1e9
            result["roofline"].update({"exact_f32_samples_per_s": xf["value"], "exact_f32_us_per_step": xf["us_per_time_step"],
                                       "exact_f32_frac": xg / HBM_PEAK_GBS})
        print(json.dumps(_ordered(result)))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _ordered(result):
    """The JSON line with numbers first and prose last (the driver keeps `config` / `roofline` whole but cuts the
    tail of the line): inside every object the strings longer than 100 characters move behind everything else; at the top level the
    contract's keys, `config`, `roofline`, `cpu_baseline`, then the secondary objects, then the notes."""
    def tidy(d):
        if isinstance(d, dict):
            short = {k: tidy(v) for k, v in d.items() if not (isinstance(v, str) and len(v) > 100)}
            short.update({k: v for k, v in d.items() if isinstance(v, str) and len(v) > 100})
            return short
        if isinstance(d, list):
            return [tidy(v) for v in d]
        return d
    first = ("metric", "value", "unit", "x_realtime", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "exact_f32", "pcie_inclusive")
    last = ("per_rank_ms", "dtype_note")
    out = {k: result[k] for k in first if k in result}
    out.update({k: v for k, v in result.items() if k not in first and k not in last})
    out.update({k: result[k] for k in last if k in result})
    return tidy(out)


if __name__ == "__main__":
    main()
