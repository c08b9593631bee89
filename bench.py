#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native MockingBird hot path.

Metric (BASELINE.json): audio samples/s (16 kHz) and xRT.  Workload at N=1 = BASELINE
configs[1]: WaveRNN 9-bit mu-law, batch 1, (80, 1000) mel, batched generation (target 8000 /
overlap 800 -> 23 folds x 9600 steps).  One "step" = one full infer_waveform-equivalent pass
(conditioning networks + sample loop + device->host + float64 post-processing) over one
utterance per GPU, mel already resident in HBM.  N>1: every rank vocodes its own utterance
(weak scaling, no data-path collective) and finished waveforms are gathered to all ranks with
RCCL (lengths + padded waveforms), inside the timed region.

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
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    import torch.distributed as dist
    use_dist = world > 1
    if use_dist:
        # N > 1: the line is the sharded headline workload + its roofline; the secondary single-GPU objects and
        # the CPU baseline are N = 1 material (rank 0 would otherwise keep the other ranks waiting ~1 min)
        args.no_hifigan = args.no_tacotron = args.no_ppg2mel = args.no_cpu_baseline = args.no_wavernn_batch = True
        args.no_wavernn_unbatched = True
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from mockingbird_amd import build, _lib
    if rank == 0:
        build.build(verbose=False)
    if use_dist:
        dist.barrier()
    import synth
    from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
    from mockingbird_amd import sharding

    state = synth.wavernn_state(seed=5)["model_state"]
    model = WaveRNNDevice(state)
    F = args.frames
    target, overlap = 8000, 800
    # each rank owns one utterance (seed differs per rank), already resident in HBM
    mel = torch.from_numpy(synth.wavernn_mel(F, seed=1 + rank) / 4.0).to(dev)
    wave_len = (F - 1) * model.hop_length

    def one_pass(seed):
        samples = model.generate_samples(mel, True, target, overlap, seed=seed)
        wav = model.finish(samples, True, overlap, True, wave_len)  # float64 tail on the device, D2H of the waveform
        if use_dist:
            wavs = sharding.gather_waveforms([wav.astype(np.float32)], dev)
            return wav, sum(len(w) for w in wavs)
        return wav, len(wav)

    for i in range(args.warmup):
        one_pass(1000 + i)
    loop_ms = []
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total_samples = 0
    for i in range(args.steps):
        wav, n = one_pass(i)
        total_samples += n
        loop_ms.append(model.last_loop_ms)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    else:
        total_samples = total_samples  # single rank: its own utterance
    plan = model.last_plan
    value = total_samples / elapsed
    result = {
        "metric": "audio samples/sec (16 kHz), WaveRNN vocoder.infer_waveform",
        "value": value, "unit": "samples/s", "x_realtime": value / 16000.0,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1000.0, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: WaveRNN 9-bit mu-law RAW, batch=1 utterance/GPU, "
                               f"mel 80x{F}, batched target=8000 overlap=800 -> {plan.n_folds} folds x "
                               f"{plan.seq_len} steps, Philox sampling, fp32 weights (synthetic, seeded)",
                   "samples_per_utterance": int(len(wav)), "utterances": world,
                   "sample_loop_ms": float(np.median(loop_ms)),
                   "us_per_time_step": float(np.median(loop_ms)) * 1000.0 / plan.seq_len},
    }

    if rank == 0:
        # ---- roofline of the dominant loop kernel: rnn_rowtile_kernel<GRU> (rnn1), HBM/L2-bound on weights
        L = _lib.lib()
        ws = model._ws
        smp = torch.empty(plan.n_folds, plan.seq_len, device=dev)
        per_kernel = {}
        split = os.environ.get("MBHIP_WAVERNN_CHAIN", "") != "classic"
        names = (("gru1_finish", "rnn2_input_half", "fc1+hh1", "fc2+hh2", "fc3_sampler") if split
                 else ("rnn1_gru", "rnn2_gru", "fc1", "fc2", "fc3_sampler"))
        for which, name in enumerate(names):
            us, ab = C.c_float(), C.c_double()
            _lib.check(L.mb_wavernn_bench_kernel(model._h, C.byref(plan), _lib.ptr(mel), _lib.ptr(smp),
                                                 _lib.ptr(ws), ws.numel(), which, 0, C.byref(us), C.byref(ab),
                                                 _lib.stream_ptr()), "mb_wavernn_bench_kernel")
            torch.cuda.synchronize()
            per_kernel[name] = {"avg_us": us.value, "algorithmic_bytes": ab.value,
                                "GBps": ab.value / (us.value * 1e-6) / 1e9}
        dom_name = "fc1+hh1" if split else "rnn2_gru"
        dom = per_kernel[dom_name]
        # HBM traffic of that kernel from the committed rocprofv3 PMC pass (FETCH_SIZE doubled per
        # MI355X_MICROARCH.md's gfx950 correction + WRITE_SIZE); counters cannot be read in-process
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_wavernn.json")))
            traffic = pm.get(("fc1_hh1" if split else "rnn2_gru") + "_hbm_bytes_per_launch")
        except Exception:
            pass
        result["roofline"] = {
            "kernel": ("mb::rnn_dual_linear_kernel<4, ..., 4, ...> (WaveRNN fc1 beside the hidden half of the next step's "
                       "rnn1, in-situ marginal duration)" if split else
                       "mb::rnn_rowtile_kernel<EPI_GRU, 1, 8, ...> (WaveRNN rnn2 instance, in-situ marginal duration)"),
            "chain": "split-hidden" if split else "classic",
            "bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "avg_launch_us": dom["avg_us"],
            "whole_step": {"algorithmic_bytes": 16.3e6 + 452.0 * plan.n_folds,
                           "us": result["config"]["us_per_time_step"],
                           "GBps": (16.3e6 + 452.0 * plan.n_folds) / (result["config"]["us_per_time_step"] * 1e-6) / 1e9},
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
            }
            del su, wu
        # ---- secondary: WaveRNN throughput mode -- north_star's "batch-32 synthetic input": 32 utterances of
        # mel 80x{F} share ONE sample loop (736 fold columns instead of 23 per launch)
        if not args.no_wavernn_batch:
            nb = 32
            bm = [torch.from_numpy(synth.wavernn_mel(F, seed=100 + u) / 4.0).to(dev) for u in range(nb)]
            model.generate_samples_batch(bm, target, overlap, list(range(nb)))  # untimed: allocates the 53 GB of tables
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
                            "conditioning + loop + float64 tail + D2H, Philox sampling, fp32",
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
            result["wavernn_batch32"]["roofline"] = {
                "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                "kernel": "mb::rnn_ts2_kernel / rnn_dual_linear_ts2_kernel (whole step: 4 GEMM launches + finish)",
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
            for dt, peak, key in (("f32", MFMA_F32_PEAK_TFLOPS, "hifigan"), ("f16", MFMA_F16_PEAK_TFLOPS, "hifigan_f16")):
                gen = GanGenerator(h, st, 0, dtype=dt)
                ms, y = time_gan(gen, gm, 5 if dt == "f32" else 20)
                entry = {
                    "workload": f"HiFi-GAN V1 16k generator forward, batch 32 x mel (80,200), {dt} MFMA"
                                + (" (fp16 storage, fp32 accumulate)" if dt == "f16" else ""),
                    "dtype": dt, "value": 32 * 200 * 200 / (ms * 1e-3), "unit": "samples/s",
                    "x_realtime": 32 * 200 * 200 / (ms * 1e-3) / 16000.0, "ms_per_batch": ms,
                    "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": peak,
                                 "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / peak, "traffic": None},
                }
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
            result["tacotron"] = {
                "workload": "Tacotron generate (text encoder + GST + 200 decoder iterations r=2 + CBHG postnet), "
                            f"batch 32 x ~100 tokens (T={Tt}), 400 mel frames forced, fp32, on-device dropout RNG",
                "value": 32 * 400 / tt, "unit": "mel frames/s", "x_realtime_at_200_samples_per_frame": 32 * 400 * 200 / tt / 16000.0,
                "ms_per_batch": tt * 1e3, "decode_plus_postnet_ms": td * 1e3,
                "roofline": {"bound": "hbm", "kernel": "decoder iteration (9 launches; 81.06 MB fp32 weights + attention memory)",
                             "achieved": (81.06e6 + 32 * Tt * (1024 + 128) * 4) * 200 / td / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": (81.06e6 + 32 * Tt * (1024 + 128) * 4) * 200 / td / 1e9 / HBM_PEAK_GBS,
                             "traffic": None, "note": "upper bound on the loop's rate: the postnet time is inside decode_plus_postnet_ms"},
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
                entry[f"batch{pb}"] = {"ms": tp * 1e3, "steps": int(steps_p), "us_per_step": tp * 1e6 / steps_p,
                                       "mel_frames_per_s": pb * steps_p * 2 / tp}
            # 19.1 MB of fp32 weights are touched once per step (attention LSTM 7.3 MB, decoder LSTM 10.5 MB, rest 1.3 MB)
            wbytes = 4.0 * (256 * 80 + 128 * 256 + 2048 * (384 + 512) + 256 * 512 + 15 * 256 + 2048 * (768 + 512) + 161 * 768)
            entry["workload"] = ("ppg2mel Decoder.inference loop (prenet, attention LSTMCell, MoL attention, decoder LSTMCell, "
                                 "projection + stop), T_enc = 200, 400 steps forced, fp32, on-device dropout RNG")
            entry["roofline"] = {"bound": "hbm", "kernel": "decoder step (8 launches), weights streamed once per step",
                                 "achieved": wbytes / (entry["batch1"]["us_per_step"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": wbytes / (entry["batch1"]["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": None, "algorithmic_bytes_per_step": wbytes}
            result["ppg2mel"] = entry
        # ---- CPU baseline: the oracle (reference's ATen CPU arithmetic) on a bounded sample
        if not args.no_cpu_baseline:
            from oracle import wavernn as ow
            w = dict(state)
            with torch.no_grad():
                mels, aux = ow.conditioning(w, ow.HP, torch.from_numpy(synth.wavernn_mel(F, seed=1)[None, :, :] / 4.0),
                                            True, target, overlap)
                nf = mels.shape[0]
                # The reference would run with torch's default thread count (= all host cores); on a
                # many-core host that oversubscribes these tiny GEMVs, so probe a few settings and
                # time the bounded sample with the fastest one (reported in `cores`).
                ncores = os.cpu_count() or 1
                best_t, best_rate = 1, 0.0
                for nt in sorted({1, 4, 8, 16, min(32, ncores)}):
                    if nt > ncores:
                        continue
                    torch.set_num_threads(nt)
                    ow.sample_loop(w, ow.HP, mels[:, :2], aux[:, :2])
                    tp = time.perf_counter()
                    ow.sample_loop(w, ow.HP, mels[:, :8], aux[:, :8])
                    rate = 8 / (time.perf_counter() - tp)
                    if rate > best_rate:
                        best_t, best_rate = nt, rate
                torch.set_num_threads(best_t)
                t0c = time.perf_counter()
                steps_done = 0
                chunk = 10
                while time.perf_counter() - t0c < args.cpu_seconds and steps_done + chunk <= mels.shape[1]:
                    ow.sample_loop(w, ow.HP, mels[:, steps_done:steps_done + chunk], aux[:, steps_done:steps_done + chunk])
                    steps_done += chunk
                tc = time.perf_counter() - t0c
            raw_rate = nf * steps_done / tc
            useful = len(wav) / float(plan.n_folds * plan.seq_len)
            result["cpu_baseline"] = {
                "value": raw_rate * useful, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"oracle sample loop (reference ATen-CPU ops), {nf} folds x {steps_done} steps of the same "
                          f"workload in {tc:.1f} s; raw {raw_rate:.0f} fold-steps/s scaled by the useful-sample "
                          f"fraction {useful:.3f}",
            }
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
