"""Secondary measurements for DESIGN.md (every BASELINE.json config that fits one GPU), each beside
the CPU oracle (reference ATen-CPU arithmetic) on the same host.  Writes gpurun_out/bench_all.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import synth
from oracle import gan as og, tacotron as ot

CPU_THREADS = int(os.environ.get("BENCH_CPU_THREADS", "16"))
out = {"cpu_threads": CPU_THREADS, "host_cores": os.cpu_count()}


def gpu_time(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def cpu_time(fn, reps=2):
    torch.set_num_threads(CPU_THREADS)
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


from mockingbird_amd.vocoder.gan import GanGenerator
for kind, cfg, mflop in (("hifigan", synth.HIFIGAN_16K, 352.1), ("fregan", synth.FREGAN_16K, 385.1)):
    st = synth.gan_state(cfg, kind, seed=3)
    gen = GanGenerator(cfg, st["generator"], 0 if kind == "hifigan" else 1)
    w = og.fold_weight_norm_state(st["generator"])
    fwd = og.hifigan_forward if kind == "hifigan" else og.fregan_forward
    for B, F in ((1, 200), (32, 200), (8, 3000)):
        mel = torch.from_numpy(synth.mel_input(F, B, seed=0))
        mg = mel.cuda()
        tg = gpu_time(lambda: gen(mg), reps=3)
        rec = {"gpu_s": tg, "samples_per_s": B * F * 200 / tg, "xRT": B * F * 200 / tg / 16000,
               "TFLOPs": mflop * 1e6 * F * B / tg / 1e12, "frac_f32_mfma_peak": mflop * 1e6 * F * B / tg / 1e12 / 157.3}
        if B * F <= 6400 and B == 1:
            with torch.no_grad():
                tc = cpu_time(lambda: fwd(w, cfg, mel))
            rec.update({"cpu_s": tc, "cpu_samples_per_s": B * F * 200 / tc, "cpu_xRT": B * F * 200 / tc / 16000})
        out[f"{kind}_B{B}_F{F}"] = rec
        print(kind, B, F, rec, flush=True)
    del gen

# Tacotron configs[2]: B=32 (one batch) ~100 tokens, r=2, 400 steps forced
from mockingbird_amd.synthesizer.inference import TacotronDevice
from mockingbird_amd.synthesizer.hparams import hparams
st = synth.tacotron_state(seed=3)["model_state"]
dev = TacotronDevice(st, torch.device("cuda"))
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long()
spk = torch.tensor(np.stack(emb))
cg, sg = chars.cuda(), spk.cuda()
tg_full = gpu_time(lambda: dev.generate(cg, sg, steps=400, style_idx=-1, min_stop_token=11, seed=1), reps=3)
mem, memp = dev.encode(cg, sg, -1, None, 1)
tg_dec = gpu_time(lambda: dev.decode(mem, memp, cg, 400, 11, seed=1), reps=3)
rec = {"gpu_generate_s": tg_full, "gpu_decode_postnet_s": tg_dec, "frames_per_s": 32 * 400 / tg_full,
       "xRT_at_200_samples_per_frame": 32 * 400 * 200 / tg_full / 16000, "decoder_us_per_iteration_incl_postnet": tg_dec / 200 * 1e6}
with torch.no_grad():
    torch.set_num_threads(CPU_THREADS)
    t0 = time.perf_counter()
    ot.generate(st, ot.HP, 2, chars, spk, steps=400, style_idx=-1, min_stop_token=11)
    tc = time.perf_counter() - t0
rec.update({"cpu_generate_s": tc, "cpu_frames_per_s": 32 * 400 / tc})
out["tacotron_B32_T100_steps400"] = rec
print("tacotron", rec, flush=True)

# maximum_path
from mockingbird_amd.monotonic_align import maximum_path
from oracle import maximum_path as omp
rng = np.random.default_rng(0)
b, tt, ts = 32, 1000, 200
neg = rng.standard_normal((b, tt, ts)).astype(np.float32)
mask = np.ones((b, tt, ts), np.float32)
ng, mgk = torch.from_numpy(neg).cuda(), torch.from_numpy(mask).cuda()
tg = gpu_time(lambda: maximum_path(ng, mgk), reps=5)
t0 = time.perf_counter(); omp.maximum_path(neg, mask); tc = time.perf_counter() - t0
out["maximum_path_b32_1000x200"] = {"gpu_s": tg, "cpu_s": tc, "cells_per_s": b * tt * ts / tg, "GBps_12B_per_cell": b * tt * ts * 12 / tg / 1e9}
print("maximum_path", out["maximum_path_b32_1000x200"], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_all.json"), "w"), indent=1)
