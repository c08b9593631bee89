"""Batch-32 WaveRNN loop: ONE loop of 736 fold columns against TWO / FOUR concurrent loops of 368 / 184 columns (one host thread, one
HIP stream and one handle each: their hipGraph replays overlap on the device).  usage: python tools/wrn_batch_streams.py [F] [reps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
st = synth.wavernn_state(seed=1)["model_state"]
nb = 32
mels = [torch.from_numpy(synth.wavernn_mel(F, seed=100 + u) / 4.0).cuda() for u in range(nb)]
for parts in (1, 2, 4, 1):
    devs = [WaveRNNDevice(st) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    per = nb // parts
    outs = [None] * parts

    def work(i):
        with torch.cuda.stream(streams[i]):
            outs[i] = devs[i].generate_samples_batch(mels[i * per:(i + 1) * per], 8000, 800, list(range(i * per, (i + 1) * per)))
            streams[i].synchronize()

    best = None
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r > 0:
            best = dt if best is None else min(best, dt)
    nsmp = sum(o.numel() for oo in outs for o in oo)
    steps = outs[0][0].shape[1]
    print(f"parts {parts}: wall {best*1e3:.1f} ms, {nsmp/best/1e6:.2f} M fold samples/s, equivalent {best*1e6/steps:.1f} us per step of all 32 utterances; "
          f"per-loop us/step {[round(d.last_loop_ms*1e3/steps,1) for d in devs]}", flush=True)
    del devs
