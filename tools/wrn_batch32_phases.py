"""Where the 32-utterance WaveRNN call (bench.py's wavernn_batch32) spends its time outside the sample loop:
conditioning (+ workspace setup) / sample loop / float64 tail + D2H per utterance.  usage: python tools/wrn_batch32_phases.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mels = [torch.from_numpy(synth.wavernn_mel(1000, seed=100 + u) / 4.0).cuda() for u in range(32)]
seeds = list(range(32))
dev.generate_samples_batch(mels, 8000, 800, seeds)
torch.cuda.synchronize()
out = {}
for rep in range(2):
    t0 = time.perf_counter()
    outs = dev.generate_samples_batch(mels, 8000, 800, seeds)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    w = [dev.finish(o, True, 800, True, 200000) for o in outs]
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = {"generate_ms": (t1 - t0) * 1e3, "sample_loop_ms": dev.last_loop_ms, "conditioning_and_setup_ms": (t1 - t0) * 1e3 - dev.last_loop_ms,
           "finish_ms": (t2 - t1) * 1e3, "total_ms": (t2 - t0) * 1e3}
    print(json.dumps(out), flush=True)
