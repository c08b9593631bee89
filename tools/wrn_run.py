"""One WaveRNN pass of the benchmarked BASELINE configs[1] shape (mel 80 x FRAMES, batched: 23 folds x 9600 steps at
1000 frames) for profiling: python tools/wrn_run.py [frames] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(F, seed=0) / 4.0).cuda()
for _ in range(reps):
    t0 = time.perf_counter()
    s = dev.generate_samples(mel, True, 8000, 800, seed=1)
    torch.cuda.synchronize()
    print("generate s", time.perf_counter() - t0, tuple(s.shape), "loop ms", dev.last_loop_ms)
