"""Stress of the persistent WaveRNN kernel's hand-offs: the same utterance many times, alone and while another stream keeps the
GPU busy with GEMMs of varying size (uneven load, workgroups competing for compute units); every run must reproduce the
launch chain's samples exactly, whatever path the call ended up on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(40, seed=0) / 4.0).cuda()
os.environ["MBHIP_WAVERNN_PERSIST"] = "0"
ref = dev.generate_samples(mel, False, 0, 0, seed=9)
os.environ.pop("MBHIP_WAVERNN_PERSIST")
side = torch.cuda.Stream()
bad = 0
for rep in range(12):
    load = rep % 3  # 0: idle, 1: small GEMMs, 2: large GEMMs on the side stream
    stop_at = time.time() + 0.25
    if load:
        n = 512 if load == 1 else 4096
        a = torch.randn(n, n, device="cuda"); b = torch.randn(n, n, device="cuda")
        with torch.cuda.stream(side):
            for _ in range(2000 if load == 1 else 150):
                a = (a @ b) * 1e-3
    t0 = time.perf_counter()
    out = dev.generate_samples(mel, False, 0, 0, seed=9)
    torch.cuda.synchronize()
    same = bool(torch.equal(ref, out))
    bad += not same
    print(f"rep {rep} load {load}: {'same' if same else 'DIFFERENT'} launches {dev.last_loop_launches} "
          f"us/step {dev.last_loop_ms * 1e3 / out.shape[1]:.2f} wall {time.perf_counter() - t0:.3f}s", flush=True)
print("FAILED" if bad else "ok")
