"""Wall-clock marks (10 ns) of the persistent WaveRNN kernel, steps 1000..1003, workgroups 0 (fc1 side) and 32 (fc2/fc3 side)."""
import os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set, diag_get
os.environ["MBHIP_WAVERNN_RESIDENT"] = "1"
diag_set("wp_trace", "/tmp/wp_trace.bin")
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(12, seed=0) / 4.0).cuda()
dev.generate_samples(mel, False, 0, 0, seed=5)
torch.cuda.synchronize()
m = struct.unpack("128Q", open("/tmp/wp_trace.bin", "rb").read())
names = ["step start", "keys in", "A barrier", "A done", "finish done", "B barrier", "rnn2 done", "gather done", "fc gemm done", "y2 gather", "fc3 gemm", "rnn2 gemm", "P2 in"]
for wg in range(2):
    base = m[(wg * 4) * 16]
    for st in range(4):
        row = m[(wg * 4 + st) * 16:(wg * 4 + st) * 16 + 13]
        print(f"wg {'0 ' if wg == 0 else '32'} step {1000 + st}: " + "  ".join(f"{names[k]}={(row[k] - base) * 0.01:.2f}" for k in (0, 1, 2, 3, 4, 5, 11, 12, 6, 7, 8, 9, 10) if row[k]))
