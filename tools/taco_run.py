"""One Tacotron configs[2] pass (B=32, ~100 tokens, 400 forced steps) for profiling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from mockingbird_amd.synthesizer.inference import TacotronDevice
st = synth.tacotron_state(seed=3)["model_state"]
dev = TacotronDevice(st, torch.device("cuda"))
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
spk = torch.tensor(np.stack(emb)).cuda()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m, l, a = dev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=1)
    torch.cuda.synchronize(); print("generate s", time.perf_counter() - t0, m.shape)
