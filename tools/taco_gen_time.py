"""Tacotron generate B=32 timing: python tools/taco_gen_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from mockingbird_amd.synthesizer.inference import TacotronDevice
tst = synth.tacotron_state(seed=3)["model_state"]
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
spk = torch.tensor(np.stack(emb)).cuda()
dev = TacotronDevice(tst, torch.device("cuda"))
dev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=5); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=5)
torch.cuda.synchronize()
print("tacotron generate B=32 ms", (time.perf_counter() - t0) / 5 * 1e3, "env MBHIP_DIAG", os.environ.get("MBHIP_DIAG"))
print("  decoder loop ms", dev.last_loop_ms, " postnet ms", getattr(dev, "last_postnet_ms", None))
