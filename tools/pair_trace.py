"""Phase timeline of the fused ResBlock-pair kernel (MBHIP_DIAG=pair_trace=<file> marks of workgroup 0):
python tools/pair_trace.py  -> runs one HiFi-GAN f16 forward (B=32, F=200) and prints, per distinct
kernel configuration, the median shader-clock cycles between marks of tiles 1..4."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set, diag_get
import numpy as np
path = os.path.join(ROOT, "gpurun_out", "pair_trace.txt")
os.makedirs(os.path.dirname(path), exist_ok=True)
if os.path.exists(path):
    os.remove(path)
import torch, synth
from mockingbird_amd.vocoder.gan import GanGenerator
h = synth.HIFIGAN_16K
gen = GanGenerator(h, synth.gan_state(h, "hifigan", seed=3)["generator"], 0, dtype="f16")
mel = torch.from_numpy(synth.mel_input(200, 32, seed=0)).cuda()
gen(mel); torch.cuda.synchronize()
diag_set("pair_trace", path)
gen(mel); torch.cuda.synchronize()
diag_set("pair_trace")
rows = collections.OrderedDict()
for line in open(path):
    head, marks = line.split(":")
    key = tuple(int(v) for v in head.split())
    m = np.array([int(v) for v in marks.split()], dtype=np.float64).reshape(2, 8, 16)
    rows.setdefault(key, []).append(m)
names_mma = ["wait B", "phase1", "wait W", "epi1", "wait E1", "phase2", "wait P", "epi2", "wait Y"]
names_sup = ["wait B", "x loads issue", "y write-out", "wait W+E1", "x -> LDS", "wait P+Y"]
for key, ms in rows.items():
    m = ms[0]
    C, NTW, TD, ntaps, dil, nbuf, tiles = key
    mma, sup = m[0], m[1]
    tl = [t for t in range(1, 5) if mma[t, 9] > 0 and mma[t + 1, 0] > 0 or (t < 8 and mma[t, 9] > 0)]
    if not tl:
        continue
    d_mma = np.median([[mma[t, k + 1] - mma[t, k] for k in range(9)] for t in tl], axis=0)
    per_tile = np.median([mma[t, 9] - mma[t, 0] for t in tl])
    print(f"C={C} NTW={NTW} TD={TD} k={ntaps} d={dil} nbuf={nbuf} tiles={tiles}: tile {per_tile:.0f} cyc"
          f" | MMA " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names_mma, d_mma)))
    d_sup = np.median([[sup[t, k + 1] - sup[t, k] for k in range(6)] for t in tl if sup[t, 6] > 0], axis=0)
    print("      support " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names_sup, d_sup)))
