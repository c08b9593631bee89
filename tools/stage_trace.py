"""Phase timeline of the one-launch ResBlock group kernel (resblock_stage_f16.hip; MBHIP_DIAG=stage_trace=<file> marks of thread 0 of
workgroup 0, tile 1).  Needs the trace build:  MODULE=resblock_stage_f16 tools/build_variant.sh stagetrace -DMB_STAGE_TRACE_BUILD
and MBHIP_LIB=build_variants/libmbhip_stagetrace.so.  Prints shader-clock cycles per chain / unit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set, diag_get
import numpy as np
path = os.path.join(ROOT, "gpurun_out", "stage_trace.txt")
os.makedirs(os.path.dirname(path), exist_ok=True)
if os.path.exists(path):
    os.remove(path)
import torch, synth
from mockingbird_amd.vocoder.gan import GanGenerator
kind = sys.argv[1] if len(sys.argv) > 1 else "hifigan"
h = synth.HIFIGAN_16K if kind == "hifigan" else synth.FREGAN_16K
gen = GanGenerator(h, synth.gan_state(h, kind, seed=3)["generator"], 0 if kind == "hifigan" else 1, dtype="f16")
B, F = (32, 200) if kind == "hifigan" else (8, 3000)
mel = torch.from_numpy(synth.mel_input(F, B, seed=0)).cuda()
gen(mel); torch.cuda.synchronize()
diag_set("stage_trace", path)
gen(mel); torch.cuda.synchronize()
diag_set("stage_trace")
names = ["conv1", "epi1", "wait E1", "conv2", "epi2|wait P", "wait E2|epi O"]
for line in open(path):
    head, marks = line.split(":")
    C, NTW, nch, nun, tiles = (int(v) for v in head.split())
    m = np.array([int(v) for v in marks.split()], dtype=np.float64).reshape(4, 4, 8)
    print(f"C={C} NTW={NTW} chains={nch} units={nun} tiles={tiles}")
    t_first = m[0, 0, 0]
    for c in range(nch):
        for u in range(nun):
            d = [m[c, u, k + 1] - m[c, u, k] for k in range(6)]
            nxt = m[c, u + 1, 0] if u + 1 < nun else (m[c + 1, 0, 0] if c + 1 < nch else 0)
            gap = nxt - m[c, u, 6] if nxt else float("nan")
            print(f"  chain {c} unit {u}: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d)) + f", to next {gap:.0f}")
    print(f"  chains total {m[nch - 1, nun - 1, 6] - t_first:.0f} cycles")
