#!/bin/bash
# round 6: encoder CBHG on the time-major convs (A/B) + tacotron tests; GAN branch streams A/B
for d in "" taco_post_cm; do echo "== MBHIP_DIAG=$d"; MBHIP_DIAG=$d python tools/taco_gen_time.py 2>&1 | tail -2; done
python - <<'PY'
import os, sys, time
sys.path[:0] = ['.', 'tests']
import numpy as np, torch, synth
from mockingbird_amd.synthesizer.inference import TacotronDevice
tst = synth.tacotron_state(seed=3)["model_state"]
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
spk = torch.tensor(np.stack(emb)).cuda()
for d in ("", "taco_post_cm"):
    os.environ["MBHIP_DIAG"] = d
    dev = TacotronDevice(tst, torch.device("cuda"))
    for _ in range(3): dev.encode(chars, spk, style_idx=-1, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dev.encode(chars, spk, style_idx=-1, seed=1)
    torch.cuda.synchronize(); print("encode ms", d, (time.perf_counter() - t0) / 20 * 1e3)
PY
bash tools/sessions/gpu_r06_x.sh
python -m pytest tests/test_tacotron_gpu.py -x -q -m gpu 2>&1 | tail -4
export MBHIP_LIB=build_variants/libmbhip_cttrace.so
for a in "512 1024 400 1 32 split" "1024 512 400 3 32 split" "512 384 400 1 32 split"; do echo "== ctm $a"; python tools/ctm_trace.py $a 2>&1 | tail -8; done
