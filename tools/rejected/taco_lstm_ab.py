"""A/B of the merged LSTM launch (taco_lstm2_kernel) (MBHIP_TACO_LSTM_MERGED=1) against the default two launches; `dbg`: also without its fences (timing only): configs[2] decode loop,
B = 32, 200 iterations forced, HIP events around the loop (mb_taco_last_loop_ms), alternating, same process."""
import os, sys
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
mem, memp = dev.encode(chars, spk, -1, None, 1)
ref = None
for rep in range(3):
    for name, env, dbg in (("merged", "1", None), ("split", None, None)) + ((("noacq", "1", "1"), ("norel", "1", "2"), ("nofence", "1", "3")) if len(sys.argv) > 1 else ()):
        os.environ.pop("MBHIP_TACO_LSTM_MERGED", None)
        os.environ.pop("MBHIP_TACO_L2_DBG", None)
        if env:
            os.environ["MBHIP_TACO_LSTM_MERGED"] = env
        if dbg:
            os.environ["MBHIP_TACO_L2_DBG"] = dbg
        dev.decode(mem, memp, chars, 400, 11, seed=1)
        mel, _, _ = dev.decode(mem, memp, chars, 400, 11, seed=1)
        us = dev.last_loop_ms * 1e3 / dev.last_loop_iterations
        same = None if ref is None else bool(torch.equal(ref, mel.cpu()))
        ref = mel.cpu() if ref is None else ref
        print(rep, name, "us per iteration %.2f" % us, "identical" if same else same, flush=True)
