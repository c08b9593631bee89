import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.ppg2mel import Ppg2MelDecoder
pdec = Ppg2MelDecoder(synth.ppg2mel_decoder_state(synth.PPG2MEL_HP, seed=3, stop_bias=-6.0), synth.PPG2MEL_HP)
pb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mem = torch.from_numpy(synth.ppg2mel_memory(pb, 200, seed=1)).cuda()
for i in range(3):
    pdec.decode(mem, seed=1)
torch.cuda.synchronize()
