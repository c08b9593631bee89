"""Where a wide-batch launch of rnn_ts3_body.h spends its time: the batch-32 loop with the k loop cut to two stages (MBHIP_DIAG=ts3_dbg=1),
without epilogues (2), with neither (3).  Diagnostics: the samples are wrong in those runs.  usage: python tools/wrn_batch32_dbg.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set, diag_get
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mels = [torch.from_numpy(synth.wavernn_mel(1000, seed=100 + u) / 4.0).cuda() for u in range(32)]
seeds = list(range(500, 532))
out = {}
for dbg in ("0", "1", "2", "3", "0"):
    diag_set("ts3_dbg", dbg)
    outs = dev.generate_samples_batch(mels, 8000, 800, seeds)
    torch.cuda.synchronize()
    out.setdefault(dbg, []).append(dev.last_loop_ms * 1e3 / outs[0].shape[1])
    print("dbg", dbg, "us per step", out[dbg][-1], flush=True)
    del outs
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wrn_batch32_dbg.json"), "w"), indent=1)
