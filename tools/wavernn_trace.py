"""Per-kernel device timeline of the WaveRNN sample loop (MBHIP_DIAG=trace_file=<file> diagnostics):
first-wave start / last-store end of each of the 6 launches of a step, 100 MHz wall clock."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set, diag_get
import numpy as np, torch
import synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
batched = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path = os.path.join(ROOT, "gpurun_out", "wavernn_trace.bin")
model = WaveRNNDevice(synth.wavernn_state(seed=5)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(frames, seed=1) / 4.0).cuda()
model.generate_samples(mel, bool(batched), 8000, 800, seed=0)
diag_set("trace_file", path)
model.generate_samples(mel, bool(batched), 8000, 800, seed=1)
diag_set("trace_file")
p = model.last_plan
print("folds", p.n_folds, "seq", p.seq_len, "us/step", model.last_loop_ms * 1e3 / p.seq_len)
raw = np.fromfile(path, dtype=np.uint64).reshape(-1, 6, 512, 2)
marks = raw[:, :, 496:500, :].reshape(raw.shape[0], 6, 8).astype(np.float64)
if marks[8:, :5, 6].min() > 0 and marks[8:, :5, :7].max() < 1e18:
    dm = (marks[:, :, 1:7] - marks[:, :, 0:6])  # shader cycles between consecutive marks
    print("marks (wave 0 of workgroup 0, us @2.4GHz): issue | loads-arrive | mfma | lds+barrier | epi-wait+reduce | math+store")
    for k in range(5):
        print("   ", ["rnn1", "rnn2", "fc1", "fc2", "fc3"][k], (np.median(dm[8:, k], axis=0) / 2400.0).round(2).tolist())
clk = raw[:, :, 511, :].astype(np.float64)
b0 = raw[:, :, 0, :].astype(np.float64)
mhz = (clk[..., 1] - clk[..., 0]) / np.maximum(b0[..., 1] - b0[..., 0], 1) * 100.0
print("shader clock MHz inside workgroup 0 (median per kernel)", np.median(mhz[8:], axis=0).round(0).tolist())
raw = raw.copy(); raw[:, :, 496:500, 0] = np.uint64(0xFFFFFFFFFFFFFFFF); raw[:, :, 496:500, 1] = 0; raw[:, :, 511, 0] = np.uint64(0xFFFFFFFFFFFFFFFF); raw[:, :, 511, 1] = 0
st_all = raw[..., 0].astype(np.float64); st_all[raw[..., 0] == np.uint64(0xFFFFFFFFFFFFFFFF)] = np.inf
en_all = raw[..., 1].astype(np.float64)
nblk = (raw[0, :, :, 1] != 0).sum(axis=1)
t = np.stack([st_all.min(axis=2), en_all.max(axis=2)], axis=2)
base = t[0, 0, 0]
t = (t - base).astype(np.int64)
# spread of workgroup starts / per-workgroup lifetime
spread = (np.where(np.isfinite(st_all), st_all, -np.inf).max(axis=2) - st_all.min(axis=2)) * 0.01
life = np.where(en_all > 0, en_all - st_all, np.nan) * 0.01
print("workgroups per launch", nblk.tolist())
print("start spread (us, median over steps)", np.median(spread[8:], axis=0).round(2).tolist())
print("per-workgroup lifetime (us, median)", np.nanmedian(life[8:], axis=(0, 2)).round(2).tolist(), "max", np.nanmax(np.nanmedian(life[8:], axis=0), axis=1).round(2).tolist())
names = ["rnn1", "rnn2", "fc1", "fc2", "fc3", "sample"]
tick = 0.01  # us per tick (100 MHz)
st, en = t[:, :, 0], t[:, :, 1]
body = (en - st) * tick
gap_next = np.empty_like(body)
gap_next[:, :5] = (st[:, 1:] - en[:, :5]) * tick
gap_next[:-1, 5] = (st[1:, 0] - en[:-1, 5]) * tick
gap_next[-1, 5] = np.nan
sl = slice(8, -1)
res = {}
live = [k for k in range(6) if nblk[k] > 0]
for k, n in enumerate(names):
    if k not in live:
        continue
    if k == live[-1]:
        gap_next[:-1, k] = (st[1:, 0] - en[:-1, k]) * tick
        gap_next[-1, k] = np.nan
    res[n] = dict(body_us=float(np.median(body[sl, k])), gap_after_us=float(np.nanmedian(gap_next[sl, k])))
    print(f"{n:7s} body {res[n]['body_us']:6.2f} us   gap-after {res[n]['gap_after_us']:6.2f} us")
step = (st[1:, 0] - st[:-1, 0]) * tick
print("step (start-to-start) median", float(np.median(step[sl])), "us")
json.dump(dict(folds=p.n_folds, us_per_step=model.last_loop_ms * 1e3 / p.seq_len, kernels=res,
               step_us=float(np.median(step[sl]))), open(os.path.join(ROOT, "gpurun_out", f"wavernn_trace_b{batched}.json"), "w"), indent=1)
