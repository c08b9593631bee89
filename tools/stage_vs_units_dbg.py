"""Diagnostics for tests/test_resblock_stage_gpu.py::test_resblock_stage_equals_unit_launches: the stage launch and the nine
per-unit launches each against the ATen reference, several repetitions; where they differ."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import hiputil
import test_resblock_stage_gpu as ts

B, C, T = 1, 32, 3000
x = ts._rand(B, C, T, seed=3)
chains = ts._make(C, (3, 7, 11), ((1, 3, 5),) * 3, seed=11)
ref = ts._ref(x, chains, 0.1)
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if warm:  # leave other kernels' residue in LDS first
    import torch.nn.functional as F
    for s in range(3):
        xx = ts._rand(2, 80, 333, seed=31 + s)
        hiputil.conv1d_f16_hip(xx, ts._rand(512, 80, 7, seed=32) / 23.7, ts._rand(512, seed=33), pad=3)
        xp = ts._rand(3, 32, 2049, seed=34)
        hiputil.conv1d_f16_hip(xp, ts._rand(1, 32, 7, seed=35) / 15.0, ts._rand(1, seed=36), pad=3, in_act=1, in_slope=0.01, out_act=2, y_f32=True)
        hiputil.conv1d_f16_hip(ts._rand(2, 64, 1000, seed=41), ts._rand(32, 64, 1, seed=42) / 8, ts._rand(32, seed=43), in_repeat=2)
for rep in range(4):
    y = hiputil.resblock_stage_f16_hip(x, chains, slope=0.1)
    acc = None
    for units in chains:
        xr = x
        for i, (w1, b1, w2, b2, d) in enumerate(units):
            last = i == len(units) - 1
            if last:
                acc = hiputil.resblock_pair_f16_hip(xr, w1, b1, w2, b2, dilation=d, out_scale=1.0 / 3.0, accumulate_into=acc) \
                    if acc is not None else hiputil.resblock_pair_f16_hip(xr, w1, b1, w2, b2, dilation=d, out_scale=1.0 / 3.0)
            else:
                xr = hiputil.resblock_pair_f16_hip(xr, w1, b1, w2, b2, dilation=d)
    tol = 3e-3 + 2.0 ** -8 * ref.double().abs().clamp(min=1.0)
    for name, v in (("stage", y), ("units", acc)):
        d = (v.double() - ref.double()).abs()
        bad = d > 2 * tol
        pos = bad.any(dim=1)[0].nonzero().flatten()
        print(rep, name, "max", float(d.max()), "bad", int(bad.sum()), "positions", (int(pos.min()), int(pos.max())) if len(pos) else None, flush=True)
