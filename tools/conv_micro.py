"""One conv shape through mb_conv1d, timed: python tools/conv_micro.py B Cin Cout T k dil [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C
import torch, hiputil
from mockingbird_amd import _lib
B, Cin, Cout, T, k, dil = [int(x) for x in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
L = _lib.lib()
w = torch.randn(Cout, Cin, k) / (Cin * k) ** 0.5
packed, _ = hiputil.pack_conv(w)
pw = packed.cuda(); x = torch.randn(B, Cin, T).cuda(); y = torch.empty(B, Cout, T).cuda()
a = _lib.ConvArgs()
a.d_x, a.d_wpacked, a.d_y = x.data_ptr(), pw.data_ptr(), y.data_ptr()
a.x_bstride, a.y_bstride, a.res_bstride = Cin * T, Cout * T, Cout * T
a.batch, a.c_in, a.c_out, a.t_in, a.t_out = B, Cin, Cout, T, T
a.ksize, a.dilation, a.pad, a.up = k, dil, (k - 1) * dil // 2, 1
a.in_act, a.in_slope, a.in_scale, a.out_scale, a.in_repeat = (2 if os.environ.get("POOL") == "1" else 1), 0.1, 1.0, 1.0, 1  # POOL=1: MaxPool1d(2, 1, 1)[:T] fused into the input (the CBHG projections)
for dbg in os.environ.get("DBGS", "0").split(","):
    os.environ["MBHIP_CONV_SPLIT_DBG"] = dbg
    for _ in range(3):
        L.mb_conv1d(C.byref(a), None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.mb_conv1d(C.byref(a), None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * Cout * Cin * k * T * B
    by = 4.0 * (Cin + Cout) * T * B
    print(f"dbg={dbg}: {us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s algorithmic  {by / us / 1e3:.0f} GB/s (read x + write y)", flush=True)
