"""mb_conv_split_tm at given shapes, us per launch:  python tools/ctm_bench.py [reps] c_in,M,T,k[,split] ...   (batch 32; SPAIR_B overrides)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C
import torch, hiputil
from mockingbird_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(os.environ.get("SPAIR_B", "32"))
L = _lib.lib()
for spec in sys.argv[2:]:
    f = spec.split(",")
    Cin, M, T, k = [int(v) for v in f[:4]]
    split = len(f) > 4 and f[4] == "split"
    w = torch.randn(M, Cin, k) / (Cin * k) ** 0.5
    packed = torch.empty(L.mb_conv_split_tm_packed_halves(M, Cin, k), dtype=torch.float16)
    us = torch.zeros(1, dtype=torch.float32)
    _lib.check(L.mb_conv_split_tm_pack(w.data_ptr(), M, Cin, k, packed.data_ptr(), us.data_ptr()), "pack")
    pw = packed.cuda()
    xt = torch.randn(B, T, Cin, device="cuda")
    if split:
        xt = hiputil.split_tensor(xt)
    y = torch.empty(B, T, M, device="cuda")
    a = _lib.ConvSplitTmArgs()
    a.d_x, a.d_y, a.d_wpacked = xt.data_ptr(), y.data_ptr(), pw.data_ptr()
    a.batch, a.t, a.c_in, a.c_out, a.ksize, a.dilation, a.pad = B, T, Cin, M, k, 1, (k - 1) // 2
    a.in_slope, a.unscale, a.out_scale, a.out_act = (1.0 if split else 0.1), float(us[0]), 1.0, 0
    a.x_split = int(split)
    fn = lambda: _lib.check(L.mb_conv_split_tm(C.byref(a), _lib.stream_ptr()), "conv")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e3
    gf = 2.0 * B * T * Cin * M * k
    print(f"conv_split_tm {spec:28s} {t:8.1f} us  {gf / t / 1e6:7.1f} TFLOP/s algorithmic  {(B * T * (Cin + M) * 4) / t / 1e6:6.2f} TB/s in+out  (MBHIP_LIB={os.environ.get('MBHIP_LIB', '')})")
