"""The fused split ResBlock unit (mb_resblock_pair_split, time-major fp32) against the two conv1d_split_kernel launches it replaces
(mb_conv1d, channel-major fp32), per stage shape of HiFi-GAN 32 x 200:  python tools/spair_bench.py [reps] [C,T ...]
Prints us per launch, algorithmic TFLOP/s (2 convs, one product per MAC) and the fraction of the 833 TFLOP/s ceiling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C
import json
import torch, hiputil
from mockingbird_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[2:]] or [(256, 1000), (128, 5000), (64, 20000), (32, 40000)]
B = int(os.environ.get("SPAIR_B", "32"))
L = _lib.lib()


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out = []
for (Cc, T) in shapes:
    x = torch.randn(B, Cc, T)
    xt = hiputil.f32_cm_to_tm(x)
    yt = torch.empty_like(xt)
    xc = x.cuda(); hc = torch.empty_like(xc); yc = torch.empty_like(xc)
    for k in [int(v) for v in os.environ.get("SPAIR_KS", "3,7,11").split(",")]:
        for d in [int(v) for v in os.environ.get("SPAIR_DS", "1,3,5").split(",")]:
            w1 = torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5
            w2 = torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5
            b1 = torch.zeros(Cc).cuda(); b2 = torch.zeros(Cc).cuda()
            pw, us1, us2 = hiputil.pack_pair_split(w1, w2)
            a = _lib.ResPairSplitArgs()
            a.d_x, a.d_y, a.d_wpacked, a.d_b1, a.d_b2 = xt.data_ptr(), yt.data_ptr(), pw.data_ptr(), b1.data_ptr(), b2.data_ptr()
            a.batch, a.channels, a.t, a.ksize, a.dilation = B, Cc, T, k, d
            a.slope, a.out_scale, a.unscale1, a.unscale2 = 0.1, 1.0, us1, us2
            t_pair = timeit(lambda: _lib.check(L.mb_resblock_pair_split(C.byref(a), None), "pair"))
            # legacy: two per-conv launches on channel-major tensors (conv1 lrelu-in; conv2 lrelu-in + residual)
            p1 = hiputil.pack_conv(w1)[0].cuda(); p2 = hiputil.pack_conv(w2)[0].cuda()
            def conv(pw_, src, dst, dil, res):
                c = _lib.ConvArgs()
                c.d_x, c.d_wpacked, c.d_y = src.data_ptr(), pw_.data_ptr(), dst.data_ptr()
                c.d_res = res.data_ptr() if res is not None else None
                c.x_bstride = c.y_bstride = c.res_bstride = Cc * T
                c.batch, c.c_in, c.c_out, c.t_in, c.t_out = B, Cc, Cc, T, T
                c.ksize, c.dilation, c.pad, c.up = k, dil, (k - 1) * dil // 2, 1
                c.in_act, c.in_slope, c.in_scale, c.out_scale, c.in_repeat = 1, 0.1, 1.0, 1.0, 1
                return c
            c1 = conv(p1, xc, hc, d, None); c2 = conv(p2, hc, yc, 1, xc)
            def legacy():
                L.mb_conv1d(C.byref(c1), None); L.mb_conv1d(C.byref(c2), None)
            t_leg = timeit(legacy) if os.environ.get("SPAIR_NOLEG") != "1" else 0.0
            fl = 2 * 2.0 * Cc * Cc * k * T * B
            rec = {"C": Cc, "T": T, "k": k, "d": d, "pair_us": round(t_pair, 1), "legacy_us": round(t_leg, 1),
                   "pair_tflops": round(fl / t_pair / 1e6, 1), "frac_833": round(fl / t_pair / 1e6 / 833.0, 3)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
tot_p = sum(r["pair_us"] for r in out); tot_l = sum(r["legacy_us"] for r in out)
print(json.dumps({"sum_pair_us": round(tot_p, 1), "sum_legacy_us": round(tot_l, 1)}))
