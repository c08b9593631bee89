"""Shader-clock marks of workgroup 0 of mb_resblock_pair_split (diagnostics build: MODULE=resblock_pair_split tools/build_variant.sh
sptrace -DSPAIR_TRACE_BUILD; run with MBHIP_LIB=build_variants/libmbhip_sptrace.so):  python tools/spair_trace.py C T k d [B]
Prints, for tiles 1..3 of workgroup 0, what the MMA wave 0 and the support wave 4 spent per interval (cycles): MFMA chunks, waits at the
barriers, epilogues."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
Cc, T, k, d = [int(v) for v in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
path = tempfile.mktemp(suffix=".trace")
os.environ["MBHIP_DIAG"] = (os.environ.get("MBHIP_DIAG", "") + "," if os.environ.get("MBHIP_DIAG") else "") + "spair_trace=" + path
import ctypes as C
import torch, hiputil
from mockingbird_amd import _lib
L = _lib.lib()
x = torch.randn(B, Cc, T)
xt = hiputil.f32_cm_to_tm(x); yt = torch.empty_like(xt)
w1 = torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5; w2 = torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5
b1 = torch.zeros(Cc).cuda(); b2 = torch.zeros(Cc).cuda()
pw, us1, us2 = hiputil.pack_pair_split(w1, w2)
a = _lib.ResPairSplitArgs()
a.d_x, a.d_y, a.d_wpacked, a.d_b1, a.d_b2 = xt.data_ptr(), yt.data_ptr(), pw.data_ptr(), b1.data_ptr(), b2.data_ptr()
a.batch, a.channels, a.t, a.ksize, a.dilation = B, Cc, T, k, d
a.slope, a.out_scale, a.unscale1, a.unscale2 = 0.1, 1.0, us1, us2
for _ in range(3):
    _lib.check(L.mb_resblock_pair_split(C.byref(a), None), "pair")
torch.cuda.synchronize()
line = open(path).read().strip().split("\n")[-1]
head, marks = line.split(":")
geo = [int(v) for v in head.split()]
m = [int(v) for v in marks.split()]
NCH = Cc // (32 if Cc >= 32 else 16)
print("C MT WN NTW YS ntaps dil nbuf tiles =", geo)
for it in range(4):
    mm = m[(0 * 4 + it) * 64:(0 * 4 + it) * 64 + 64]
    ss = m[(1 * 4 + it) * 64:(1 * 4 + it) * 64 + 64]
    if not mm[1]:
        continue
    bw = sum(mm[2 * c + 1] - mm[2 * c] for c in range(NCH))           # MMA wave waiting at the B barriers
    ch = [(mm[2 * c + 2] if c + 1 < NCH else mm[32]) - mm[2 * c + 1] for c in range(NCH)]  # chunk bodies
    print(f"tile {it}: MMA total {mm[39] - mm[0]}: B waits {bw}, conv1 chunks {sum(ch)} (each ~{sum(ch) // NCH}), W wait {mm[33] - mm[32]}, epi1 {mm[34] - mm[33]}, "
          f"E1 wait {mm[35] - mm[34]}, conv2 {mm[36] - mm[35]}, P wait {mm[37] - mm[36]}, epi2 {mm[38] - mm[37]}, Y wait {mm[39] - mm[38]}")
    if mm[63] > mm[62]:
        print(f"        wall {(mm[63] - mm[62]) / 100.0:.1f} us -> shader clock {(mm[39] - mm[0]) / ((mm[63] - mm[62]) / 100.0) / 1e3:.2f} GHz")
    sbw = sum(ss[2 * c + 1] - ss[2 * c] for c in range(NCH))
    swork = sum((ss[2 * c + 2] if c + 1 < NCH else ss[32]) - ss[2 * c + 1] for c in range(NCH))
    print(f"        support total {ss[36] - ss[0]}: B waits {sbw}, work between Bs {swork}, W wait {ss[33] - ss[32]}, E1 {ss[34] - ss[33]}, P {ss[35] - ss[34]}, Y {ss[36] - ss[35]}")
os.unlink(path)
