"""Shader-clock marks of workgroup 0 of mb_conv_split_tm (diagnostics build: MODULE=conv_split_tm tools/build_variant.sh ctmtrace -DCTM_TRACE_BUILD;
run with MBHIP_LIB=build_variants/libmbhip_ctmtrace.so):  python tools/ctm_trace.py C_in M T k [B] [split]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
Cin, M, T, k = [int(v) for v in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
split = len(sys.argv) > 6 and sys.argv[6] == "split"
path = tempfile.mktemp(suffix=".trace")
os.environ["MBHIP_DIAG"] = (os.environ.get("MBHIP_DIAG", "") + "," if os.environ.get("MBHIP_DIAG") else "") + "ctm_trace=" + path
import torch, hiputil
x = torch.randn(B, Cin, T); w = torch.randn(M, Cin, k) / (Cin * k) ** 0.5
for _ in range(2):
    hiputil.conv_split_tm_hip(x, w, None, pad=(k - 1) // 2, x_split=split)
line = open(path).read().strip().split("\n")[-1]
head, marks = line.split(":")
print("CK MT WN NTW c_in c_out ntaps NCH nbuf ntiles =", head.split())
m = [int(v) for v in marks.split()]
NCH = int(head.split()[7])
for it in range(4):
    mm = m[it * 64:it * 64 + 64]; ss = m[(4 + it) * 64:(4 + it) * 64 + 64]
    if not mm[1]:
        continue
    n = min(NCH, 8) // 2
    bw = sum(mm[4 * c + 1] - mm[4 * c] for c in range(n)) + sum(mm[4 * c + 3] - mm[4 * c + 2] for c in range(n))
    ch = sum(mm[4 * c + 2] - mm[4 * c + 1] for c in range(n))
    print(f"tile {it}: MMA first {2 * n} chunks {mm[4 * (n - 1) + 3] - mm[0]}: barrier waits {bw}, even-chunk bodies {ch} (~{ch // n} each); tile total {mm[63] - mm[0]}, YF wait {mm[61] - mm[60]}, epilogue {mm[62] - mm[61]}")
    sw = sum(ss[4 * c + 1] - ss[4 * c] for c in range(n)); fill = sum(ss[4 * c + 2] - ss[4 * c + 1] for c in range(n)); wr = sum(ss[4 * c + 3] - ss[4 * c + 2] for c in range(n))
    print(f"        support even intervals: barrier waits {sw}, issue+commit {fill} (~{fill // n} each), write part {wr} (~{wr // n} each)")
os.unlink(path)
