"""Build libmbhip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container
as well as on the MI355X box.  Objects are cached by source mtime.
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIB = ROOT / "libmbhip.so"
ARCH = "gfx950"
SOURCES = ["common.hip", "conv1d.hip", "conv1d_f16.hip", "resblock_f16.hip", "resblock_stage_f16.hip", "resblock_stage_f32.hip", "resblock_pair_split.hip", "conv_split_tm.hip", "gan.hip", "rnn.hip", "wavernn.hip", "wavernn_post.hip", "wave_post.hip", "tacotron.hip", "ppg2mel.hip", "ppg_net.hip",
           "maximum_path.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src: Path, dst: Path, extra=()):
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in (src, *extra))


def build(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    headers = list(CSRC.glob("*.h")) + [ROOT.parent / "include" / "mbhip.h", Path(__file__)]
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    objs = []
    procs = []
    for src in srcs:
        obj = CSRC / (src.stem + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
            if verbose:
                print("[mbhip build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src.name} failed ---\n{out.decode(errors='replace')}\n")
        elif verbose and out.strip():
            print(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if force or procs or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print("[mbhip build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
