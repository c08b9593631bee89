"""The resident (one-launch) loops under uneven load: tools/resident_stress.py in its own process (a fallback changes a per-process
default) -- every run equals the quiet resident run or, after a lost hand-off, the launch chain's result."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_resident_loops_beside_a_gemm_stream(cuda, lib):
    env = {k: v for k, v in os.environ.items() if not k.startswith("MBHIP_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resident_stress.py"), "6"], capture_output=True, text=True, timeout=600,
                       env=env, stdin=subprocess.DEVNULL)
    tail = "\n".join(r.stdout.strip().splitlines()[-30:])
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, tail, r.stderr[-2000:])
    assert "WRONG" not in r.stdout
