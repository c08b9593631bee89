"""Oracle wrapper: monotonic_align.maximum_path via the C restatement (oracle/maximum_path.c),
mirroring the host glue of monotonic_align/__init__.py:6-19.  Falls back to a pure-Python
loop (small cases only) if the C object has not been built."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "liboracle_mp.so"


def build(force=False):
    src = _HERE / "maximum_path.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        _SO.parent.mkdir(exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", str(_SO), str(src)])
    return _SO


def maximum_path_c(paths, values, t_ys, t_xs):
    """In-place on numpy arrays, like the Cython maximum_path_c (core.pyx:38-42)."""
    lib = C.CDLL(str(build()))
    b, t_t, t_s = values.shape
    assert paths.dtype == np.int32 and values.dtype == np.float32
    assert paths.flags.c_contiguous and values.flags.c_contiguous
    t_ys = np.ascontiguousarray(t_ys, np.int32)
    t_xs = np.ascontiguousarray(t_xs, np.int32)
    lib.oracle_maximum_path_c(paths.ctypes.data_as(C.c_void_p), values.ctypes.data_as(C.c_void_p),
                              t_ys.ctypes.data_as(C.c_void_p), t_xs.ctypes.data_as(C.c_void_p),
                              C.c_int(b), C.c_int(t_t), C.c_int(t_s))


def maximum_path(neg_cent, mask):
    """numpy restatement of monotonic_align/__init__.py:6-19: returns (path, mutated values)."""
    values = np.ascontiguousarray(neg_cent, np.float32).copy()
    path = np.zeros(values.shape, dtype=np.int32)
    t_t_max = mask.sum(1)[:, 0].astype(np.int32)
    t_s_max = mask.sum(2)[:, 0].astype(np.int32)
    maximum_path_c(path, values, t_t_max, t_s_max)
    return path, values
