"""Build the REAL reference monotonic_align kernel (the reference's only native code,
monotonic_align/core.pyx) from where it lies under /root/reference into oracle/_ref/
with Cython + gcc (no reference build system is run; setup.py there only calls cythonize).
Used to pin oracle/maximum_path.c.  oracle/_ref/ is git-ignored but travels to the GPU box."""
import os
import shutil
import subprocess
import sys
import sysconfig
from pathlib import Path

REF = Path(os.environ.get("MOCKINGBIRD_REF", "/root/reference"))
OUT = Path(__file__).resolve().parent / "_ref"


def build(force=False):
    src = REF / "monotonic_align" / "core.pyx"
    if not src.exists():
        raise FileNotFoundError(f"{src} not present (only the build container has the reference)")
    OUT.mkdir(exist_ok=True)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so = OUT / f"ref_core{ext}"
    if so.exists() and not force and so.stat().st_mtime >= src.stat().st_mtime:
        return so
    c_file = OUT / "ref_core.c"
    # cythonize a scratch copy under a different module name; sources are not kept in the repo
    pyx = OUT / "ref_core.pyx"
    shutil.copyfile(src, pyx)
    subprocess.check_call([sys.executable, "-m", "cython", "-3", str(pyx), "-o", str(c_file)])
    inc = sysconfig.get_paths()["include"]
    import numpy as np
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", f"-I{inc}", f"-I{np.get_include()}",
                           str(c_file), "-o", str(so)])
    pyx.unlink()
    c_file.unlink()
    return so


def load():
    import importlib.util
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so = OUT / f"ref_core{ext}"
    if not so.exists():
        raise FileNotFoundError(so)
    spec = importlib.util.spec_from_file_location("ref_core", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
