"""The C-ABI library builds, loads, and exports every symbol include/mbhip.h declares
(no compute calls -- runs without a GPU)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (ROOT / "include" / "mbhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mb_[a-z0-9_]+)\s*\(", text)))


def test_header_has_no_torch_types():
    text = (ROOT / "include" / "mbhip.h").read_text()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S).lower()
    assert 'extern "C"' in text


def test_every_declared_symbol_is_exported_and_bound(lib):
    from mockingbird_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 25
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], text=True)
    exported = set(re.findall(r" T (mb_[a-z0-9_]+)", out))
    missing = [s for s in syms if s not in exported]
    assert not missing, f"declared in mbhip.h but not exported: {missing}"
    unbound = [s for s in syms if s not in _lib.SIGNATURES]
    assert not unbound, f"declared in mbhip.h but not bound in _lib.SIGNATURES: {unbound}"
    extra = [s for s in _lib.SIGNATURES if s not in syms]
    assert not extra, f"bound but not declared in the header: {extra}"
    assert lib.mb_abi_version() == 4


def test_struct_layouts_match_header_sizes(lib, tmp_path):
    """ctypes mirrors of the ABI structs have the same size as the C structs."""
    from mockingbird_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include "mbhip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(mb_conv1d_args),sizeof(mb_gan_config),sizeof(mb_wavernn_config),'
                   'sizeof(mb_wavernn_plan),sizeof(mb_taco_config),sizeof(mb_conv1d_f16_args),'
                   'sizeof(mb_resblock_pair_f16_args),sizeof(mb_ppg2mel_config));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)])
    sizes = list(map(int, subprocess.check_output([str(exe)], text=True).split()))
    mine = [C.sizeof(_lib.ConvArgs), C.sizeof(_lib.GanConfig), C.sizeof(_lib.WaveRNNConfig),
            C.sizeof(_lib.WaveRNNPlan), C.sizeof(_lib.TacoConfig), C.sizeof(_lib.ConvF16Args),
            C.sizeof(_lib.ResPairF16Args), C.sizeof(_lib.Ppg2MelConfig)]
    assert sizes == mine


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mockingbird_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.MbHipError, match="no CPU fallback"):
        _lib.lib()


def test_product_never_imports_the_oracle():
    for p in (ROOT / "mockingbird_amd").rglob("*.py"):
        txt = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), p


def test_cpu_tensor_without_gpu_fails_loudly(lib):
    """No CPU path: with no GPU visible the facades refuse instead of silently computing on the host
    (with a GPU, host tensors are copied in and out -- tests/test_maximum_path_gpu.py)."""
    import torch
    from mockingbird_amd import _lib
    from mockingbird_amd.monotonic_align import maximum_path
    if torch.cuda.is_available():
        pytest.skip("GPU visible: host tensors are accepted and computed on the device")
    with pytest.raises(_lib.MbHipError, match="no CPU path"):
        maximum_path(torch.zeros(1, 4, 3), torch.ones(1, 4, 3))
