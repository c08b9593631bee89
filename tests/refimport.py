"""Import the real reference (/root/reference) with the shims SURVEY.md section 8c lists.
Only usable in the build container; the GPU box has no /root/reference."""
import os
import sys
import types

REF = os.environ.get("MOCKINGBIRD_REF", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models", "vocoder"))


def setup():
    if not available():
        raise RuntimeError("reference checkout not present")
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import numpy as np
    if not hasattr(np, "cumproduct"):
        np.cumproduct = np.cumprod  # fatchord_version.py:64 (NumPy 2 removed it)
    for name in ("librosa", "librosa.filters", "soundfile"):  # wavernn/audio.py:3,6 (unused by generate)
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


def import_ppg2mel_decoder():
    """models/ppg2mel/rnn_decoder_mol.py without running the package __init__ (which imports the whole
    voice-conversion stack: espnet-style encoders, preprocessing, ...)."""
    import importlib
    setup()
    import models  # noqa: F401
    if "models.ppg2mel" not in sys.modules:
        pkg = types.ModuleType("models.ppg2mel")
        pkg.__path__ = [os.path.join(REF, "models", "ppg2mel")]
        sys.modules["models.ppg2mel"] = pkg
    return importlib.import_module("models.ppg2mel.rnn_decoder_mol")


def import_ppg2mel_package():
    """The real models/ppg2mel package (__init__.py: MelDecoderMOLv2 with its conv front end and CNN postnet)."""
    import importlib
    setup()
    sys.modules.pop("models.ppg2mel", None)  # drop the stub package import_ppg2mel_decoder() may have registered
    for k in [k for k in sys.modules if k.startswith("models.ppg2mel.")]:
        del sys.modules[k]
    return importlib.import_module("models.ppg2mel")
