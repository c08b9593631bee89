"""mockingbird_amd -- MI355X (gfx950) native hot path of babysor/MockingBird.

Mel synthesis (Tacotron decoder loop + CBHG postnet) and waveform vocoding (WaveRNN sample
loop, HiFi-GAN / Fre-GAN generators) plus monotonic_align, as hand-written HIP kernels behind a
C ABI (include/mbhip.h -> libmbhip.so) and the reference's own Python facades.  See DESIGN.md
and INTEGRATION.md."""
import importlib
import importlib.util
import sys

_ALIASES = {
    "models.synthesizer.inference": "mockingbird_amd.synthesizer.inference",
    "models.vocoder.hifigan.inference": "mockingbird_amd.vocoder.hifigan.inference",
    "models.vocoder.fregan.inference": "mockingbird_amd.vocoder.fregan.inference",
    "models.vocoder.wavernn.inference": "mockingbird_amd.vocoder.wavernn.inference",
    "monotonic_align": "mockingbird_amd.monotonic_align",
}


def install(ppg2mel=False):
    """Alias the reference's facade module names to the HIP-backed ones so gen_voice.py /
    control/mkgui/app.py / control/toolbox run unchanged.
    ppg2mel=True also aliases the voice-conversion model package `models.ppg2mel` (run.py:13,45,77;
    control/mkgui/app_vc.py:12,144): `MelDecoderMOLv2` / `load_model` become the HIP ones, the package's submodules
    (utils, train, preprocess: reference CPU code) stay importable through the reference's own directory."""
    names = dict(_ALIASES)
    for ref_name, mine in names.items():
        sys.modules[ref_name] = importlib.import_module(mine)
    if ppg2mel:
        mod = importlib.import_module("mockingbird_amd.ppg2mel")
        try:  # keep `import models.ppg2mel.utils...` working: submodules resolve in the reference's package directory
            import os
            spec = importlib.util.find_spec("models")
            for base in (spec.submodule_search_locations if spec else []):
                d = os.path.join(base, "ppg2mel")
                if os.path.isdir(d) and d not in mod.__path__:
                    mod.__path__.append(d)
        except Exception:
            pass
        sys.modules["models.ppg2mel"] = mod
        names["models.ppg2mel"] = "mockingbird_amd.ppg2mel"
    return sorted(names)
