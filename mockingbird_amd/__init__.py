"""mockingbird_amd -- MI355X (gfx950) native hot path of babysor/MockingBird.

Mel synthesis (Tacotron decoder loop + CBHG postnet) and waveform vocoding (WaveRNN sample
loop, HiFi-GAN / Fre-GAN generators) plus monotonic_align, as hand-written HIP kernels behind a
C ABI (include/mbhip.h -> libmbhip.so) and the reference's own Python facades.  See DESIGN.md
and INTEGRATION.md."""
import importlib
import sys

_ALIASES = {
    "models.synthesizer.inference": "mockingbird_amd.synthesizer.inference",
    "models.vocoder.hifigan.inference": "mockingbird_amd.vocoder.hifigan.inference",
    "models.vocoder.fregan.inference": "mockingbird_amd.vocoder.fregan.inference",
    "models.vocoder.wavernn.inference": "mockingbird_amd.vocoder.wavernn.inference",
    "monotonic_align": "mockingbird_amd.monotonic_align",
}


def install():
    """Alias the reference's facade module names to the HIP-backed ones so gen_voice.py /
    control/mkgui/app.py / control/toolbox run unchanged."""
    for ref_name, mine in _ALIASES.items():
        sys.modules[ref_name] = importlib.import_module(mine)
    return sorted(_ALIASES)
