"""Drop-in for models/vocoder/hifigan/inference.py (reference lines 9-11, 22-74):
same module-level load_model / is_loaded / infer_waveform and globals, the
generator forward runs as HIP kernels through libmbhip.so (mb_gan_forward)."""
from ..gan import GanFacade

_facade = GanFacade(0, "./vocoder/hifigan/config_16k_.json", "hifigan")

generator = None
output_sample_rate = None
_device = None


def load_model(weights_fpath, config_fpath=None, verbose=True, dtype=None):
    """Reference signature plus an additive keyword: dtype "f32" (parity path) | "f16"
    (fp16 matrix cores); default from env MBHIP_GAN_DTYPE, else "f32"."""
    global generator, output_sample_rate, _device
    _facade.load_model(weights_fpath, config_fpath, verbose, dtype=dtype)
    generator, output_sample_rate, _device = _facade.generator, _facade.output_sample_rate, _facade._device


def is_loaded():
    return _facade.is_loaded()


def infer_waveform(mel, progress_callback=None):
    return _facade.infer_waveform(mel, progress_callback)


accepts_device_mels = True  # (pipeline.gen_wavs: mels may be device tensors)


def infer_waveform_batch(mels, progress_callback=None, **kw):
    """Additive API: see GanFacade.infer_waveform_batch (normalize / pcm16 / breaks / device_out keywords)."""
    return _facade.infer_waveform_batch(mels, progress_callback, **kw)
