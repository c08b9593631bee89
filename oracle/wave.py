"""Oracle (TEST INFRASTRUCTURE, never imported by the product path): the reference's numpy / libsndfile
waveform tail, restated.  Arrays keep the dtype they are given (float32 from the GAN vocoders, float64 from
WaveRNN.generate); NumPy >= 2 scalar promotion, as installed here and used to generate tests/golden/wave.npz.

  peak_normalize   gen_voice.py:41, control/toolbox/__init__.py:313
  encode_16bits    models/vocoder/wavernn/audio.py:38-39
  save_wav_pcm     models/synthesizer/audio.py:12-15 (the int16 array handed to scipy.io.wavfile.write)
  sndfile_pcm16    libsndfile src/pcm.c f2s_clip_array / d2s_clip_array, the conversion behind
                   sf.write(path, y, sr, "PCM_16") (run.py:91) with python-soundfile's SFC_SET_CLIPPING on.
                   libsndfile / SoundFile (requirements.txt: "SoundFile", unpinned) are neither vendored by the
                   reference nor installed here: PARITY UNPINNED for this one function -- restated from the
                   library's published source, checked only against hand-computed cases."""
import numpy as np


def peak_normalize(wav, target=0.97):
    return wav / np.abs(wav).max() * target


def encode_16bits(x):
    return np.clip(x * 2 ** 15, -2 ** 15, 2 ** 15 - 1).astype(np.int16)


def save_wav_pcm(wav):
    wav = wav.copy()
    wav *= 32767 / max(0.01, np.max(np.abs(wav)))
    return wav.astype(np.int16)


def sndfile_pcm16(x):
    """scaled = x * 0x8000 in the sample type; >= 0x7FFF -> 0x7FFF, <= -0x8000 -> -0x8000, else lrint (half-even)."""
    scaled = x * x.dtype.type(32768.0)
    out = np.rint(np.clip(scaled, -32768.0, 32767.0))  # np.rint rounds half to even, like lrint in the default mode
    out = np.where(scaled >= 32767.0, 32767.0, np.where(scaled <= -32768.0, -32768.0, out))
    return out.astype(np.int16)
