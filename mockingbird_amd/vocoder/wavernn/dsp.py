"""Host-side float64 post-processing of WaveRNN.generate (the reference does this
in numpy as well): xfade_and_unfold (fatchord_version.py:340-402), decode_mu_law /
de_emphasis (wavernn/audio.py:92-107), truncate + fade-out (:250-253)."""
import numpy as np
from scipy.signal import lfilter


def xfade_and_unfold(y: np.ndarray, overlap: int) -> np.ndarray:
    num_folds, length = y.shape
    target = length - 2 * overlap
    total_len = num_folds * (target + overlap) + overlap
    silence_len = overlap // 2
    fade_len = overlap - silence_len
    t = np.linspace(-1, 1, fade_len, dtype=np.float64)
    fade_in = np.concatenate([np.zeros(silence_len, np.float64), np.sqrt(0.5 * (1 + t))])
    fade_out = np.concatenate([np.sqrt(0.5 * (1 - t)), np.zeros(silence_len, np.float64)])
    y[:, :overlap] *= fade_in
    y[:, -overlap:] *= fade_out
    unfolded = np.zeros(total_len, dtype=np.float64)
    step = target + overlap
    for i in range(num_folds):
        unfolded[i * step:i * step + length] += y[i]
    return unfolded


def decode_mu_law(y: np.ndarray, mu: int) -> np.ndarray:
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def de_emphasis(x: np.ndarray, coef: float) -> np.ndarray:
    return lfilter([1], [1, -coef], x)


def finish(samples: np.ndarray, batched: bool, overlap: int, n_classes: int, mu_law: bool,
           apply_preemphasis: bool, preemphasis: float, wave_len: int, hop_length: int) -> np.ndarray:
    out = samples.astype(np.float64)
    out = xfade_and_unfold(out, overlap) if batched else out[0]
    if mu_law:
        out = decode_mu_law(out, n_classes)
    if apply_preemphasis:
        out = de_emphasis(out, preemphasis)
    fade_out = np.linspace(1, 0, 20 * hop_length)
    out = out[:wave_len]
    out[-20 * hop_length:] *= fade_out  # raises like the reference for mels < 26 frames (SURVEY finding 5)
    return out
