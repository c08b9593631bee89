"""Drop-in for models/vocoder/wavernn/inference.py:8-64: same ``load_model /
is_loaded / infer_waveform`` and globals; ``WaveRNN.generate``
(fatchord_version.py:153-257) runs on the MI355X through libmbhip.so."""
import ctypes as C
import time

import numpy as np
import torch

from ... import _lib, weights
from . import hparams as hp

_model = None   # type: WaveRNNDevice
_device = None


class WaveRNNDevice:
    """Owns an ``mb_wavernn`` handle (weights resident in HBM)."""

    def __init__(self, state_dict, hparams=hp):
        self.hp = hparams
        self.cfg = weights.wavernn_config(hparams)
        L = _lib.lib()
        ws = weights.wavernn_weight_list(state_dict, self.cfg)
        n = L.mb_wavernn_num_weights(C.byref(self.cfg))
        if n != len(ws):
            raise _lib.MbHipError(f"weight list has {len(ws)} tensors, ABI expects {n}")
        for i, w in enumerate(ws):
            want = L.mb_wavernn_weight_numel(C.byref(self.cfg), i)
            if w.numel() != want:
                raise _lib.MbHipError(f"weight {i}: {tuple(w.shape)} has {w.numel()} elements, expected {want}")
        h = C.c_void_p()
        _lib.check(L.mb_wavernn_create(C.byref(self.cfg), _lib.host_ptr_array(ws), len(ws), C.byref(h)),
                   "mb_wavernn_create")
        self._h = h
        self.n_classes = 30 if self.cfg.mode == 1 else 2 ** self.cfg.bits  # fatchord_version.py:95-98
        self.noise_width = 11 if self.cfg.mode == 1 else self.n_classes   # MOL: 10 mixture-indicator uniforms + 1 logistic uniform
        self.hop_length = hparams.hop_length
        self.sample_rate = hparams.sample_rate
        self._ws = None
        self._progress = torch.zeros(4, dtype=torch.int32).pin_memory() if torch.cuda.is_available() else None
        self.last_loop_ms = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.mb_wavernn_destroy(h)
            self._h = None

    def plan(self, frames, batched, target, overlap):
        p = _lib.WaveRNNPlan()
        _lib.check(_lib.lib().mb_wavernn_plan_generate(self._h, frames, int(bool(batched)), target, overlap,
                                                       C.byref(p)), "mb_wavernn_plan_generate")
        return p

    def generate_samples(self, mel: torch.Tensor, batched, target, overlap, noise=None, seed=None,
                         forced=None, want_logits=False, progress_callback=None):
        """mel [80, F] float32 CUDA (already normalised) -> samples [n_folds, seq_len] CUDA float32
        (the tensor stacked at fatchord_version.py:236), optionally the fc3 logits [S, N, C]."""
        if not mel.is_cuda:
            raise _lib.MbHipError("WaveRNN needs a CUDA(HIP) tensor; there is no CPU path")
        if seed is None:  # the reference samples from torch's global generator (fatchord_version.py:224-226)
            seed = _lib.fresh_seed()
        mel = mel.to(torch.float32).contiguous()
        F = mel.shape[1]
        p = self.plan(F, batched, target, overlap)
        dev = mel.device
        if self._ws is None or self._ws.numel() < p.workspace_bytes or self._ws.device != dev:
            self._ws = torch.empty(p.workspace_bytes, dtype=torch.uint8, device=dev)
        samples = torch.empty(p.n_folds, p.seq_len, dtype=torch.float32, device=dev)
        logits = (torch.empty(p.seq_len, p.n_folds, self.n_classes, dtype=torch.float32, device=dev)
                  if want_logits else None)
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
            if tuple(noise.shape) != (p.seq_len, p.n_folds, self.noise_width):
                raise _lib.MbHipError(f"noise must be {(p.seq_len, p.n_folds, self.noise_width)}, got {tuple(noise.shape)}")
        if forced is not None:
            forced = forced.to(dev, torch.float32).contiguous()
            if tuple(forced.shape) != (p.n_folds, p.seq_len):
                raise _lib.MbHipError(f"forced must be {(p.n_folds, p.seq_len)}, got {tuple(forced.shape)}")
        self._progress.zero_()
        L = _lib.lib()
        start = time.time()
        args = (self._h, C.byref(p), _lib.ptr(mel), _lib.ptr(noise), int(seed), _lib.ptr(samples), _lib.ptr(logits), _lib.ptr(forced),
                _lib.ptr(self._progress), _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr())
        if progress_callback is None:
            _lib.check(L.mb_wavernn_generate(*args), "mb_wavernn_generate")
        else:
            # the kernels publish the completed step count every 100 steps through the pinned word (fatchord_version.py:232-234).
            # The resident kernels make mb_wavernn_generate host-blocking, so the call runs on a worker thread (ctypes drops the
            # GIL) while this thread reports progress; the device / stream context is the caller's either way.
            import threading
            res, dev_index, stream = {}, mel.device.index, torch.cuda.current_stream()

            def run():
                torch.cuda.set_device(dev_index)
                with torch.cuda.stream(stream):
                    res["rc"] = L.mb_wavernn_generate(*args)
                    msg = L.mb_last_error() if res["rc"] else None  # (the error text is per thread: read it here)
                    res["err"] = msg.decode() if msg else ""

            th = threading.Thread(target=run)
            th.start()
            last, done_ev = -1, None
            while True:
                if not th.is_alive() and done_ev is None:
                    done_ev = torch.cuda.Event()
                    done_ev.record()
                if done_ev is not None and done_ev.query():
                    break
                i = int(self._progress[0]) - 1
                if i >= 0 and i != last:
                    last = i
                    gen_rate = (i + 1) / max(time.time() - start, 1e-9) * p.n_folds / 1000
                    progress_callback(i, p.seq_len, p.n_folds, gen_rate)
                time.sleep(0.002)
            th.join()
            if res.get("rc", -1) < 0:
                raise _lib.MbHipError(f"mb_wavernn_generate failed ({res.get('rc')}): {res.get('err')}")
        torch.cuda.current_stream().synchronize()
        ms, nl = C.c_float(), C.c_int()
        _lib.check(L.mb_wavernn_last_loop_ms(self._h, C.byref(ms), C.byref(nl)), "mb_wavernn_last_loop_ms")
        self.last_loop_ms, self.last_loop_launches, self.last_plan = ms.value, nl.value, p
        path, fb = C.c_int(), C.c_int()
        _lib.check(L.mb_wavernn_last_path(self._h, C.byref(path), C.byref(fb)), "mb_wavernn_last_path")
        # which form of the loop produced these samples ("chain" | "persist1" | "pipe" | "pipe16") and whether a resident launch was
        # discarded on the way (None | "abort" | "range"): the chain's stream differs from pipe16's at near-ties
        self.last_path = ("chain", "persist1", "pipe", "pipe16")[path.value]
        self.last_fallback = (None, "abort", "range")[fb.value]
        return (samples, logits) if want_logits else samples

    def sampler_noise(self, seed, steps, folds, step0=0):
        """Test hook (mb_wavernn_debug_noise): the Exp(1) draws the on-device sampler of the production paths consumes
        for `seed` -> CUDA tensor [steps, folds, n_classes] (RAW: Exp(1) words; MOL: [steps, folds, 11] uniforms); handed to
        the oracle's sample loop as `noise`, it makes the oracle sample what generate_samples(seed=seed) samples."""
        out = torch.empty(steps, folds, self.noise_width, dtype=torch.float32, device="cuda")
        if self.cfg.mode != 0:  # MOL: the uniform draws (mixture indicators, then the logistic one)
            _lib.check(_lib.lib().mb_wavernn_debug_noise_mol(int(seed), int(step0), int(steps), int(folds), self.noise_width - 1,
                                                         _lib.ptr(out), _lib.stream_ptr()), "mb_wavernn_debug_noise_mol")
        else:
            _lib.check(_lib.lib().mb_wavernn_debug_noise(int(seed), int(step0), int(steps), int(folds), self.n_classes,
                                                         _lib.ptr(out), _lib.stream_ptr()), "mb_wavernn_debug_noise")
        torch.cuda.current_stream().synchronize()
        return out

    def generate_samples_batch(self, mels, target, overlap, seeds=None):
        """Several utterances in ONE sample loop (mb_wavernn_generate_batch; additive API).  mels: list of
        [80, F_u] CUDA tensors (already divided by mel_max_abs_value); seeds: one per utterance (default: drawn
        from torch's global generator).
        Returns the list of per-utterance sample tensors [folds_u, seq_len]; utterance u equals
        generate_samples(mels[u], True, target, overlap, seed=seeds[u])."""
        if not mels:
            return []
        if any(not m.is_cuda for m in mels):
            raise _lib.MbHipError("WaveRNN needs CUDA(HIP) tensors; there is no CPU path")
        mels = [m.to(torch.float32).contiguous() for m in mels]
        n = len(mels)
        seeds = [_lib.fresh_seed() for _ in range(n)] if seeds is None else [int(x) for x in seeds]
        if self.cfg.mode == 1:  # MOL: the shared loop is the RAW fused-sampler chain; utterances run one by one
            return [self.generate_samples(m, True, target, overlap, seed=sd) for m, sd in zip(mels, seeds)]
        dev = mels[0].device
        out = []
        for lo, hi in self.batch_groups([int(m.shape[1]) for m in mels], target, overlap, self._budget(dev)):
            out += self._run_group(mels[lo:hi], seeds[lo:hi], target, overlap, dev)
        return out

    workspace_budget_bytes = None  # cap on one sample loop's workspace; None = 85 % of the HBM free at call time

    def _budget(self, dev):
        if self.workspace_budget_bytes is not None:
            return int(self.workspace_budget_bytes)
        free, _ = torch.cuda.mem_get_info(dev)
        held = self._ws.numel() if self._ws is not None and self._ws.device == dev else 0
        return int(0.85 * (free + held))

    def _plan_batch(self, frames, target, overlap):
        n = len(frames)
        cfr = (C.c_int * n)(*frames)
        offs = (C.c_int * (n + 1))()
        plan = _lib.WaveRNNBatchPlan()
        _lib.check(_lib.lib().mb_wavernn_plan_generate_batch(self._h, n, cfr, int(target), int(overlap), C.byref(plan), offs),
                   "mb_wavernn_plan_generate_batch")
        return plan, cfr, offs

    def batch_groups(self, frames, target, overlap, budget):
        """The conditioning tables cost about 27 MB per 1000 mel frames (per-frame tables); a batch
        that still does not fit is cut into consecutive groups whose workspace fits `budget` bytes and the groups run one sample
        loop after the other (per-utterance seeds make the result independent of the grouping).  One utterance
        that does not fit by itself is an error, not an allocator failure."""
        groups, lo = [], 0
        while lo < len(frames):
            hi = lo + 1
            if self._plan_batch(frames[lo:hi], target, overlap)[0].workspace_bytes > budget:
                need = self._plan_batch(frames[lo:hi], target, overlap)[0].workspace_bytes
                raise _lib.MbHipError(f"WaveRNN: utterance {lo} ({frames[lo]} frames) needs a {need / 2**30:.1f} GiB "
                                      f"workspace, over the {budget / 2**30:.1f} GiB budget")
            # grow geometrically, then bisect: plan calls are host arithmetic
            step = 1
            while hi < len(frames):
                nxt = min(len(frames), hi + step)
                if self._plan_batch(frames[lo:nxt], target, overlap)[0].workspace_bytes > budget:
                    if step == 1:
                        break
                    step = max(1, step // 2)
                    continue
                hi, step = nxt, step * 2
            groups.append((lo, hi))
            lo = hi
        return groups

    def _run_group(self, mels, seeds, target, overlap, dev):
        n = len(mels)
        L = _lib.lib()
        plan, frames, offs = self._plan_batch([int(m.shape[1]) for m in mels], target, overlap)
        if self._ws is None or self._ws.numel() < plan.workspace_bytes or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
        samples = torch.empty(plan.n_folds, plan.seq_len, dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * n)(*[m.data_ptr() for m in mels])
        cseeds = (C.c_uint64 * n)(*seeds)
        _lib.check(L.mb_wavernn_generate_batch(self._h, C.byref(plan), frames, ptrs, cseeds, _lib.ptr(samples),
                                               _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()),
                   "mb_wavernn_generate_batch")
        torch.cuda.current_stream().synchronize()
        ms, nl = C.c_float(), C.c_int()
        _lib.check(L.mb_wavernn_last_loop_ms(self._h, C.byref(ms), C.byref(nl)), "mb_wavernn_last_loop_ms")
        self.last_loop_ms, self.last_loop_launches, self.last_batch_plan = ms.value, nl.value, plan
        path, fb = C.c_int(), C.c_int()
        _lib.check(L.mb_wavernn_last_path(self._h, C.byref(path), C.byref(fb)), "mb_wavernn_last_path")
        self.last_path = ("chain", "persist1", "pipe", "pipe16")[path.value]
        self.last_fallback = (None, "abort", "range")[fb.value]  # "range": the loop was rerun on the fp32 GEMMs (rnn_ts2_body.h)
        return [samples[offs[u]:offs[u + 1]] for u in range(n)]

    def generate_batch(self, mels, target, overlap, mu_law, seeds=None):
        """Batch counterpart of generate(): list of [80, F_u] mels -> list of float64 waveforms."""
        mu_law = mu_law if self.cfg.mode == 0 else False  # fatchord_version.py:154
        outs = self.generate_samples_batch([m.cuda() for m in mels], target, overlap, seeds)
        return [self.finish(smp, True, overlap, mu_law, (m.shape[-1] - 1) * self.hop_length) for smp, m in zip(outs, mels)]

    def generate(self, mels, batched, target, overlap, mu_law, progress_callback=None, noise=None, seed=None):
        """Signature of WaveRNN.generate (fatchord_version.py:153): mels [1, 80, F] tensor -> float64 wav."""
        mel = mels[0] if mels.dim() == 3 else mels
        mu_law = mu_law if self.cfg.mode == 0 else False  # fatchord_version.py:154
        wave_len = (mel.shape[-1] - 1) * self.hop_length
        samples = self.generate_samples(mel.cuda(), batched, target, overlap, noise=noise, seed=seed,
                                        progress_callback=progress_callback)
        return self.finish(samples, batched, overlap, mu_law, wave_len)

    def finish(self, samples, batched, overlap, mu_law, wave_len, device_out=False):
        """Float64 tail of WaveRNN.generate (fatchord_version.py:236-257) on the device: xfade_and_unfold,
        decode_mu_law, de_emphasis, truncation to wave_len, linear fade-out over 20 hops.  Only the finished
        waveform crosses PCIe.  Returns np.float64 like the reference."""
        if not samples.is_cuda:
            raise _lib.MbHipError("WaveRNN.finish needs a CUDA(HIP) tensor; there is no CPU path")
        samples = samples.to(torch.float32).contiguous()
        n_folds, seq_len = samples.shape
        fade_n = 20 * self.hop_length
        unfolded = n_folds * (seq_len - overlap) + overlap if batched else seq_len
        n_out = min(wave_len, unfolded)
        if n_out < fade_n:  # the reference dies here with numpy's broadcast error (SURVEY finding 5)
            raise ValueError(f"operands could not be broadcast together with shapes ({n_out},) ({fade_n},)")
        L = _lib.lib()
        need = L.mb_wavernn_finish_workspace_bytes(n_folds, seq_len, int(bool(batched)), overlap)
        ws = torch.empty(need, dtype=torch.uint8, device=samples.device)
        wav = torch.empty(n_out, dtype=torch.float64, device=samples.device)
        got = C.c_int()
        _lib.check(L.mb_wavernn_finish(_lib.ptr(samples), n_folds, seq_len, int(bool(batched)), overlap,
                                       self.n_classes, int(bool(mu_law)), int(bool(self.hp.apply_preemphasis)),
                                       float(self.hp.preemphasis), wave_len, fade_n, _lib.ptr(wav), C.byref(got),
                                       _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "mb_wavernn_finish")
        # device_out: keep the float64 waveform in HBM (multi-GPU gather, further device-side post-processing)
        return wav[:got.value] if device_out else wav[:got.value].cpu().numpy()


def load_model(weights_fpath, verbose=True):
    # NB: the toolbox passes a 2nd positional config path that lands in `verbose`
    # (control/toolbox/__init__.py:471); any truthy/falsy value is tolerated.
    global _model, _device
    if verbose:
        print("Building Wave-RNN")
    if not torch.cuda.is_available():
        raise _lib.MbHipError("Wave-RNN: no MI355X visible; this build has no CPU path")
    _device = torch.device('cuda')
    if verbose:
        print("Loading model weights at %s" % weights_fpath)
    checkpoint = torch.load(str(weights_fpath), map_location="cpu")
    _model = WaveRNNDevice(checkpoint['model_state'])


def is_loaded():
    return _model is not None


def infer_waveform(mel, normalize=True, batched=True, target=8000, overlap=800, progress_callback=None):
    if _model is None:
        raise Exception("Please load Wave-RNN in memory before using it")
    if normalize:
        mel = mel / hp.mel_max_abs_value
    mel = torch.from_numpy(mel[None, ...])
    wav = _model.generate(mel, batched, target, overlap, hp.mu_law, progress_callback)
    return wav, hp.sample_rate


def infer_waveform_batch(mels, normalize=True, target=8000, overlap=800, seeds=None, *, peak_normalize=None, pcm16=None,
                         breaks=None, break_hop=None, break_seconds=0.15, break_sample_rate=None, device_out=False):
    """Additive API: a list of (80, F_u) numpy mels -> (list of float64 waveforms, sample
    rate), all utterances sharing ONE batched sample loop.  `normalize` is infer_waveform's mel scaling
    (inference.py:60-61).  The keyword-only tail is the one of the GAN facades' infer_waveform_batch, so that
    pipeline.gen_wavs drives any vocoder: gen_voice.py's host-side tail on the device, in its order --
      breaks[i] = frames per sentence of item i, cut at frames * break_hop samples + break_seconds of silence
                  (gen_voice.py:30-34; gap length from break_sample_rate, default this vocoder's rate);
      peak_normalize = 0.97 -> wav / abs(wav).max() * 0.97 (gen_voice.py:41);
      pcm16 = 'encode_16bits' | 'save_wav' (pinned by goldens) | 'sndfile' (unpinned restatement of libsndfile, by name only) -> int16 PCM;
      device_out=True keeps the results in HBM (device-to-device gather)."""
    if _model is None:
        raise Exception("Please load Wave-RNN in memory before using it")
    ms = [torch.from_numpy(np.asarray(m, np.float32) / (hp.mel_max_abs_value if normalize else 1.0)) for m in mels]
    plain = peak_normalize is None and pcm16 is None and breaks is None and not device_out
    if plain:
        return _model.generate_batch(ms, target, overlap, hp.mu_law, seeds), hp.sample_rate
    from .. import wave
    mu_law = hp.mu_law if _model.cfg.mode == 0 else False  # fatchord_version.py:154
    outs = _model.generate_samples_batch([m.cuda() for m in ms], target, overlap, seeds)
    res = []
    for i, (smp, m) in enumerate(zip(outs, ms)):
        r = _model.finish(smp, True, overlap, mu_law, (m.shape[-1] - 1) * _model.hop_length, device_out=True)
        if breaks is not None:
            r = wave.insert_breaks(r, breaks[i], break_hop, break_sample_rate or hp.sample_rate, break_seconds)
        if peak_normalize is not None:
            r = r.contiguous()
            wave.peak_normalize_(r, peak_normalize)
        if pcm16 is not None:
            r = wave.pack_pcm16(r, pcm16)
        res.append(r if device_out else r.cpu().numpy())
    return res, hp.sample_rate
