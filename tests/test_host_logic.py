"""Host-side logic that needs no GPU: weight folding/packing, generate() geometry, waveform
post-processing, utterance sharding."""
import ctypes as C

import numpy as np
import pytest
import torch

import synth
from oracle import gan as og, wavernn as ow


def test_fold_weight_norm_matches_torch():
    from mockingbird_amd import weights
    st = synth.gan_state(synth.small(synth.HIFIGAN_16K, 32), "hifigan", seed=2)["generator"]
    ref = og.fold_weight_norm_state(st)
    for name in ("conv_pre", "ups.1", "resblocks.3.convs1.2", "conv_post"):
        w = weights.fold_weight_norm(st, name)
        assert torch.allclose(w, ref[name + ".weight"], rtol=1e-6, atol=1e-7)
    # ConvTranspose1d: norm is over dims (1,2) per INPUT channel (dim 0)
    v = st["ups.0.weight_v"]
    w = weights.fold_weight_norm(st, "ups.0")
    assert torch.allclose(w.reshape(v.shape[0], -1).norm(dim=1), st["ups.0.weight_g"].flatten(), rtol=1e-5)


def test_gan_weight_list_matches_abi_counts(lib):
    from mockingbird_amd import weights
    for kind, cfg in (("hifigan", synth.HIFIGAN_16K), ("fregan", synth.FREGAN_16K), ("hifigan", synth.HIFIGAN_24K)):
        h = synth.small(cfg, 32)
        c = weights.gan_config(h, 0 if kind == "hifigan" else 1)
        assert c.interp_ups == int(cfg is synth.HIFIGAN_24K)  # models.py:107: only h.sampling_rate == 24000
        ws = weights.gan_weight_list(synth.gan_state(h, kind, seed=1)["generator"], c)
        assert lib.mb_gan_num_weights(C.byref(c)) == len(ws)
        for i, w in enumerate(ws):
            assert lib.mb_gan_weight_numel(C.byref(c), i) == w.numel(), i


def test_wavernn_weight_list_matches_abi_counts(lib):
    from mockingbird_amd import weights
    from mockingbird_amd.vocoder.wavernn import hparams as hp
    c = weights.wavernn_config(hp)
    ws = weights.wavernn_weight_list(synth.wavernn_state(seed=1)["model_state"], c)
    assert lib.mb_wavernn_num_weights(C.byref(c)) == len(ws)
    for i, w in enumerate(ws):
        assert lib.mb_wavernn_weight_numel(C.byref(c), i) == w.numel(), i


def test_conv_pack_is_a_permutation_with_zero_padding(lib):
    """mb_conv1d_pack: the image is [fp32 A fragments | 64-float header | fp16 hi / lo / hi 2^-11 A fragments].  Every weight appears
    exactly once in the fp32 part (conv and all polyphase taps of the transposed conv), padding is zero; the split part
    holds w * 2^s as hi + lo halves (header word 0 = 2^-s) that add back to the weight to 22 bits, and the third image is
    hi * 2^-11 (the partner of the scaled activation residual, conv1d.hip)."""
    import hiputil
    for (shape, transposed, up, pad) in (((40, 20, 3), False, 1, 1), ((24, 16, 10), True, 5, 3), ((64, 32, 4), True, 2, 1)):
        w = torch.arange(1, int(np.prod(shape)) + 1, dtype=torch.float32).reshape(shape) / 1000.0
        packed, (c_out, c_in, k) = hiputil.pack_conv(w, transposed, up, pad)
        n_mt, n_cb, n_ks = (c_out + 31) // 32, (c_in + 7) // 8, (c_in + 15) // 16
        n_f32 = n_mt * n_cb * k * 256
        assert packed.numel() == n_f32 + 64 + n_mt * n_ks * k * 768
        f32 = packed[:n_f32]
        nz = f32[f32 != 0]
        assert nz.numel() == w.numel() and torch.equal(nz.sort().values, w.flatten().sort().values)
        unscale = float(packed[n_f32])
        assert unscale > 0 and np.log2(unscale) == round(np.log2(unscale)) and float(packed[n_f32 + 1:n_f32 + 64].abs().max()) == 0
        assert 2 ** 13 <= float(w.abs().max()) / unscale < 2 ** 14
        halves = packed[n_f32 + 64:].view(torch.float16).reshape(-1, 3, 512).double()  # [(p, mt, ks, tap)][hi | lo | hi 2^-11][lane * 8 + e]
        assert torch.equal(halves[:, 2] * 2048.0, halves[:, 0])  # exact: hi is a normal fp16 here (max |w| 2^s >= 2^13)
        rec = ((halves[:, 0] + halves[:, 1]) * unscale).flatten()
        rnz = rec[rec != 0]
        assert rnz.numel() == w.numel()
        err = (rnz.sort().values - w.flatten().double().sort().values).abs() / w.flatten().double().sort().values
        assert float(err.max()) < 2.0 ** -21, float(err.max())


@pytest.mark.parametrize("frames,target,overlap", [(30, 600, 100), (1000, 8000, 800), (200, 8000, 800), (44, 8000, 800)])
def test_plan_matches_fold_with_overlap(frames, target, overlap):
    """Fold geometry of mb_wavernn_plan_generate == fold_with_overlap (fatchord_version.py:313-322);
    checked through the oracle's restatement on a dummy tensor."""
    x = torch.zeros(1, frames * 200, 1)
    folded = ow.fold_with_overlap(x, target, overlap)
    total = frames * 200
    nf = (total - overlap) // (target + overlap)
    if total - (nf * (target + overlap) + overlap) != 0:
        nf += 1
    assert folded.shape[0] == nf and folded.shape[1] == target + 2 * overlap


def test_shard_indices_balanced_and_complete():
    from mockingbird_amd.sharding import shard_indices
    lens = [90, 110, 95, 101, 99, 108, 93, 97, 100, 105, 91, 102]
    for world in (1, 2, 4, 8):
        parts = [shard_indices(lens, world, r) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(len(lens)))
        loads = [sum(lens[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lens)


def test_tacotron_weight_list_matches_abi_counts(lib):
    from mockingbird_amd import weights
    st = synth.tacotron_state(seed=1)["model_state"]
    c = weights.taco_config(st)
    assert (c.n_mels, c.project_dims, c.decoder_dims, c.lstm_dims, c.r, c.postnet_K, c.num_highways) == (80, 1024, 128, 1024, 2, 5, 4)
    assert (c.has_encoder, c.num_chars, c.embed_dims, c.encoder_dims, c.encoder_K, c.speaker_dims, c.style_dims) == (1, 75, 512, 256, 5, 256, 512)
    ws = weights.taco_weight_list(st, c)
    assert lib.mb_taco_num_weights(C.byref(c)) == len(ws)
    for i, w in enumerate(ws):
        assert lib.mb_taco_weight_numel(C.byref(c), i) == w.numel(), i


def test_resblock_pair_weight_stream_layout(lib):
    """mb_resblock_pair_f16_pack (host code): the fp16 image is ONE circular stream per 32-channel output
    tile in the kernel's consumption order [conv1: chunk, tap, k-block][conv2: ...][lane][8], with the
    v_mfma_f32_32x32x16_f16 A-fragment lane map (lane l holds row l&31, k = 8*(l>>5) + e)."""
    import ctypes as C_
    from mockingbird_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for Cc, k in ((64, 3), (32, 7), (128, 3), (16, 3)):
        w1 = rng.standard_normal((Cc, Cc, k)).astype(np.float32)
        w2 = rng.standard_normal((Cc, Cc, k)).astype(np.float32)
        n = L.mb_resblock_pair_f16_packed_halves(Cc, k)
        assert n == max(1, Cc // 32) * 2 * (Cc // 16) * k * 512
        img = np.empty(n, np.float16)
        _lib.check(L.mb_resblock_pair_f16_pack(w1.ctypes.data, w2.ctypes.data, Cc, k, img.ctypes.data),
                   "mb_resblock_pair_f16_pack")
        CK = 64 if Cc >= 64 else (32 if Cc >= 32 else 16)
        KB, NCH, MTT = CK // 16, Cc // CK, max(1, Cc // 32)
        img = img.reshape(MTT, 2, NCH, k, KB, 64, 8)
        for mt, ph, c, j, u, lane in [(0, 0, 0, 0, 0, 0), (MTT - 1, 1, NCH - 1, k - 1, KB - 1, 63), (MTT // 2, 1, 0, 1, min(1, KB - 1), 37)]:
            w = (w1, w2)[ph]
            co = mt * 32 + (lane & 31)
            ci0 = c * CK + u * 16 + (lane >> 5) * 8
            want = w[co, ci0:ci0 + 8, j].astype(np.float16) if co < Cc else np.zeros(8, np.float16)
            assert np.array_equal(img[mt, ph, c, j, u, lane], want)
    assert L.mb_resblock_pair_f16_supported(64, 11, 5) == 1 and L.mb_resblock_pair_f16_supported(48, 3, 1) == 0
    assert L.mb_resblock_pair_f16_supported(64, 4, 1) == 0  # even kernel sizes have no "same" padding


def test_resblock_stage_weight_stream_layout(lib):
    """mb_resblock_stage_f16_pack / _supported (host code): ONE circular fp16 stream in the kernel's consumption order
    [ResBlock j][unit u][conv1 | conv2][tap][k-block][lane][8] (one 32-row output tile, rows >= C zero), the
    v_mfma_f32_32x32x16_f16 A-fragment lane map; the supported() rule follows LDS (window + halo of the widest ResBlock)."""
    import ctypes as C_
    from mockingbird_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(1)
    for Cc, ks, nd in ((32, (3, 7, 11), 3), (16, (3, 5), 4), (32, (7,), 2)):
        nk = len(ks)
        w1 = [rng.standard_normal((Cc, Cc, ks[j])).astype(np.float32) for j in range(nk) for _ in range(nd)]
        w2 = [rng.standard_normal((Cc, Cc, ks[j])).astype(np.float32) for j in range(nk) for _ in range(nd)]
        karr = (C_.c_int * nk)(*ks)
        n = L.mb_resblock_stage_f16_packed_halves(Cc, nk, karr, nd)
        KB = Cc // 16
        assert n == sum(2 * nd * k for k in ks) * KB * 512
        img = np.empty(n, np.float16)
        p1 = (C_.c_void_p * len(w1))(*[w.ctypes.data for w in w1])
        p2 = (C_.c_void_p * len(w2))(*[w.ctypes.data for w in w2])
        _lib.check(L.mb_resblock_stage_f16_pack(p1, p2, Cc, nk, karr, nd, img.ctypes.data), "mb_resblock_stage_f16_pack")
        o = 0
        for j in range(nk):
            for u in range(nd):
                for ph, ws in ((0, w1), (1, w2)):
                    w = ws[j * nd + u]
                    blk = img[o:o + ks[j] * KB * 512].reshape(ks[j], KB, 64, 8)
                    o += ks[j] * KB * 512
                    for tap, kb, lane in ((0, 0, 0), (ks[j] - 1, KB - 1, 63), (ks[j] // 2, 0, 37)):
                        co, ci0 = lane & 31, kb * 16 + (lane >> 5) * 8
                        want = w[co, ci0:ci0 + 8, tap].astype(np.float16) if co < Cc else np.zeros(8, np.float16)
                        assert np.array_equal(blk[tap, kb, lane], want), (Cc, j, u, ph, tap, kb, lane)
        assert o == n
    k3 = (C_.c_int * 3)(3, 7, 11)
    d135 = (C_.c_int * 9)(*([1, 3, 5] * 3))
    d1357 = (C_.c_int * 12)(*([1, 3, 5, 7] * 3))
    assert L.mb_resblock_stage_f16_supported(32, 3, k3, 3, d135) == 1 and L.mb_resblock_stage_f16_supported(32, 3, k3, 4, d1357) == 1
    assert L.mb_resblock_stage_f16_supported(16, 3, k3, 4, d1357) == 1
    assert L.mb_resblock_stage_f16_supported(64, 3, k3, 3, d135) == 1 and L.mb_resblock_stage_f16_supported(128, 3, k3, 3, d135) == 0
    k1 = (C_.c_int * 1)(3)
    d1 = (C_.c_int * 3)(1, 3, 5)
    assert L.mb_resblock_stage_f16_supported(128, 1, k1, 3, d1) == 1 and abs(L.mb_resblock_stage_f16_efficiency(128, 1, k1, 3, d1) - 104 / 128) < 1e-6
    assert L.mb_resblock_stage_f16_supported(48, 1, k1, 3, d1) == 0
    k4 = (C_.c_int * 3)(3, 4, 11)
    assert L.mb_resblock_stage_f16_supported(32, 3, k4, 3, d135) == 0   # even kernel sizes have no "same" padding
    big = (C_.c_int * 9)(*([1, 9, 27] * 3))
    assert L.mb_resblock_stage_f16_supported(32, 3, k3, 3, big) == 0    # the halo of the widest ResBlock leaves no rows in the tile


def test_wavernn_finish_workspace_and_shape_errors(lib):
    """Host-side checks of mb_wavernn_finish: workspace formula, argument validation before any launch."""
    from mockingbird_amd import _lib
    L = _lib.lib()
    n_folds, S, ov = 23, 9600, 800
    unfolded = n_folds * (S - ov) + ov
    need = L.mb_wavernn_finish_workspace_bytes(n_folds, S, 1, ov)
    assert need >= unfolded * 8 and need < unfolded * 8 + 64 * 1024
    assert L.mb_wavernn_finish_workspace_bytes(1, 100, 1, 60) == 0  # seq_len <= 2*overlap
    got = C.c_int()
    rc = L.mb_wavernn_finish(None, 1, 100, 0, 0, 512, 1, 1, 0.97, 50, 10, None, C.byref(got), None, 0, None)
    assert rc < 0 and b"null pointer" in L.mb_last_error()


def test_wave_wire_format_has_no_cpu_path(lib):
    """vocoder/wave.py refuses host tensors and unknown encodings loudly (no numpy fallback in the product)."""
    from mockingbird_amd.vocoder import wave
    from mockingbird_amd._lib import MbHipError
    with pytest.raises(MbHipError, match="no CPU path"):
        wave.pack_pcm16(torch.zeros(8))
    with pytest.raises(MbHipError, match="no CPU path"):
        wave.peak_normalize_(torch.zeros(8, dtype=torch.float64))
    with pytest.raises(ValueError, match="mode must be one of"):
        wave.pack_pcm16(torch.zeros(8), "pcm24")
    assert lib.mb_wave_workspace_bytes() >= 8
    # the default encoding is one pinned by a golden of the reference function; the unpinned libsndfile restatement is by name only
    import inspect
    assert inspect.signature(wave.pack_pcm16).parameters["mode"].default == "encode_16bits"


def test_gan_out_samples_rule_24k():
    """Length rule of the 24 kHz upsampler (models.py:107-112): T*u + 2*((k-1)//2) - (k-1) per stage."""
    t = 16
    for u, k in zip(synth.HIFIGAN_24K["upsample_rates"], synth.HIFIGAN_24K["upsample_kernel_sizes"]):
        t = t * u + 2 * ((k - 1) // 2) - (k - 1)
    import numpy as np, os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gan24k.npz"))
    assert g["hifigan24k_uic64_f16_b2_s5"].shape[-1] == t == 4785


def test_bench_traffic_floor_counts_every_launch_once():
    """bench.gan_floor_bytes (the denominator of `traffic / floor` in DESIGN 4b'): read-x + write-y of every launch of an fp16
    generator forward as the launcher in gan.hip issues them.  Hand count for HiFi-GAN V1 at 1 x 1 frame, and the Fre-GAN-only
    launches (cond_up, x += mel, res_output (+ x)) on top of the shared structure."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    h = synth.HIFIGAN_16K
    # conv_pre: 80 -> 512; stages (u, C): (5, 256) (5, 128) (4, 64) (2, 32); per stage: upsampler read + write, then
    # 256 ch: 3 ResBlocks x 3 units x (read + write) + 2 accumulator reads; 128 / 64 ch: k = 3 is ONE launch (2) + 2 x 3 units x 2
    # + 2 accumulator reads; 32 ch: one launch for the group (2); conv_post: read 32 ch, write fp32
    T, el = 1, 2.0
    want = (80 + 512) * el
    C = 512
    for u, per_elem in ((5, 18 + 2), (5, 2 + 12 + 2), (4, 2 + 12 + 2), (2, 2)):
        want += T * C * el
        T, C = T * u, C // 2
        want += T * C * el + per_elem * T * C * el
    want += T * C * el + T * 4.0
    assert bench.gan_floor_bytes(h, 1, 1) == want
    hf = synth.FREGAN_16K
    base = bench.gan_floor_bytes(hf, 2, 7)
    assert bench.gan_floor_bytes(hf, 2, 7, fregan=True) > base               # that generator's own launches
    assert bench.gan_floor_bytes(hf, 4, 7, fregan=True) == 2 * bench.gan_floor_bytes(hf, 2, 7, fregan=True)  # linear in the batch


def test_wavernn_loop_path_table(lib):
    """The ONE selection function of the WaveRNN sample loop (csrc/wavernn.hip wavernn_pick_path, exported as
    mb_wavernn_loop_path): (columns, mode, production, images, residency, failure memo, switches) -> path.  VERDICT r03 item 8."""
    CHAIN, P1, PIPE, PIPE16 = 0, 1, 2, 3
    f = lib.mb_wavernn_loop_path
    U, OFF, ON, EXACT = -1, 0, 1, 2  # MBHIP_WAVERNN_RESIDENT: unset / 0 / 1 / exact
    #        columns mode prod q16 cus failed resident -> path
    table = [
        (1, 0, 1, 1, 256, 0, U, PIPE16), (1, 1, 1, 1, 256, 0, U, PIPE16), (1, 0, 1, 1, 256, 0, OFF, CHAIN),           # one column: the operand-pair kernel (round 5)
        (1, 0, 1, 0, 256, 0, U, P1), (1, 1, 1, 0, 256, 0, U, CHAIN), (1, 0, 1, 1, 256, 0, EXACT, P1), (1, 1, 1, 1, 256, 0, EXACT, CHAIN),  # no images / exact: fmaf-chain kernel, RAW only
        (1, 0, 1, 1, 191, 0, U, CHAIN), (1, 0, 1, 1, 192, 0, U, P1), (1, 0, 1, 1, 223, 0, U, P1), (1, 0, 1, 1, 224, 0, U, PIPE16),  # 192 / 224 workgroups
        (1, 0, 1, 1, 256, 1, U, CHAIN), (1, 0, 1, 1, 256, 1, ON, PIPE16), (1, 0, 1, 1, 256, 1, EXACT, P1),            # failure memo / explicit switch
        (2, 0, 1, 1, 256, 0, U, PIPE16), (23, 0, 1, 1, 256, 0, U, PIPE16), (32, 0, 1, 1, 256, 0, U, PIPE16),
        (33, 0, 1, 1, 256, 0, U, PIPE16), (64, 0, 1, 1, 256, 0, U, PIPE16), (65, 0, 1, 1, 256, 0, U, PIPE16), (96, 0, 1, 1, 256, 0, U, PIPE16),
        (97, 0, 1, 1, 256, 0, U, CHAIN), (97, 0, 1, 1, 256, 0, ON, CHAIN), (65, 0, 1, 1, 256, 0, EXACT, CHAIN),       # six groups of 16
        (64, 1, 1, 1, 256, 0, U, PIPE16), (65, 1, 1, 1, 256, 0, U, CHAIN), (65, 1, 1, 1, 256, 0, ON, CHAIN),          # MOL: the fused chain's 64 columns
        (23, 0, 1, 1, 256, 0, EXACT, PIPE), (33, 0, 1, 1, 256, 0, EXACT, CHAIN),                                      # the exact kernel: 32 columns
        (23, 0, 1, 0, 256, 0, U, PIPE), (23, 1, 1, 1, 256, 0, U, PIPE16), (40, 1, 1, 1, 256, 0, U, PIPE16),           # no images / MOL
        (23, 1, 1, 1, 256, 0, EXACT, PIPE), (40, 1, 1, 1, 256, 0, EXACT, CHAIN), (23, 1, 1, 0, 256, 0, U, PIPE),
        (23, 0, 1, 1, 256, 0, OFF, CHAIN), (23, 0, 1, 1, 256, 0, ON, PIPE16),
        (23, 0, 1, 1, 223, 0, U, CHAIN), (23, 0, 1, 1, 224, 0, U, PIPE16), (23, 0, 1, 1, 0, 0, ON, CHAIN),            # 224 workgroups
        (23, 0, 1, 1, 256, 1, U, CHAIN), (23, 0, 1, 1, 256, 1, ON, PIPE16),                                           # failure memo
        (23, 0, 0, 1, 256, 0, ON, CHAIN), (0, 0, 1, 1, 256, 0, U, CHAIN),                                             # not a production call
    ]
    for row in table:
        assert f(*row[:7]) == row[7], row


def test_taco_loop_form_table(lib):
    """The ONE selection function of the production-dims Tacotron decoder loop (csrc/tacotron.hip taco_pick_form, exported as
    mb_taco_loop_form): (batch, text length, fast attention kernel, compute units, images, lost-hand-off memo, MBHIP_DIAG taco_front /
    taco_f16 / taco_fold) -> launches per iteration and whether the K >= 1024 tiles multiply on the fp16 pipe."""
    import ctypes as C
    U = -1
    #        batch T   lsa cus img failed front f16 fold -> (launches, f16)
    table = [
        (32, 101, 1, 256, 1, 0, U, U, U, (4, 1)), (16, 96, 1, 256, 1, 0, U, U, U, (4, 1)), (1, 12, 1, 256, 1, 0, U, U, U, (4, 1)),      # <= 128 symbols: the folded form
        (32, 128, 1, 256, 1, 0, U, U, U, (4, 1)), (32, 129, 1, 256, 1, 0, U, U, U, (5, 1)), (17, 192, 1, 256, 1, 0, U, U, U, (5, 1)),   # 129..192: the LDS window is gone
        (16, 129, 1, 256, 1, 0, U, U, U, (5, 0)), (1, 192, 1, 256, 1, 0, U, U, U, (5, 0)),                                              # ... and one column tile stays on the fp32 pipe
        (32, 193, 0, 256, 1, 0, U, U, U, (7, 0)), (32, 640, 0, 256, 1, 0, U, U, U, (7, 0)),                                             # the general attention kernel
        (33, 101, 1, 256, 1, 0, U, U, U, (7, 0)), (64, 101, 1, 256, 1, 0, U, U, U, (7, 0)), (0, 101, 1, 256, 1, 0, U, U, U, (7, 0)),    # more than two column tiles
        (32, 101, 1, 175, 1, 0, U, U, U, (7, 0)), (32, 101, 1, 176, 1, 0, U, U, U, (4, 1)), (8, 101, 1, 80, 1, 0, U, U, U, (4, 1)),     # 48 + 4 batch co-resident workgroups
        (32, 101, 1, 256, 1, 1, U, U, U, (7, 0)), (32, 101, 1, 256, 1, 1, 1, U, U, (7, 0)),                                             # a lost hand-off is remembered
        (32, 101, 1, 256, 1, 0, 0, U, U, (7, 0)), (32, 101, 1, 256, 1, 0, 0, 1, 1, (7, 0)),                                             # taco_front=0
        (32, 101, 1, 256, 1, 0, U, 0, U, (5, 0)), (16, 101, 1, 256, 1, 0, U, 0, U, (5, 0)),                                             # taco_f16=0: no short riders, no fold
        (32, 101, 1, 256, 1, 0, U, U, 0, (5, 1)), (16, 101, 1, 256, 1, 0, U, U, 0, (5, 0)),                                             # taco_fold=0
        (32, 101, 1, 256, 1, 0, U, 0, 1, (4, 0)), (16, 150, 1, 256, 1, 0, U, 1, U, (5, 1)), (16, 150, 1, 256, 1, 0, U, 1, 1, (5, 1)),   # forced combinations (A/B only)
        (32, 101, 1, 256, 0, 0, U, U, U, (5, 0)), (32, 101, 1, 256, 0, 0, U, 1, 1, (5, 0)),                                             # a handle without the images
    ]
    for row in table:
        f16 = C.c_int(-1)
        got = lib.mb_taco_loop_form(*row[:9], C.byref(f16))
        assert (got, f16.value) == row[9], row
    assert lib.mb_taco_loop_form(32, 101, 1, 256, 1, 0, U, U, U, None) == 4  # the flag is optional


def test_switch_inventory_is_small_and_documented():
    """the library reads at most 20 environment switches and DESIGN.md's table names every one of them (a switch
    that is not documented -- or not tested: tests/test_env_switches_gpu.py and the tests named in the table -- does not stay)."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for pat in ("mockingbird_amd/csrc/*.hip", "mockingbird_amd/csrc/*.h", "mockingbird_amd/*.py", "mockingbird_amd/*/*.py", "mockingbird_amd/*/*/*.py"):
        for path in glob.glob(os.path.join(root, pat)):
            names |= set(re.findall(r'(?:getenv\(|environ(?:\.get)?[\[(])\s*"(MBHIP_[A-Z0-9_]+)"', open(path, errors="replace").read()))
    assert 10 <= len(names) <= 20, sorted(names)
    design = open(os.path.join(root, "DESIGN.md")).read()
    table = design[design.index("### Run-time switches"):]
    table = table[:table.index("\n## ")]
    missing = [n for n in sorted(names) if "`" + n not in table]
    assert not missing, missing
    # no comment or docstring of the library still names a switch that no longer exists
    mentioned = set()
    for pat in ("mockingbird_amd/csrc/*.hip", "mockingbird_amd/csrc/*.h", "include/*.h"):
        for path in glob.glob(os.path.join(root, pat)):
            mentioned |= set(re.findall(r"MBHIP_[A-Z0-9_]+", open(path, errors="replace").read()))
    mentioned.discard("MBHIP_H")  # the header's include guard
    assert mentioned <= names, sorted(mentioned - names)


def test_diag_variable_parsing(lib, monkeypatch):
    """MBHIP_DIAG="key=value,key,...": exact key match (no prefix hits), bare keys read as 1, values may contain '=' and '/'."""
    def get(key):
        buf = C.create_string_buffer(64)
        n = lib.mb_diag_lookup(key.encode(), buf, 64)
        return None if n < 0 else buf.value.decode()
    monkeypatch.delenv("MBHIP_DIAG", raising=False)
    assert get("wq_flags") is None
    monkeypatch.setenv("MBHIP_DIAG", "wq_flags=17,abort_wp,wp_trace=/tmp/a=b.bin,ts3_dbg=3")
    assert get("wq_flags") == "17" and get("abort_wp") == "1" and get("wp_trace") == "/tmp/a=b.bin" and get("ts3_dbg") == "3"
    assert get("wq") is None and get("abort") is None and get("trace") is None
    monkeypatch.setenv("MBHIP_DIAG", "abort_pr")
    assert get("abort_pr") == "1" and get("abort_wp") is None
    monkeypatch.setenv("MBHIP_DIAG", "")
    assert get("abort_pr") is None


def test_resblock_stage_f32_weight_stream_layout(lib):
    """mb_resblock_stage_f32_pack (CPU only): one stream per 32-row output tile in consumption order [chain][unit][conv1 | conv2][tap]
    [16-channel k-step][hi | lo | hi 2^-11][lane][8]; (hi + lo) x 2^-s gives the weight back to 2^-21 of the conv's largest weight,
    the third image is hi 2^-11, the scale puts that largest weight into [2^13, 2^14)."""
    import ctypes as C
    rng = np.random.default_rng(3)
    for ch, ks, nd in ((32, [3, 7], 2), (64, [3], 3), (64, [11], 1)):
        nk = len(ks)
        w1 = [rng.standard_normal((ch, ch, ks[c])).astype(np.float32) * (0.02 + 0.3 * c) for c in range(nk) for _ in range(nd)]
        w2 = [rng.standard_normal((ch, ch, ks[c])).astype(np.float32) * 0.05 for c in range(nk) for _ in range(nd)]
        ksa = (C.c_int * nk)(*ks)
        dil = (C.c_int * (nk * nd))(*([1, 3, 5][:nd] * nk))
        assert lib.mb_resblock_stage_f32_supported(ch, nk, ksa, nd, dil) == 1
        assert lib.mb_resblock_stage_f32_supported(128, nk, ksa, nd, dil) == 0
        eff = lib.mb_resblock_stage_f32_efficiency(ch, nk, ksa, nd, dil)
        assert 0.2 < eff <= 1.0
        n = lib.mb_resblock_stage_f32_packed_halves(ch, nk, ksa, nd)
        assert n == (ch // 32) * sum(2 * nd * k for k in ks) * (ch // 16) * 3 * 512
        packed = np.zeros(n, np.float16)
        unscale = np.zeros(nk * nd * 2, np.float32)
        p1 = (C.c_void_p * (nk * nd))(*[w.ctypes.data for w in w1])
        p2 = (C.c_void_p * (nk * nd))(*[w.ctypes.data for w in w2])
        assert lib.mb_resblock_stage_f32_pack(p1, p2, ch, nk, ksa, nd, packed.ctypes.data, unscale.ctypes.data) == 0
        o = 0
        for mt in range(ch // 32):
            for c in range(nk):
                for u in range(nd):
                    for ph in range(2):
                        w = (w2 if ph else w1)[c * nd + u]
                        us = float(unscale[(c * nd + u) * 2 + ph])
                        wmax = float(np.abs(w).max())
                        assert 2.0 ** 13 <= wmax / us < 2.0 ** 14 and np.log2(us) == round(np.log2(us))
                        for j in range(ks[c]):
                            for kb in range(ch // 16):
                                blk = packed[o:o + 1536].astype(np.float64).reshape(3, 64, 8)
                                o += 1536
                                lane = np.arange(64)[:, None]
                                e = np.arange(8)[None, :]
                                co = mt * 32 + (lane & 31)
                                ci = kb * 16 + (lane >> 5) * 8 + e
                                ref = w[co, ci, j].astype(np.float64)
                                assert np.abs((blk[0] + blk[1]) * us - ref).max() <= wmax * 2.0 ** -21
                                assert np.array_equal(blk[2], (blk[0] / 2048.0).astype(np.float16).astype(np.float64))
        assert o == n
