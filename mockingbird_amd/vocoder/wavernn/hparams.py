"""Values of models/vocoder/wavernn/hparams.py:1-44 (which itself mirrors
models/synthesizer/hparams.py:5-16) that the inference path reads."""
sample_rate = 16000
n_fft = 1024
num_mels = 80
hop_length = 256          # synthesizer hop_size (hparams.py:8); the vocoder upsamples x200 (SURVEY finding 5)
win_length = 1024
mel_max_abs_value = 4.0
preemphasis = 0.97
apply_preemphasis = True
bits = 9
mu_law = True
voc_mode = 'RAW'
voc_upsample_factors = (5, 5, 8)
voc_rnn_dims = 512
voc_fc_dims = 512
voc_compute_dims = 128
voc_res_out_dims = 128
voc_res_blocks = 10
voc_pad = 2
voc_gen_batched = True
voc_target = 8000
voc_overlap = 400
