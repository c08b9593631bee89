"""The subset of models/synthesizer/hparams.py:3-78 the inference path reads, as an attribute
bag with the reference's ``loadJson`` override semantics (utils/hparams.py:88-99)."""
import json


class HParams:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __getitem__(self, k):
        return getattr(self, k)

    def __contains__(self, k):
        return k in self.__dict__

    def loadJson(self, fpath):
        with open(fpath, "r", encoding="utf-8") as f:
            data = json.load(f)
        for k, v in data.items():
            if k not in ["tts_schedule", "tts_finetune_layers"]:
                self.__dict__[k] = v
        return self


hparams = HParams(
    sample_rate=16000, n_fft=1024, num_mels=80, hop_size=256, win_size=1024, fmin=55, min_level_db=-100,
    ref_level_db=20, max_abs_value=4., preemphasis=0.97, preemphasize=True,
    tts_embed_dims=512, tts_encoder_dims=256, tts_decoder_dims=128, tts_postnet_dims=512, tts_encoder_K=5,
    tts_lstm_dims=1024, tts_postnet_K=5, tts_num_highways=4, tts_dropout=0.5, tts_cleaner_names=["basic_cleaners"],
    tts_stop_threshold=-3.4, max_mel_frames=900, rescale=True, rescaling_max=0.9, synthesis_batch_size=16,
    speaker_embedding_size=256, use_gst=True, use_ser_for_gst=True,
)
