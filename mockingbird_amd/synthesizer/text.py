"""Text -> token ids with the reference's table and cleaner
(models/synthesizer/utils/symbols.py:10-18, text.py:13-40, cleaners.py:72-76 basic_cleaners).
Host-side glue; the accelerated path starts at the token ids."""
import re

_pad, _eos = "_", "~"
_characters = 'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz1234567890!\'(),-.:;? '
symbols = [_pad, _eos] + list(_characters)
_symbol_to_id = {s: i for i, s in enumerate(symbols)}
_whitespace_re = re.compile(r"\s+")
_curly_re = re.compile(r"(.*?)\{(.+?)\}(.*)")


def basic_cleaners(text):
    return re.sub(_whitespace_re, " ", text.lower())


def _symbols_to_sequence(syms):
    return [_symbol_to_id[s] for s in syms if s in _symbol_to_id and s not in ("_", "~")]


def text_to_sequence(text, cleaner_names=("basic_cleaners",)):
    if list(cleaner_names) != ["basic_cleaners"]:
        raise ValueError("only basic_cleaners (the reference's tts_cleaner_names default) is restated here")
    sequence = []
    while len(text):
        m = _curly_re.match(text)
        if not m:
            sequence += _symbols_to_sequence(basic_cleaners(text))
            break
        sequence += _symbols_to_sequence(basic_cleaners(m.group(1)))
        sequence += _symbols_to_sequence(["@" + s for s in m.group(2).split()])  # ARPAbet: not in the table
        text = m.group(3)
    sequence.append(_symbol_to_id["~"])
    return sequence


def to_pinyin(texts):
    """inference.py:100: lazy_pinyin(TONE3, neutral_tone_with_five).  pypinyin leaves non-Chinese
    text untouched, so without it ASCII prompts are still handled exactly."""
    try:
        from pypinyin import lazy_pinyin, Style
    except ImportError:
        for v in texts:
            if any(ord(ch) > 127 for ch in v):
                raise ImportError("pypinyin is required for non-ASCII prompts (reference front-end dependency)")
        return [" ".join([v]) for v in texts]
    return [" ".join(lazy_pinyin(v, style=Style.TONE3, neutral_tone_with_five=True)) for v in texts]
