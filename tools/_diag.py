"""MBHIP_DIAG="key=value,key,..." is the one environment variable behind libmbhip's diagnostics, A/B knobs and test hooks
(csrc/common.hip diag_str).  diag_set edits one key of it in this process, leaving the others alone."""
import os


def diag_set(key, value=None):
    """value=None removes the key; value='' sets the bare key (reads as 1)."""
    items = [kv for kv in os.environ.get("MBHIP_DIAG", "").split(",") if kv and kv.split("=", 1)[0] != key]
    if value is not None:
        items.append(key if value == "" else f"{key}={value}")
    if items:
        os.environ["MBHIP_DIAG"] = ",".join(items)
    else:
        os.environ.pop("MBHIP_DIAG", None)


def diag_get(key):
    for kv in os.environ.get("MBHIP_DIAG", "").split(","):
        k, _, v = kv.partition("=")
        if k == key:
            return v or "1"
    return None
