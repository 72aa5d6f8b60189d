"""Synthetic flow_record_t streams via libnfagg_synth.so:
generated on the device for the large configs, with a host mirror of the same
generator for cross-checks. Bench/test support, not part of the drop-in ABI."""
import ctypes as C

import numpy as np

from . import _lib as L
from .records import FLOW_RECORD

_s = None


def _lib():
    global _s
    if _s is None:
        _s = L.load_synth()
    return _s


def zipf_thresholds(n_keys: int, s: float) -> np.ndarray:
    out = np.zeros(n_keys, dtype=np.uint64)
    _lib().nfagg_synth_zipf_thresholds(n_keys, s, out.ctypes.data_as(C.c_void_p))
    return out


def shard_population(n_keys: int, n_shards: int, shard: int) -> np.ndarray:
    out = np.zeros(n_keys, dtype=np.uint64)
    _lib().nfagg_synth_shard_population(n_keys, n_shards, shard, out.ctypes.data_as(C.c_void_p))
    return out


def stream_host(n, j0=0, seed=1, n_keys=1000, thresholds=None, hot_permille=0, variant=0, pop_index=None) -> np.ndarray:
    out = np.zeros(n, dtype=FLOW_RECORD)
    th = thresholds.ctypes.data_as(C.c_void_p) if thresholds is not None else None
    pi = pop_index.ctypes.data_as(C.c_void_p) if pop_index is not None else None
    _lib().nfagg_synth_stream_host(out.ctypes.data_as(C.c_void_p), n, j0, seed, n_keys, th, hot_permille, variant, pi)
    return out


def stream_device(d_out: int, n, j0=0, seed=1, n_keys=1000, d_thresholds: int = 0, hot_permille=0, variant=0,
                  d_pop_index: int = 0, stream: int = 0):
    """Fill device memory at d_out with n records. d_thresholds / d_pop_index are
    device pointers (0 = none)."""
    rc = _lib().nfagg_synth_stream(C.c_void_p(d_out), n, j0, seed, n_keys, C.c_void_p(d_thresholds or None),
                                   hot_permille, variant, C.c_void_p(d_pop_index or None), C.c_void_p(stream or None))
    if rc != 0:
        raise RuntimeError("nfagg_synth_stream launch failed")


def yardstick(which: int, d_a: int, d_b: int, nbytes: int, d_sink: int, reps: int = 3) -> float:
    """Milliseconds per pass of one of the HBM yardstick kernels (nfagg_synth.hip): 0 = records read as the fold reads them (seven
    16-byte loads of every 144-byte record), 1 = plain 16-byte stream read, 2 = 16-byte copy d_a -> d_b. bench.py's roofline notes."""
    ms = _lib().nfagg_synth_yardstick(which, C.c_void_p(d_a), C.c_void_p(d_b or None), nbytes, C.c_void_p(d_sink), reps)
    if ms < 0:
        raise RuntimeError("yardstick kernel failed")
    return ms
