"""Experiment: where does an epoch of the caller-driven nfagg_ingest / nfagg_evict loop (CACHE_MAX_FLOWS 5000, host buffers) spend its
time? Raw ctypes on NFAGG_LIB (so that older builds can be compared). usage: small_table_phases.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from oracle import oracle as O

lib = C.CDLL(os.environ["NFAGG_LIB"])


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_entries", C.c_uint64), ("table_log2_slots", C.c_uint32), ("mode", C.c_uint32),
                ("sketch_flags", C.c_uint32), ("cm_depth", C.c_uint32), ("cm_log2_width", C.c_uint32), ("hll_p", C.c_uint32), ("staging_records", C.c_uint64),
                ("n_shards", C.c_uint32), ("shard_id", C.c_uint32), ("profile", C.c_uint32), ("ingest_variant", C.c_uint32), ("ext_sketch", C.c_void_p * 4),
                ("copy_threads", C.c_uint32), ("group_flags", C.c_uint32), ("local_fold", C.c_uint32)]


n = 2_000_000
th = O.zipf_thresholds(1_000_000, 1.1)
recs = O.gen_stream(n, seed=2, n_keys=1_000_000, thresholds=th)
cfg = Config(); cfg.struct_size = C.sizeof(Config); cfg.max_entries = 5000
h = C.c_void_p()
assert lib.nfagg_create(C.byref(cfg), C.byref(h)) == 0
out = np.zeros(8192, dtype=O.FLOW_RECORD)
base = recs.ctypes.data
for rep in range(3):
    off, ev, t_in, t_ev, calls = 0, 0, 0.0, 0.0, 0
    consumed, n_out = C.c_size_t(0), C.c_size_t(0)
    t00 = time.perf_counter()
    while off < n:
        t0 = time.perf_counter()
        rc = lib.nfagg_ingest(h, C.c_void_p(base + off * 144), n - off, C.byref(consumed))
        t1 = time.perf_counter()
        t_in += t1 - t0; calls += 1
        off += consumed.value
        if rc == 1:
            lib.nfagg_evict(h, 1, out.ctypes.data_as(C.c_void_p), 8192, C.byref(n_out))
            t_ev += time.perf_counter() - t1; ev += 1
    lib.nfagg_evict(h, 2, out.ctypes.data_as(C.c_void_p), 8192, C.byref(n_out))
    dt = time.perf_counter() - t00
    print("rep %d: %.1f M rec/s, %d epochs, ingest %.0f us/epoch (%d calls), evict %.0f us/epoch" % (rep, n / dt / 1e6, ev, t_in / ev * 1e6, calls, t_ev / ev * 1e6))
lib.nfagg_destroy(h)
