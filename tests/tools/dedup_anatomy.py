"""Experiment (round 4): anatomy of a kernel-dedup parity failure. Raw ctypes on the library named by NFAGG_LIB (so that an
older build without the newest symbols can be driven too). usage: dedup_anatomy.py <variant> <style> [local_fold]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (one HIP runtime per process: torch's)
from oracle import oracle as O
from conftest import dedup_stream

lib = C.CDLL(os.environ["NFAGG_LIB"])


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_entries", C.c_uint64), ("table_log2_slots", C.c_uint32), ("mode", C.c_uint32),
                ("sketch_flags", C.c_uint32), ("cm_depth", C.c_uint32), ("cm_log2_width", C.c_uint32), ("hll_p", C.c_uint32), ("staging_records", C.c_uint64),
                ("n_shards", C.c_uint32), ("shard_id", C.c_uint32), ("profile", C.c_uint32), ("ingest_variant", C.c_uint32), ("ext_sketch", C.c_void_p * 4),
                ("copy_threads", C.c_uint32), ("group_flags", C.c_uint32), ("local_fold", C.c_uint32)]


variant, style = int(sys.argv[1]), int(sys.argv[2])
local_fold = len(sys.argv) > 3
lib.nfagg_last_error.restype = C.c_char_p
for rep in range(3):
    th = O.zipf_thresholds(300, 1.1)
    recs = dedup_stream(O, 40000, seed=200 + style + rep, n_keys=300, thresholds=th, style=style)
    want = O.run_accounter(recs, 1 << 22, mode=1)[0][1]
    cfg = Config()
    cfg.struct_size = C.sizeof(Config)
    cfg.max_entries = 1 << 16
    cfg.mode = 1
    cfg.ingest_variant = variant
    cfg.local_fold = 1 if local_fold else 0
    h = C.c_void_p()
    rc = lib.nfagg_create(C.byref(cfg), C.byref(h))
    assert rc == 0, (rc, lib.nfagg_last_error(None))
    consumed = C.c_size_t(0)
    rc = lib.nfagg_ingest(h, recs.ctypes.data_as(C.c_void_p), len(recs), C.byref(consumed))
    assert rc == 0 and consumed.value == len(recs), (rc, lib.nfagg_last_error(h))
    ln = C.c_uint64(0)
    lib.nfagg_len(h, C.byref(ln))
    out = np.zeros(1 << 16, dtype=O.FLOW_RECORD)
    n = C.c_size_t(0)
    rc = lib.nfagg_evict(h, 2, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n))
    got = out[: n.value]
    keys = np.ascontiguousarray(got["id"]).view(np.uint8).reshape(len(got), 40)
    uniq, inv, cnt = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
    raw = got.view(np.uint8).reshape(len(got), 144)
    ident = 0
    for k in np.nonzero(cnt > 1)[0]:
        rows = raw[inv == k]
        ident += int(all((rows[0] == r).all() for r in rows[1:]))
    wkeys = {bytes(k) for k in np.ascontiguousarray(want["id"]).view(np.uint8).reshape(len(want), 40)}
    foreign = sum(1 for k in uniq if bytes(k) not in wkeys)
    print("rep %d rc %d len %d evicted %d want %d unique keys %d keys with duplicates %d (identical copies: %d) keys not in the stream %d err %s"
          % (rep, rc, ln.value, n.value, len(want), len(uniq), int((cnt > 1).sum()), ident, foreign, lib.nfagg_last_error(h)))
    if foreign and rep == 0:
        wk = np.ascontiguousarray(want["id"]).view(np.uint64).reshape(len(want), 5)
        shown = 0
        for r in got:
            kb = np.ascontiguousarray(r["id"]).view(np.uint8).tobytes()
            if kb in wkeys:
                continue
            kw = np.frombuffer(kb, dtype=np.uint64)
            match = [(kw[j] == wk[:, j]) for j in range(5)]
            best = np.argmax(sum(m.astype(int) for m in match))
            print("   foreign key words", [hex(int(x)) for x in kw], "best stream key matches words", [int(m[best]) for m in match],
                  "other words found in stream at col:", [int((kw[j] == wk[:, j]).any()) for j in range(5)],
                  "bytes %d packets %d ifx %d" % (r["metrics"]["bytes"], r["metrics"]["packets"], r["metrics"]["if_index_first_seen"]))
            shown += 1
            if shown >= 6:
                break
    if len(uniq) == len(want) == len(got):
        a = got[np.lexsort(keys.T[::-1])]
        print("   bit-exact:", np.array_equal(np.sort(raw.view("V144").reshape(-1)), np.sort(want.view(np.uint8).reshape(len(want), 144).view("V144").reshape(-1))))
    lib.nfagg_destroy(h)
