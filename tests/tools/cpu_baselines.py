#!/usr/bin/env python3
"""Test infrastructure: the CPU baselines of bench.py's cpu_baseline on their own (oracle/): one core, partition-then-fold, local fold
then key-sharded merge, over the first 20 M records of the configs[1] stream (host generator = the device generator)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
n, keys = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 1_000_000
recs = O.gen_stream(n, seed=2, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1))
print("cores", os.cpu_count(), "records", n)
acc = O.Accounter(1 << 21, 0); t = time.perf_counter(); acc.ingest(recs); ev = acc.evict(); dt = time.perf_counter() - t
print("one core: %.2f s  %.1f M records/s  (%d flows)" % (dt, n / dt / 1e6, len(ev)))
for T in (16, 64, 128, 256):
    if T > 2 * (os.cpu_count() or 1):
        continue
    f, fl, ps, fs, big = O.partition_fold_mt(recs, T, 1 << 21)
    print("partition-then-fold, %3d threads: %.3f + %.3f s  %.1f M records/s  largest shard %.3f" % (T, ps, fs, n / (ps + fs) / 1e6, big))
    f, fl, a, b, share = O.local_fold_mt(recs, T, 1 << 21)
    assert f == n and fl == len(ev)
    print("local fold + merge,  %3d threads: %.3f + %.3f s  %.1f M records/s  largest merge shard %.3f" % (T, a, b, n / (a + b) / 1e6, share))
