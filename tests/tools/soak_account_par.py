#!/usr/bin/env python3
"""Test infrastructure: ingest_variant 31 (the epoch-parallel evict-on-full loop) with fresh seeds for a time budget — table sizes
from 1 to 20 000 entries, ragged calls, hot flows, every eviction against the oracle. Usage: python tests/tools/soak_account_par.py [seconds] [first seed]"""
import os, sys, time, traceback
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import netobserv_ebpf_agent_amd as nf
from oracle import oracle as O
from test_account_gpu import _check, _stream

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t_end = time.time() + budget
runs = recs_total = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    max_entries = int(rng.choice([1, 2, 7, 64, 300, 1000, 5000, 20_000]))
    keys = int(rng.choice([max(2, max_entries // 2), max_entries + 1, 3 * max_entries + 10, 50_000, 400_000]))
    n = int(rng.choice([120_000, 400_000, 1_200_000]))
    if max_entries <= 7:
        n = min(n, 200_000)                          # tens of thousands of evictions: bound the oracle's time
    recs = _stream(O, n, keys, seed=seed, hot=int(rng.choice([0, 0, 500, 950])), variant=int(rng.choice([0, 1])))
    batches = [int(rng.choice([1, 777, 70_000, 150_000, 400_000, 1 << 30])) for _ in range(600)]
    desc = dict(seed=seed, max_entries=max_entries, keys=keys, n=n)
    try:
        with nf.FlowTable(max_entries=max_entries, ingest_variant=31, staging_records=int(rng.choice([0, 1 << 18, 1 << 21]))) as tab:
            _check(nf, O, tab, recs, max_entries, batches)
    except Exception:
        print("FAILED:", desc, flush=True)
        traceback.print_exc()
        sys.exit(1)
    runs += 1; recs_total += n
print(f"soak_account_par ok: {runs} streams, {recs_total} records, every eviction bit-exact vs the oracle (seeds up to {seed})")
