#!/usr/bin/env python3
"""Test infrastructure: nfagg_account on its default path (calls of more than a few epochs: epochs found first, csrc/nfagg_epoch_par.hip;
the others: the kernel chain) with fresh seeds for a time budget — table sizes from 1 to 20 000 entries, ragged calls, hot flows, the
sketches fed along, every eviction against the oracle. Usage: python tests/tools/soak_account_par.py [seconds] [first seed] [--large]
--large (round 6): the table sizes beyond the kernel chain's 32 768 — 10 000 ... 250 000 entries (scripts/agent.yml:35-36,
pkg/flow/tracer_map_bench_test.go:64-111), streams of 2-6 M records, hot flows (segments of tens of thousands of records: folded in chunks)."""
import os, sys, time, traceback
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import netobserv_ebpf_agent_amd as nf
from oracle import oracle as O
from test_account_gpu import _check, _stream

large = "--large" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--large"]
budget = float(argv[0]) if len(argv) > 0 else 60.0
seed = int(argv[1]) if len(argv) > 1 else 100
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
    if large:
        max_entries = int(rng.choice([10_000, 40_000, 100_000, 100_000, 250_000]))
        keys = int(rng.choice([max_entries + 1, 3 * max_entries + 10, 1_000_000, 3_000_000]))
        n = int(rng.choice([2_000_000, 4_000_000, 6_000_000]))
    recs = _stream(O, n, keys, seed=seed, hot=int(rng.choice([0, 0, 500, 950])), variant=int(rng.choice([0, 1])))
    batches = [int(rng.choice([1, 777, 70_000, 150_000, 400_000, 1 << 30])) for _ in range(600)]
    if large:
        batches = [int(rng.choice([131_071, 131_072, 500_000, 1_048_576, 3_000_000, 1 << 30])) for _ in range(600)]
    desc = dict(seed=seed, max_entries=max_entries, keys=keys, n=n)
    try:
        sk = bool(rng.integers(0, 2))
        with nf.FlowTable(max_entries=max_entries, staging_records=int(rng.choice([0, 1 << 18, 1 << 21])),
                          sketches=(nf.SKETCH_CM | nf.SKETCH_HLL) if sk else 0, cm_log2_width=12, hll_p=8) as tab:
            _check(nf, O, tab, recs, max_entries, batches)
            if sk:
                cs, cd, hs, hd = O.sketches(recs, 4, 12, 8)
                assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cs) and np.array_equal(tab.sketch_snapshot(nf.CM_DST), cd)
                assert np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs) and np.array_equal(tab.sketch_snapshot(nf.HLL_DST), hd)
    except Exception:
        print("FAILED:", desc, flush=True)
        traceback.print_exc()
        sys.exit(1)
    runs += 1; recs_total += n
print(f"soak_account_par ok: {runs} streams, {recs_total} records, every eviction bit-exact vs the oracle (seeds up to {seed})")
