#!/usr/bin/env python3
"""Test infrastructure: a diag-only ingest variant of lib/libnfagg_diag.so (NFAGG_LIB must point at it) against the oracle on seeded
streams large enough for the two-pass fold — for experiments whose results are meant to be RIGHT (variant 24: wave-level duplicate
combining, variant 28: records two tiles ahead). Usage: NFAGG_LIB=.../libnfagg_diag.so python tests/tools/diag_variant_parity.py VARIANT"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import netobserv_ebpf_agent_amd as nf
from oracle import oracle as O
from conftest import assert_records_equal

variant = int(sys.argv[1])
for seed, n, keys, hot, sv in ((1, 3_000_000, 200_000, 0, 1), (2, 2_000_000, 5_000, 700, 1), (3, 4_000_000, 1_000_000, 0, 0), (4, 1_500_000, 40, 0, 1)):
    recs = O.gen_stream(n, seed=seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), variant=sv, hot_permille=hot)
    want = O.run_accounter(recs, 1 << 21)[0][1]
    with nf.FlowTable(max_entries=1 << 21, ingest_variant=variant) as tab:
        assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, n)
        got = nf.sort_by_key(tab.evict(nf.REASON_CLOSING))
    assert_records_equal(got, want, f"variant {variant}, seed {seed}")
    print(f"variant {variant}: {n} records over {keys} flows (hot {hot} permille): {len(want)} flows bit-exact vs the oracle")
