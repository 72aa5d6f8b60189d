#!/usr/bin/env python3
"""Test infrastructure: one-key kernel-dedup streams through every dedup routing, against the oracle (found by the soak)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import netobserv_ebpf_agent_amd as nf
from oracle import oracle as O

for keys in (1, 2):
    for n in (1_500_000, 3_500_000, 5_000_000):
        recs = O.gen_stream(n, seed=918135167, n_keys=keys, variant=1)
        want = None
        for iv in (0, 1, 10, 16):
            for me in (2, 1 << 20):
                if me < keys:
                    continue
                try:
                    with nf.FlowTable(max_entries=me, mode=nf.MODE_KERNEL_DEDUP, ingest_variant=iv, staging_records=1 << 23) as tab:
                        rc, c = tab.ingest(recs)
                        got = nf.sort_by_key(tab.evict(nf.REASON_CLOSING))
                        st = tab.stats()
                    if want is None:
                        want = O.run_accounter(recs, 1 << 20, 1)[0][1]
                    ok = (rc, c) == (nf.OK, n) and got.tobytes() == want.tobytes()
                    print("keys", keys, "n", n, "iv", iv, "me", me, "ok" if ok else "DIFF", flush=True)
                except Exception as e:
                    print("keys", keys, "n", n, "iv", iv, "me", me, "ERR", str(e)[:60], flush=True)
