#!/usr/bin/env python3
"""Seeded parity streams for a host that HAS a Go toolchain (tools/go/accounter_parity_test.go): `write DIR` produces the
streams (the scrambled generator of the parity tests: every order-dependent field of model.AccumulateBase varies), `compare DIR`
folds them through libnfagg (nfagg_account: needs an MI355X) and through the CPU oracle and compares both, eviction by
eviction and bit for bit, with the dump of the reference's own Accounter.Account (evictions_<k>.bin) when it is there.
Test infrastructure (lives under tests/: it uses the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [  # (records, keys, seed, hot permille, max_entries)
    (200_000, 20_000, 101, 0, 1 << 20), (200_000, 20_000, 102, 500, 3_000), (60_000, 500, 103, 0, 100), (300_000, 100_000, 104, 0, 5_000),
]


def main():
    cmd, d = sys.argv[1], sys.argv[2]
    from oracle import oracle as O
    if cmd == "write":
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "manifest.txt"), "w") as mf:
            for k, (n, keys, seed, hot, me) in enumerate(CASES):
                recs = O.gen_stream(n, seed=seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=hot, variant=1)
                recs.tofile(os.path.join(d, "stream_%d.bin" % k))
                mf.write("%d %d %d\n" % (k, n, me))
        print("wrote %d streams to %s" % (len(CASES), d))
        return 0
    import netobserv_ebpf_agent_amd as nf
    bad = 0
    for k, (n, keys, seed, hot, me) in enumerate(CASES):
        recs = np.fromfile(os.path.join(d, "stream_%d.bin" % k), dtype=O.FLOW_RECORD)
        want = [np.asarray(e) for _, e in O.run_accounter(recs, me)]
        with nf.FlowTable(max_entries=me) as tab:
            rc, c, epochs = tab.account(recs.view(nf.FLOW_RECORD))
            assert rc == nf.OK and c == n
            got = [nf.sort_by_key(e) for e in epochs] + [nf.sort_by_key(tab.evict(nf.REASON_CLOSING))]
        ok = len(got) == len(want) and all(g.tobytes() == w.tobytes() for g, w in zip(got, want))
        line = "stream %d: libnfagg vs oracle %s (%d evictions)" % (k, "IDENTICAL" if ok else "DIFFER", len(want))
        bad += not ok
        ref = os.path.join(d, "evictions_%d.bin" % k)
        if os.path.exists(ref):
            raw = open(ref, "rb").read()
            at, refev = 0, []
            while at < len(raw):
                cnt = int(np.frombuffer(raw, dtype="<u4", count=1, offset=at)[0]); at += 4
                refev.append(nf.sort_by_key(np.frombuffer(raw, dtype=nf.FLOW_RECORD, count=cnt, offset=at).copy())); at += cnt * 144
            same = len(refev) == len(got) and all(g.tobytes() == r.tobytes() for g, r in zip(got, refev))
            line += "; vs the reference's Accounter.Account dump: %s" % ("IDENTICAL" if same else "DIFFER")
            bad += not same
        else:
            line += "; no reference dump (run tools/go/accounter_parity_test.go where Go is available): parity UNPINNED for the order-dependent fields"
        print(line)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
