#!/usr/bin/env python3
"""nfagg_account on the reference's default CACHE_MAX_FLOWS = 5000 (pkg/config/config.go:146), 8 M records of the configs[1] stream:
device-resident, from a pageable host buffer, from page-locked host buffers — the three numbers of bench.py's
extra.cache_max_flows_5000, on their own (seconds instead of a whole bench run). usage: account_paths_bench.py [--variant V] [--reps R]
[--staging S] [--sketches]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


variant, reps, staging = arg("--variant", 0), arg("--reps", 3), arg("--staging", 0)
n, keys, M = arg("--records", 8_000_000), 1_000_000, arg("--max-entries", 5000)
sk = (nf.SKETCH_CM | nf.SKETCH_HLL) if "--sketches" in sys.argv else 0
th = synth.zipf_thresholds(keys, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
d_ev = torch.empty((n + 8192) * 144, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr())
torch.cuda.synchronize()
host = d.cpu().numpy().view(nf.FLOW_RECORD)
pin_in, pin_ev = nf.PinnedRecords(n), nf.PinnedRecords(n // 2 + 8192)
pin_in.records[:] = host
h_ev = np.empty(n // 2 + 8192, dtype=nf.FLOW_RECORD)
h_ev.view(np.uint8)[::4096] = 0
ends_cap = n // M + 16
pin_close = nf.PinnedRecords(max(8192, M))
h_close = np.empty(max(8192, M), dtype=nf.FLOW_RECORD)
h_close.view(np.uint8)[::4096] = 0
close_buf = {"device": pin_close.records, "page_locked": pin_close.records, "pageable": h_close}
res = {"variant": variant, "records": n, "max_entries": M, "staging_records": staging, "sketches": bool(sk)}
with nf.FlowTable(max_entries=M, ingest_variant=variant, staging_records=staging, sketches=sk) as tab:
    def call(leg):
        if leg == "device":
            rc, c, ends = tab.account_device(d.data_ptr(), n, d_ev.data_ptr(), n + 8192, ends_cap)
            n_ep, flows = len(ends), (ends[-1] if ends else 0)
        else:
            rc, c, epochs = (tab.account(host, out=h_ev, max_epochs=ends_cap) if leg == "pageable" else
                             tab.account(pin_in.records, out=pin_ev.records, max_epochs=ends_cap))
            n_ep, flows = len(epochs), sum(len(e) for e in epochs)
        assert rc == nf.OK and c == n, (rc, c)
        # (the closing eviction into a buffer the caller keeps, as the evictions on full: a fresh 14 MB array per call is page faults, not the library)
        flows += len(tab.evict(nf.REASON_CLOSING, out=close_buf[leg]))
        return n_ep + 1, flows
    for leg in ("device", "page_locked", "pageable"):
        call(leg)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            evs, flows = call(leg)
            ts.append(time.perf_counter() - t0)
        best = min(ts)
        res[leg] = {"ms_best": round(best * 1e3, 2), "ms_all": [round(t * 1e3, 2) for t in ts], "Mrecords_per_s": round(n / best / 1e6, 1),
                    "evictions": evs, "evicted_flows": int(flows)}
pin_in.close(); pin_ev.close()
print(json.dumps(res))
