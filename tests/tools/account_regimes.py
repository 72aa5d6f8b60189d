#!/usr/bin/env python3
"""nfagg_account in the regimes the reference's own seam and benchmarks produce (round-5 review, items 2a / 2b):

  small calls   Accounter.Account receives ONE record per channel operation through a channel of BUFFERS_LENGTH = 50
                (pkg/agent/agent.go:408, pkg/config/config.go:134, pkg/flow/tracer_ringbuf.go:112-134): a shim that drains what is
                queued calls nfagg_account with 1 ... a few thousand records. Consecutive calls of n records each over the configs[1]
                stream, CACHE_MAX_FLOWS 5000, from a page-locked and from a pageable buffer: us per call, records/s — with the
                one-core oracle Accounter on the same calls beside it (the crossover).
  table sizes   CACHE_MAX_FLOWS 5000 / 10 000 / 100 000 (pkg/flow/tracer_map_bench_test.go:64-111 brackets 1 k / 10 k / 100 k;
                scripts/agent.yml:35-36 deploys 10 000): 8 M records per call, device-resident / page-locked / pageable.

usage: account_regimes.py [--small] [--tables] [--entries 5000,10000,100000] [--sizes 1,64,...] [--records N] [--no-oracle]
Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def stream(n, keys=1_000_000):
    th = synth.zipf_thresholds(keys, 1.1)
    d_th = torch.from_numpy(th.view(np.int64)).cuda()
    d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr())
    torch.cuda.synchronize()
    return d


def small_calls(sizes, M, with_oracle, total_cap=4_000_000, max_calls=2000):
    """Consecutive calls of n records each (the table's state carries over from call to call, as in the agent)."""
    res = {}
    d = stream(total_cap)
    host = d.cpu().numpy().view(nf.FLOW_RECORD)
    pin_in = nf.PinnedRecords(total_cap)
    pin_in.records[:] = host
    pin_ev = nf.PinnedRecords(total_cap // 2 + 2 * M + 8192)
    h_ev = np.empty(total_cap // 2 + 2 * M + 8192, dtype=nf.FLOW_RECORD)
    h_ev.view(np.uint8)[::4096] = 0
    O = None
    if with_oracle:
        from oracle import oracle as O_
        O_.build()
        O = O_
    for n in sizes:
        calls = max(3, min(max_calls, total_cap // n))
        row = {"calls": calls}
        for leg in ("page_locked", "pageable"):
            src = pin_in.records if leg == "page_locked" else host
            out = pin_ev.records if leg == "page_locked" else h_ev
            with nf.FlowTable(max_entries=M) as tab:
                ts, evs = [], 0
                # warm-up: the first calls allocate (staging ring, scratch)
                for k in range(min(3, calls)):
                    tab.account(src[k * n:(k + 1) * n], out=out, max_epochs=n // M + 4)
                tab.evict(nf.REASON_CLOSING, cap=max(8192, M))
                t_all0 = time.perf_counter()
                for k in range(calls):
                    t0 = time.perf_counter()
                    rc, c, epochs = tab.account(src[k * n:(k + 1) * n], out=out, max_epochs=n // M + 4)
                    ts.append(time.perf_counter() - t0)
                    assert rc == nf.OK and c == n, (rc, c)
                    evs += len(epochs)
                t_all = time.perf_counter() - t_all0
                st = tab.stats()
            ts.sort()
            row[leg] = {"us_per_call_median": round(ts[len(ts) // 2] * 1e6, 1), "us_per_call_p10": round(ts[len(ts) // 10] * 1e6, 1),
                        "us_per_call_p90": round(ts[(len(ts) * 9) // 10] * 1e6, 1),
                        "Mrecords_per_s": round(n * calls / t_all / 1e6, 3), "evictions": evs,
                        "paths": {"epochs_first": int(st.account_epochs_first), "chain": int(st.account_chain), "declined": int(st.account_declined)}}
        if O is not None:
            acc = O.Accounter(M, 0)
            raw = host.view(np.uint8).reshape(-1)
            ts = []
            t_all0 = time.perf_counter()
            for k in range(calls):
                t0 = time.perf_counter()
                off, end = k * n, (k + 1) * n
                while off < end:
                    off += acc.ingest(raw[off * 144:end * 144])
                    if off < end:
                        acc.evict()
                ts.append(time.perf_counter() - t0)
            t_all = time.perf_counter() - t_all0
            acc.close()
            ts.sort()
            row["oracle_1_core"] = {"us_per_call_median": round(ts[len(ts) // 2] * 1e6, 1), "Mrecords_per_s": round(n * calls / t_all / 1e6, 3)}
        res[str(n)] = row
    pin_in.close(); pin_ev.close()
    return res


def tables(entries, n, reps=3):
    res = {}
    d = stream(n)
    host = d.cpu().numpy().view(nf.FLOW_RECORD)
    pin_in = nf.PinnedRecords(n)
    pin_in.records[:] = host
    for M in entries:
        d_ev = torch.empty((n + 2 * M + 8192) * 144, dtype=torch.uint8, device="cuda")
        pin_ev = nf.PinnedRecords(n // 2 + 2 * M + 8192)
        h_ev = np.empty(n // 2 + 2 * M + 8192, dtype=nf.FLOW_RECORD)
        h_ev.view(np.uint8)[::4096] = 0
        ends_cap = n // M + 16
        pin_close = nf.PinnedRecords(max(8192, M))
        h_close = np.empty(max(8192, M), dtype=nf.FLOW_RECORD)
        h_close.view(np.uint8)[::4096] = 0
        close_buf = {"device": pin_close.records, "page_locked": pin_close.records, "pageable": h_close}
        row = {}
        with nf.FlowTable(max_entries=M) as tab:
            def call(leg):
                if leg == "device":
                    rc, c, ends = tab.account_device(d.data_ptr(), n, d_ev.data_ptr(), n + 2 * M + 8192, ends_cap)
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
                row[leg] = {"ms_best": round(min(ts) * 1e3, 2), "ms_all": [round(t * 1e3, 2) for t in ts],
                            "Mrecords_per_s": round(n / min(ts) / 1e6, 1), "evictions": evs, "evicted_flows": int(flows)}
            st = tab.stats()
            row["paths"] = {"epochs_first": int(st.account_epochs_first), "chain": int(st.account_chain), "declined": int(st.account_declined)}
        res[str(M)] = row
        del d_ev
        pin_ev.close(); pin_close.close()
    pin_in.close()
    return res


if __name__ == "__main__":
    out = {}
    both = "--small" not in sys.argv and "--tables" not in sys.argv
    if "--tables" in sys.argv or both:
        entries = [int(x) for x in arg("--entries", "5000,10000,100000").split(",")]
        out["tables"] = tables(entries, int(arg("--records", "8000000")))
    if "--small" in sys.argv or both:
        sizes = [int(x) for x in arg("--sizes", "1,64,1024,16384,65536,262144,1048576").split(",")]
        out["small_calls"] = small_calls(sizes, int(arg("--small-entries", "5000")), "--no-oracle" not in sys.argv)
    print(json.dumps(out))
