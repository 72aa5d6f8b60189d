#!/usr/bin/env python3
"""Small CACHE_MAX_FLOWS (the reference's default is 5000, pkg/config/config.go:146): every few thousand records the map is
full and the caller evicts. Rates of the device-resident loop (nfagg_ingest_device / nfagg_evict_device) and of the HOST path
a cgo shim uses (nfagg_ingest from a pageable buffer / nfagg_evict into host memory). Usage: [NFAGG_LIB=...] python
tools/small_table_bench.py [max_entries ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth

sizes = [int(x) for x in sys.argv[1:]] or [5000, 100000]
keys = 1_000_000
th = synth.zipf_thresholds(keys, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
for me in sizes:
    n = 4_000_000 if me < 50_000 else 20_000_000
    d = torch.empty(n * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr()); torch.cuda.synchronize()
    host = d.cpu().numpy().view(nf.FLOW_RECORD)
    out = torch.empty((me + 16) * 144, dtype=torch.uint8, device="cuda")
    with nf.FlowTable(max_entries=me) as tab:
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter(); off = ev = fl = 0
            while off < n:
                rc, c = tab.ingest_device(d.data_ptr() + off * 144, n - off); off += c
                if rc == nf.FULL:
                    fl += tab.evict_device(out.data_ptr(), me + 16, nf.REASON_FULL); ev += 1
            fl += tab.evict_device(out.data_ptr(), me + 16, nf.REASON_CLOSING); ev += 1
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"max_entries {me}: device-resident {n / dt / 1e6:8.1f} M records/s  ({ev} evictions, {fl} flows, {dt / ev * 1e6:.0f} us per epoch)")
    with nf.FlowTable(max_entries=me) as tab:
        for rep in range(2):
            t0 = time.perf_counter(); off = ev = fl = 0
            while off < n:
                rc, c = tab.ingest(host[off:]); off += c
                if rc == nf.FULL:
                    fl += len(tab.evict(nf.REASON_FULL, cap=me + 16)); ev += 1
            fl += len(tab.evict(nf.REASON_CLOSING, cap=me + 16)); ev += 1
            dt = time.perf_counter() - t0
        print(f"max_entries {me}: host buffers   {n / dt / 1e6:8.1f} M records/s  ({ev} evictions, {fl} flows, {dt / ev * 1e6:.0f} us per epoch)")
