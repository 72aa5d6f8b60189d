#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer ingest paths (never the bench `value`): nfagg_ingest from pageable host
memory (memcpy into the pinned ring + hipMemcpyAsync + fold) and staging acquire/commit with the buffer already filled."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth

n, flows = 20_000_000, 1_000_000
th = synth.zipf_thresholds(flows, 1.1)
recs = synth.stream_host(n, seed=2, n_keys=flows, thresholds=th)
for staging, threads in ((1 << 20, 1), (1 << 20, 0), (1 << 20, 4), (1 << 20, 8), (1 << 20, 16), (1 << 22, 0), (1 << 22, 8)):      # 0 = what the copy workers calibration found best
    with nf.FlowTable(max_entries=1 << 26, staging_records=staging, copy_threads=threads) as tab:
        tab.ingest(recs[: 2 * staging]); tab.evict()
        t0 = time.perf_counter()
        rc, c = tab.ingest(recs)
        tab.sync()
        dt = time.perf_counter() - t0
        print(f"nfagg_ingest (pageable host buffer, staging {staging} records, {threads} copy thread(s)): {n / dt / 1e6:.1f} M records/s = {n * 144 / dt / 1e9:.1f} GB/s")
        tab.evict()
        if threads != 1:
            continue
        # staging acquire/commit: fill cost excluded (the ring reader would write straight into the pinned buffer)
        buf = tab.staging_acquire(); m = len(buf); buf[:] = recs[:m]; tab.staging_commit(m)
        t0 = time.perf_counter(); done = 0
        while done < n - m:
            buf = tab.staging_acquire()
            rc, c = tab.staging_commit(m)          # contents are whatever the buffer holds: same bytes each time
            done += m
        tab.sync()
        dt = time.perf_counter() - t0
        print(f"staging acquire/commit (pinned, pre-filled, {m} records per commit): {done / dt / 1e6:.1f} M records/s = {done * 144 / dt / 1e9:.1f} GB/s")
