#!/usr/bin/env python3
"""Diagnostics: where does the partition pass of the kernel-dedup fold (csrc/nfagg_dedup_cached.hip k_dedup_parts) spend its
time? Runs the bench's dedup stream through libnfagg_diag.so with ingest_variant 10 (the product path) and the ablations
13 (no flush), 14 (no flush, no fold into the entry), 15 (gather + decode only); results of 13..15 are wrong by construction.
Run on the GPU box under rocprofv3 --kernel-trace --stats (the per-kernel averages are what is read), or alone (wall times).
    python tools/dedup_ablation.py [flows] [records] [hot_permille]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NFAGG_LIB", os.path.join(ROOT, "netobserv-ebpf-agent_amd", "lib", "libnfagg_diag.so"))
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth

flows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
hot = int(sys.argv[3]) if len(sys.argv) > 3 else 0
th = synth.zipf_thresholds(flows, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=flows, d_thresholds=d_th.data_ptr(), hot_permille=hot, variant=2)
torch.cuda.synchronize()
out = torch.empty((flows + 4096) * 144, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for variant in (10, 13, 14, 15):
    tab = nf.FlowTable(max_entries=1 << 21, mode=nf.MODE_KERNEL_DEDUP, ingest_variant=variant)
    ms = []
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tab.ingest_device(d.data_ptr(), n)
        len(tab)                                    # synchronises on the library's stream
        ms.append((time.perf_counter() - t0) * 1e3)
        tab.evict_device(out.data_ptr(), flows + 4096)
    st = tab.stats()
    ph = (C.c_uint64 * 8)()
    nf._lib.lib.nfagg_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    nf._lib.lib.nfagg_debug_phase_cycles(tab._h, ph)
    print("variant %d: ingest call %.3f ms (min of 3), bypassed %.3f of the records, %d items per call through the overflow list"
          % (variant, min(ms), st.records_bypassed / (3 * n), ph[7] // 3), flush=True)
    tab.close()
