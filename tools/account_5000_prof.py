#!/usr/bin/env python3
"""The device-resident nfagg_account_device leg of bench.py's extra.cache_max_flows_5000 on its own (8 M records of the configs[1]
stream, CACHE_MAX_FLOWS = 5000, the evict-on-full loop on the device), for the rocprofv3 passes of tools/profile_bench.sh
(PROF_PROG="python tools/account_5000_prof.py"): prints one bench-shaped JSON line. usage: account_5000_prof.py [--steps K] [--variant V] [--max-entries M] [--chunk RECORDS_PER_CALL]"""
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

steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
variant = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 0     # 30: the kernel chain always (the fallback)
M = int(sys.argv[sys.argv.index("--max-entries") + 1]) if "--max-entries" in sys.argv else 5000  # CACHE_MAX_FLOWS (round 6: 10 000, 100 000 too)
chunk = int(sys.argv[sys.argv.index("--chunk") + 1]) if "--chunk" in sys.argv else 0            # records per call (0: the whole stream in one)
n, keys = 8_000_000, 1_000_000
th = synth.zipf_thresholds(keys, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
d_ev = torch.empty((n + 8192) * 144, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr())
torch.cuda.synchronize()
ends_cap = n // M + 16
d_close = torch.empty((M + 8192) * 144, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
with nf.FlowTable(max_entries=M, ingest_variant=variant) as tab:
    t0 = time.perf_counter()
    flows = evs = 0
    per_step = []
    for _ in range(steps):
        t_s = time.perf_counter()
        for lo in range(0, n, chunk or n):
            m = min(chunk or n, n - lo)
            rc, c, ends = tab.account_device(d.data_ptr() + lo * 144, m, d_ev.data_ptr(), n + 8192, ends_cap)
            assert rc == nf.OK and c == m
            flows += ends[-1] if ends else 0
            evs += len(ends)
        flows += tab.evict_device(d_close.data_ptr(), M + 8192, nf.REASON_CLOSING)
        evs += 1
        per_step.append((time.perf_counter() - t_s) * 1e3)
    dt = time.perf_counter() - t0
print(json.dumps({"config": {"workload": "extra.cache_max_flows_%d.account_device_resident: %d M records, CACHE_MAX_FLOWS %d" % (M, n // 1_000_000, M),
                             "hot_permille": 0, "stream_variant": 0, "mode": "accounter", "max_entries": M, "ingest_variant": variant, "chunk": chunk,
                             "evicted_flows_per_step": flows // steps},
                  "roofline": {"launches": steps, "records_per_launch": n}, "ms_per_call": round(dt / steps * 1e3, 3), "ms_best": round(min(per_step), 3), "ms_median": round(sorted(per_step)[len(per_step) // 2], 3), "evictions_per_call": evs // steps}))
