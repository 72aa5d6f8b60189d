#!/usr/bin/env python3
"""Where should nfagg_account's epochs-found-first path begin? Consecutive calls of n records (CACHE_MAX_FLOWS 5000, configs[1] stream),
device-resident and from page-locked buffers, with the entry bar as the library has it and lowered (libnfagg_diag.so:
NFAGG_DIAG_PAR_MIN). usage: NFAGG_LIB=.../libnfagg_diag.so [NFAGG_DIAG_PAR_MIN=N] python tests/tools/par_entry_bar.py"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth
M, total = 5000, 4_000_000
th = synth.zipf_thresholds(1_000_000, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(total * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
synth.stream_device(d.data_ptr(), total, seed=2, n_keys=1_000_000, d_thresholds=d_th.data_ptr()); torch.cuda.synchronize()
pin = nf.PinnedRecords(total); pin.records[:] = d.cpu().numpy().view(nf.FLOW_RECORD)
pout = nf.PinnedRecords(total // 2 + 2 * M + 8192)
d_ev = torch.empty((total // 2 + 2 * M + 8192) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
res = {"par_min": os.environ.get("NFAGG_DIAG_PAR_MIN", "default")}
for n in (16384, 32768, 49152, 65536, 98304, 131072):
    calls = min(200, total // n)
    row = {}
    for leg in ("device", "page_locked"):
        with nf.FlowTable(max_entries=M) as tab:
            def call(k):
                if leg == "device":
                    rc, c, ends = tab.account_device(d.data_ptr() + k * n * 144, n, d_ev.data_ptr(), total // 2 + 2 * M + 8192, n // M + 4)
                else:
                    rc, c, ends = tab.account(pin.records[k * n:(k + 1) * n], out=pout.records, max_epochs=n // M + 4)
                assert rc == nf.OK and c == n
            for k in range(3): call(k)
            ts = []
            for k in range(calls):
                t0 = time.perf_counter(); call(k); ts.append(time.perf_counter() - t0)
            st = tab.stats()
        ts.sort()
        row[leg] = {"us_median": round(ts[len(ts) // 2] * 1e6, 1), "Mrec_s": round(n / ts[len(ts) // 2] / 1e6, 1), "par": int(st.account_epochs_first), "chain": int(st.account_chain)}
    res[str(n)] = row
print(json.dumps(res))
