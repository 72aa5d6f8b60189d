#!/usr/bin/env python3
"""libnfagg_diag.so only: k_finalize + k_evict with the first record's identity words in ONE DENSE 64-byte unit per live-list position
(sequential) instead of the slot's cold half line (random; its other half is a neighbour's) — the layout the round 2-5 reviews asked
for, as a timing experiment (evicted MAC high words are wrong while it is on). 100 M records over 10 M flows (the flows_10m leg: one
rank of configs[3]) and over 1 M flows (configs[1]); rocprofv3 --kernel-trace --stats gives the kernels' times.
usage: NFAGG_LIB=.../libnfagg_diag.so python tests/tools/dense_identity_timing.py [0|1]"""
import ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth
on = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 100_000_000
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
out = torch.empty((10_000_000 + 4096) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
res = {"dense_identity": on}
for keys, M in ((10_000_000, 1 << 24), (1_000_000, 1 << 21)):
    th = synth.zipf_thresholds(keys, 1.1)
    d_th = torch.from_numpy(th.view(np.int64)).cuda()
    synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr()); torch.cuda.synchronize()
    with nf.FlowTable(max_entries=M, profile=True) as tab:
        fn = nf._lib.lib.nfagg_debug_dense_identity
        fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
        assert fn(tab._h, on) == 0
        for rep in range(4):
            if rep == 1:
                tab.sync(); tab.reset_profile(); t0 = time.perf_counter()
            rc, c = tab.ingest_device(d.data_ptr(), n); assert rc == nf.OK and c == n
            flows = tab.evict_device(out.data_ptr(), 10_000_000 + 4096, nf.REASON_TIMEOUT)
        tab.sync(); dt = (time.perf_counter() - t0) / 3
        st = tab.stats()
        res["flows_%dm" % (keys // 1_000_000)] = {"ms_per_step": round(dt * 1e3, 3), "fold_call_ms": round(st.ingest_kernel_ms / st.ingest_launches, 4),
                                                 "evict_ms": round(st.evict_kernel_ms / st.evict_launches, 4), "flows": int(flows)}
        assert fn(tab._h, 0) == 0
print(json.dumps(res))
