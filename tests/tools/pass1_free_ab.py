#!/usr/bin/env python3
"""A/B: the 100 M-record fold call with pass 1 as shipped (ingest_variant 0) and without its barriers (17: published cache entries,
staging groups drained by their last writer) — fold call by HIP events, records compared with each other bit for bit (and with the
oracle by the test suite). usage: pass1_free_ab.py [--flows 1000000] [--hot 0] [--reps 5]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth
def arg(name, d): return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
n, keys, hot, reps = 100_000_000, arg("--flows", 1_000_000), arg("--hot", 0), arg("--reps", 5)
dedup = "--dedup" in sys.argv
variants = [int(x) for x in (sys.argv[sys.argv.index("--variants") + 1] if "--variants" in sys.argv else "0,17").split(",")]
th = synth.zipf_thresholds(keys, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
out = torch.empty((keys + 4096) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr(), hot_permille=hot, variant=2 if dedup else 0); torch.cuda.synchronize()
res, ev = {"flows": keys, "hot_permille": hot, "dedup": dedup}, {}
M = 1 << 21 if keys <= 1_000_000 else 1 << 24
for rnd in range(2):
    for v in variants:
        with nf.FlowTable(max_entries=M, ingest_variant=v, profile=True, mode=nf.MODE_KERNEL_DEDUP if dedup else nf.MODE_ACCOUNTER) as tab:
            for rep in range(reps + 1):
                if rep == 1:
                    tab.sync(); tab.reset_profile()
                rc, c = tab.ingest_device(d.data_ptr(), n); assert rc == nf.OK and c == n
                flows = tab.evict_device(out.data_ptr(), keys + 4096, nf.REASON_TIMEOUT)
            tab.sync()
            st = tab.stats()
            res.setdefault("variant_%d" % v, []).append({"fold_call_ms": round(st.ingest_kernel_ms / st.ingest_launches, 4), "bypassed_share": round(st.records_bypassed / (n * (reps + 1)), 4), "flows": int(flows)})
            if rnd == 0:
                ev[v] = nf.sort_by_key(out[: flows * 144].cpu().numpy().view(nf.FLOW_RECORD))
res["bit_identical"] = all(ev[v].tobytes() == ev[variants[0]].tobytes() for v in variants)
print(json.dumps(res))
