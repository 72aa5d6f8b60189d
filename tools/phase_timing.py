#!/usr/bin/env python3
"""Diagnostics: per-phase wave-cycle breakdown of the cached ingest kernel (ingest_variant 6)
on the bench stream. Run on the GPU box: python tools/phase_timing.py [flows] [records]."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NFAGG_LIB", os.path.join(ROOT, "netobserv-ebpf-agent_amd", "lib", "libnfagg_diag.so"))   # phase-timing builds
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth

flows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 6
th = synth.zipf_thresholds(flows, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=flows, d_thresholds=d_th.data_ptr())
torch.cuda.synchronize()
out = torch.empty(flows * 144 + 16, dtype=torch.uint8, device="cuda")
tab = nf.FlowTable(max_entries=int(os.environ.get("MAX_ENTRIES", 1 << 21)), profile=True, ingest_variant=variant)
for it in range(3):
    tab.ingest_device(d.data_ptr(), n)
    tab.evict_device(out.data_ptr(), flows)
st = tab.stats()
ph = (C.c_uint64 * 8)()
nf._lib.lib.nfagg_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
nf._lib.lib.nfagg_debug_phase_cycles(tab._h, ph)
names = ["load+hash", "A", "barrier1", "B", "barrier2", "C", "flush"]
if variant == 9:     # pass 2 of the two-pass fold
    names = ["gather+hash", "claim", "barrier", "fold", "flushes", "retry/direct", "sort+init+tail"]
print("direct merges:", st.records_direct if hasattr(st, "records_direct") else "n/a")
tot = sum(ph[:7]) or 1
print(f"flows={flows} n={n} kernel_ms={st.ingest_kernel_ms / st.ingest_launches:.3f} bypass={st.records_bypassed / (3 * n):.3f}")
for k, nm in enumerate(names):
    print(f"  {nm:10s} {ph[k] / tot * 100:5.1f} %   {ph[k] / 3 / 4096 / 1e3:9.1f} kcycles/wave/launch")
tab.close()
