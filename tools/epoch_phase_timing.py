"""Where does an epoch of the persistent epoch kernel (csrc/nfagg_epochs.hip) spend its time? Needs lib/libnfagg_diag.so
(NFAGG_LIB): nfagg_debug_epoch_phases returns 100 MHz ticks of lane 0 per phase, accumulated over the launches."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NFAGG_LIB", os.path.join(ROOT, "netobserv-ebpf-agent_amd", "lib", "libnfagg_diag.so"))
import numpy as np
import torch
import netobserv_ebpf_agent_amd as nf
from netobserv_ebpf_agent_amd import synth, _lib as L

n, keys = 4_000_000, 1_000_000
th = synth.zipf_thresholds(keys, 1.1)
d_th = torch.from_numpy(th.view(np.int64)).cuda()
d = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
synth.stream_device(d.data_ptr(), n, seed=2, n_keys=keys, d_thresholds=d_th.data_ptr())
out = torch.empty((n + 70000) * 144, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
fn = L.lib.nfagg_debug_epoch_phases
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]
names = ["P1 claim", "sync1", "P2 flags", "sync2", "P3 split(+sync)", "P4 fold", "sync4+P5 evict", "sync5"]
for me, variant in ((5000, 0), (5000, 30), (500, 0), (500, 30), (30000, 0), (30000, 30)):
    with nf.FlowTable(max_entries=me, ingest_variant=variant) as tab:
        for rep in range(2):
            t0 = time.perf_counter()
            rc, c, ends = tab.account_device(d.data_ptr(), n, out.data_ptr(), n + 70000, 65000)
            dt = time.perf_counter() - t0
            tab.evict(nf.REASON_CLOSING, cap=me + 10)
            ph = (C.c_uint64 * 8)()
            fn(tab._h, ph)
        ep = max(len(ends), 1)
        print("max_entries %d, %s: %d epochs, %.1f us per epoch, %.1f M records/s" % (me, "persistent kernel" if variant == 30 else "kernel chain", len(ends), dt / ep * 1e6, n / dt / 1e6))
        if variant == 30:
            print("   " + ", ".join("%s %.1f" % (nm, ph[k] / 100.0 / ep) for k, nm in enumerate(names)) + "  (us per epoch)")
