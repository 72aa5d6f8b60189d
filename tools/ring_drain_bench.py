#!/usr/bin/env python3
"""Host-only: nfagg_ringbuf_drain over a 64 MiB ring full of committed 144-byte flow samples (the bulk path: runs of plain samples
copied by several threads). NFAGG_LIB selects the build. Usage: python tools/ring_drain_bench.py"""
import sys, time, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import netobserv_ebpf_agent_amd as nf
L = nf._lib
size = 1 << 26
n = size // 152 - 8
data = np.zeros(size, dtype=np.uint8)
v = data[: n * 152].reshape(n, 152)
v[:, 0] = 144
v[:, 8:] = np.arange(144, dtype=np.uint8)
prod = np.array([n * 152], dtype=np.uint64); cons = np.array([0], dtype=np.uint64)
pinned = len(sys.argv) > 1 and sys.argv[1] == "pinned"
if pinned:
    keep = nf.PinnedRecords(n)                       # page-locked, like the staging buffer the drain writes into
    out = keep.records.view(np.uint8).reshape(-1)
else:
    out = np.empty(n * 144, dtype=np.uint8)
out[:] = 0
rb = L.RingBuf(data.ctypes.data, size - 1, prod.ctypes.data, cons.ctypes.data)
node = nf.device_numa_node(0)
for threads, bind in ((0, node), (4, node), (8, node), (16, node), (32, node), (16, -1)):
    nf.host_threads(threads, bind)                   # what the first nfagg_create does: (0, the GPU's NUMA node)
    info = nf.host_info()
    rates = []
    for rep in range(9):
        cons[0] = 0
        nn, sk = C.c_size_t(0), C.c_size_t(0)
        t = time.perf_counter()
        rc = L.lib.nfagg_ringbuf_drain(C.byref(rb), out.ctypes.data_as(C.c_void_p), n, C.byref(nn), C.byref(sk), None)
        dt = time.perf_counter() - t
        assert rc == 0 and nn.value == n
        rates.append(round(n / dt / 1e6))
    rates.sort()
    print("drain into %s memory, pool %s: min/median/max %d / %d / %d M records/s (median %.1f GB/s); all passes: %s"
          % ("page-locked" if pinned else "pageable", info, rates[0], rates[len(rates) // 2], rates[-1], rates[len(rates) // 2] * 144 / 1e3, rates))
assert (out.reshape(n, 144) == np.arange(144, dtype=np.uint8)).all()
