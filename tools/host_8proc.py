#!/usr/bin/env python3
"""The HOST side of an 8-GPU node, rehearsed on whatever box this runs on (round-5 review, item 5b): P processes (one per GPU of the
node it stands for), each draining a synthetic BPF-style ring (64 MiB of committed 144-byte samples) through its OWN pool of copy
workers (csrc/nfagg_hostpool.h) into its own page-locked staging buffer — nfagg_ringbuf_drain, the product's producer path
(pkg/flow/tracer_ringbuf.go:112-134 replaced) — all at once, for a few seconds. No device work: the question is whether P x (ring
reads + staging writes) fit the box's memory system and whether the pools' NUMA bindings collide, which the one GPU of the box cannot
answer for the link anyway (eight links need eight GPUs).

  process r binds its pool to NUMA node r % (nodes of the box) — on a real node: the node its GPU hangs off (nfagg_device_numa_node)
  --unbound: the pools are left to the scheduler (what round 4 had)

Prints per-process rates (min / median / max over the passes), their spread, the aggregate, and the same for ONE process alone.
usage: host_8proc.py [--procs 8] [--seconds 3] [--unbound] [--workers 0]"""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def numa_nodes():
    try:
        return sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
    except OSError:
        return [0]


def worker(r, procs, node, workers, seconds, barrier, q):
    import numpy as np
    import netobserv_ebpf_agent_amd as nf
    L = nf._lib
    size = 1 << 26
    n = size // 152 - 8
    data = np.zeros(size, dtype=np.uint8)
    v = data[: n * 152].reshape(n, 152)
    v[:, 0] = 144
    v[:, 8:] = np.arange(144, dtype=np.uint8)
    prod = np.array([n * 152], dtype=np.uint64)
    cons = np.array([0], dtype=np.uint64)
    keep = nf.PinnedRecords(n)
    out = keep.records.view(np.uint8).reshape(-1)
    out[:] = 0
    rb = L.RingBuf(data.ctypes.data, size - 1, prod.ctypes.data, cons.ctypes.data)
    nf.host_threads(workers, node)
    info = nf.host_info()
    nn, sk = C.c_size_t(0), C.c_size_t(0)
    for _ in range(2):                                                 # warm: pages, workers
        cons[0] = 0
        L.lib.nfagg_ringbuf_drain(C.byref(rb), out.ctypes.data_as(C.c_void_p), n, C.byref(nn), C.byref(sk), None)
    barrier.wait()
    rates, t_end = [], time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        cons[0] = 0
        t = time.perf_counter()
        rc = L.lib.nfagg_ringbuf_drain(C.byref(rb), out.ctypes.data_as(C.c_void_p), n, C.byref(nn), C.byref(sk), None)
        dt = time.perf_counter() - t
        assert rc == 0 and nn.value == n
        rates.append(n / dt / 1e6)
    barrier.wait()
    assert (out.reshape(n, 144) == np.arange(144, dtype=np.uint8)).all()
    rates.sort()
    q.put((r, node, str(info), len(rates), rates[0], rates[len(rates) // 2], rates[-1], sum(rates) / len(rates)))
    keep.close()


def run(procs, seconds, unbound, workers):
    nodes = numa_nodes()
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(procs), ctx.Queue()
    ps = []
    for r in range(procs):
        node = -1 if unbound else nodes[r % len(nodes)]
        p = ctx.Process(target=worker, args=(r, procs, node, workers, seconds, barrier, q))
        p.start()
        ps.append(p)
    rows = sorted(q.get(timeout=seconds + 300) for _ in range(procs))
    for p in ps:
        p.join()
    return rows


if __name__ == "__main__":
    procs, seconds, workers = arg("--procs", 8), arg("--seconds", 3.0), arg("--workers", 0)
    unbound = "--unbound" in sys.argv
    print("host: %d CPUs, NUMA nodes %s; %s pools, %s workers per pool" % (os.cpu_count(), numa_nodes(), "unbound" if unbound else "node-bound",
                                                                          workers or "calibrated"))
    for label, p in (("ONE process alone", 1), ("%d processes at once" % procs, procs)):
        rows = run(p, seconds, unbound, workers)
        print("%s — ring -> page-locked staging buffer (nfagg_ringbuf_drain), %.0f s each:" % (label, seconds))
        for r, node, info, passes, lo, med, hi, mean in rows:
            print("  process %d (node %2d, pool %s): %3d passes, min / median / max %4.0f / %4.0f / %4.0f M records/s (median %.1f GB/s written + as much read)"
                  % (r, node, info, passes, lo, med, hi, med * 144 / 1e3))
        agg = sum(x[7] for x in rows)
        meds = [x[5] for x in rows]
        print("  aggregate (sum of the processes' mean rates): %.0f M records/s = %.1f GB/s into pinned memory; slowest / fastest process median %.0f / %.0f "
              "(the link of one MI355X takes ~396 M records/s = 57 GB/s: %d links want %.0f GB/s)"
              % (agg, agg * 144 / 1e3, min(meds), max(meds), p, 57.0 * p))
