"""Ring buffer -> pinned staging buffer -> device fold -> eviction, as ONE path:
nfagg_ringbuf_drain writes the committed samples of a BPF ring STRAIGHT into the buffer nfagg_staging_acquire lent out,
nfagg_staging_commit sends it up and folds it, nfagg_evict delivers the flows. Replaces, batched, the per-sample loop of
RingBufTracer.listenAndForwardRingBuffer (pkg/flow/tracer_ringbuf.go:112-134: ringbuf read -> model.ReadFrom -> channel send)
in front of Accounter.Account (pkg/flow/account.go:58-100).

Expected result: the oracle Accounter fed with what a literal per-sample restatement of ringReader.readRecord
(vendor/github.com/cilium/ebpf/ringbuf/ring.go:44-101; tests/test_ringbuf.py) hands over, sample by sample."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_records_equal
from test_ringbuf import Ring, ref_drain

pytestmark = pytest.mark.gpu


def _drain_into(nf, ring, buf, cap):
    """nfagg_ringbuf_drain with dst = the staging buffer itself (no intermediate copy)."""
    L = nf._lib
    rb = L.RingBuf(ring.data.ctypes.data, ring.size - 1, ring.prod.ctypes.data, ring.cons.ctypes.data)
    n, sk = C.c_size_t(0), C.c_size_t(0)
    rc = L.lib.nfagg_ringbuf_drain(C.byref(rb), buf.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(sk), None)
    return rc, n.value, sk.value


def _producer(rng, ring, raw, k, state):
    """The kernel side: reserve + submit / discard samples until the ring is nearly full. Leaves at most one sample BUSY."""
    while k < len(raw) and int(ring.prod[0]) - int(ring.cons[0]) + 200 <= ring.size:
        kind = rng.integers(0, 12)
        if kind == 0:
            ring.push(bytes(rng.integers(0, 256, int(rng.choice([8, 24, 100, 152])), dtype=np.uint8)))     # wrong length: ReadFrom fails
        elif kind == 1:
            ring.push(raw[k], discard=True); k += 1
        elif kind == 2 and state["busy"] is None:
            state["busy"] = ring.push(raw[k], busy=True); k += 1                                      # reserved, not yet committed
        else:
            ring.push(raw[k]); k += 1
    return k


@pytest.mark.parametrize("max_entries,staging", [(1 << 16, 4096), (1 << 16, 37), (300, 512)])
def test_ring_drained_into_staging_then_folded_and_evicted(nf, O, max_entries, staging):
    rng = np.random.default_rng(11)
    recs = O.gen_stream(6000, seed=13, n_keys=900, thresholds=O.zipf_thresholds(900, 1.1), variant=1)
    raw = [r.tobytes() for r in recs]
    ring = Ring(1 << 15, start_pos=(1 << 15) * 77 - 152 * 3 - 24)         # the first samples wrap around the data area
    state = {"busy": None}
    want_stream = []                                                      # what the per-sample reader hands to the Accounter
    got_evictions = []
    with nf.FlowTable(max_entries=max_entries, staging_records=staging) as tab:
        k, rounds = 0, 0
        while k < len(raw) or int(ring.prod[0]) != int(ring.cons[0]):
            k = _producer(rng, ring, raw, k, state)
            if rounds % 3 == 2 and state["busy"] is not None:
                ring.commit(state["busy"]); state["busy"] = None          # the kernel commits the reserved sample
            rounds += 1
            while True:
                buf = tab.staging_acquire()
                assert len(buf) == staging
                # the literal reader on a COPY of the positions: what the reference would have delivered from this ring state
                shadow = Ring(ring.size); shadow.data, shadow.prod, shadow.cons = ring.data, ring.prod.copy(), ring.cons.copy()
                want, want_cons, why = ref_drain(shadow, staging)
                rc, n, skipped = _drain_into(nf, ring, buf, staging)
                assert rc == nf.OK and n == len(want) and int(ring.cons[0]) == want_cons
                want_stream += want
                off = 0
                rc, c = tab.staging_commit(n)
                off += c
                rest = np.array(buf[off:n], copy=True) if rc == nf.FULL else None   # the staging buffer goes back to the ring at commit
                while rc == nf.FULL:                                       # account.go:85-94: evict, resubmit the rest
                    got_evictions.append(nf.sort_by_key(tab.evict(nf.REASON_FULL)))
                    rc, c = tab.ingest(rest)
                    rest = rest[c:]
                if n < staging:
                    break                                                 # ring empty, or the next sample is still busy
            if k >= len(raw) and state["busy"] is not None:
                ring.commit(state["busy"]); state["busy"] = None
        got_evictions.append(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)))
    assert len(want_stream) > 4000
    stream = np.frombuffer(b"".join(want_stream), dtype=O.FLOW_RECORD)
    want_ev = O.run_accounter(stream, max_entries)
    assert len(want_ev) == len(got_evictions)
    if max_entries == 300:
        assert len(want_ev) > 3
    for e, (g, (_, w)) in enumerate(zip(got_evictions, want_ev)):
        assert_records_equal(g, w, "eviction %d" % e)


def test_truncated_ring_tail_stops_the_drain_but_not_the_path(nf, O):
    """Producer position inside a sample (io.ErrUnexpectedEOF in the reference, ring.go:74-80): NFAGG_EINVAL, what was drained
    before it is committed and folded, nothing is consumed past it."""
    recs = O.gen_stream(50, seed=3, n_keys=20, variant=1)
    ring = Ring(1 << 13)
    for r in recs:
        ring.push(r.tobytes())
    ring.prod[0] -= 100                                                   # the last sample is cut
    with nf.FlowTable(max_entries=1 << 12, staging_records=256) as tab:
        buf = tab.staging_acquire()
        rc, n, _ = _drain_into(nf, ring, buf, len(buf))
        assert rc == nf._lib.EINVAL and n == 49 and int(ring.cons[0]) == 49 * 152
        assert tab.staging_commit(n) == (nf.OK, 49)
        got = nf.sort_by_key(tab.evict(nf.REASON_CLOSING))
    assert_records_equal(got, O.run_accounter(recs[:49], 1 << 12)[0][1])


def test_an_explicitly_shaped_host_pool_survives_nfagg_create(nf):
    """nfagg_host_threads(2, -1) is how an agent keeps the library's copy workers off its own cores (include/nfagg.h); the implicit
    default of the first nfagg_create — 16 workers bound to the GPU's NUMA node — must not undo it (round-5 advisor finding).
    Run in a process of its own: the pool is per process and other tests have created handles already."""
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import netobserv_ebpf_agent_amd as nf\n"
            "assert nf.host_threads(2, -1) == 2\n"
            "with nf.FlowTable(max_entries=1024) as tab:\n"
            "    info = nf.host_info()\n"
            "assert info['workers'] == 2 and not info['bound'] and info['numa_node'] == -1, info\n"
            "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-500:] + p.stderr[-2000:]
