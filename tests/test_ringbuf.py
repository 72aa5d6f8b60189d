"""nfagg_ringbuf_drain (host-only, no GPU): bulk drain of a BPF ring buffer, against a literal
Python restatement of ringReader.readRecord (vendor/github.com/cilium/ebpf/ringbuf/ring.go:44-101)
called once per sample the way RingBufTracer does (pkg/flow/tracer_ringbuf.go:112-134)."""
import ctypes as C

import numpy as np
import pytest

BUSY, DISCARD, HDR = 0x80000000, 0x40000000, 8


class Ring:
    """A BPF ringbuf as user space sees it: data pages (power of two), producer and consumer positions."""

    def __init__(self, size, start_pos=0):
        self.size = size
        self.data = np.zeros(size, dtype=np.uint8)
        self.prod = np.array([start_pos], dtype=np.uint64)
        self.cons = np.array([start_pos], dtype=np.uint64)

    def push(self, payload: bytes, busy=False, discard=False):
        """What bpf_ringbuf_reserve + submit/discard leave in memory."""
        p = int(self.prod[0])
        hdr = len(payload) | (BUSY if busy else 0) | (DISCARD if discard else 0)
        blob = int(hdr).to_bytes(4, "little") + bytes(4) + payload + bytes(-len(payload) % 8)
        assert int(self.prod[0]) - int(self.cons[0]) + len(blob) <= self.size, "ring full"
        for k, b in enumerate(blob):
            self.data[(p + k) & (self.size - 1)] = b
        self.prod[0] = p + len(blob)
        return p

    def commit(self, pos):
        i = (pos + 3) & (self.size - 1)
        self.data[i] &= 0x7F


def ref_read_record(ring, cons):
    """ring.go:44-101 for one call. Returns (status, sample, new_cons): status in eor/busy/short/ok."""
    prod = int(ring.prod[0])
    mask = ring.size - 1
    while True:
        remaining = prod - cons
        if remaining == 0:
            return "eor", None, cons
        if remaining < HDR:
            return "short", None, cons
        start = cons & mask
        ln = int.from_bytes(bytes(ring.data[start:start + 4]), "little")
        if ln & BUSY:
            return "busy", None, cons
        c2 = cons + HDR
        data_len = ln & ~(BUSY | DISCARD)
        aligned = (data_len + 7) & ~7
        if prod - c2 < aligned:
            return "short", None, cons
        start = c2 & mask
        c2 += aligned
        if ln & DISCARD:
            cons = c2                       # atomic.StoreUintptr(rr.cons_pos, cons); continue
            continue
        sample = bytes(ring.data[(start + k) & mask] for k in range(data_len))
        return "ok", sample, c2


def ref_drain(ring, cap):
    """The tracer loop: read samples one by one; a sample that is not 144 bytes fails model.ReadFrom and is dropped."""
    cons = int(ring.cons[0])
    out, skipped = [], 0
    while len(out) < cap:
        st, sample, cons2 = ref_read_record(ring, cons)
        # discards consumed inside readRecord count as skipped: recount from positions
        if st != "ok":
            cons = cons2
            break
        cons = cons2
        if len(sample) == 144:
            out.append(sample)
        else:
            skipped += 1
    return out, cons, st if len(out) < cap else "full"


def drain(nf, ring, cap):
    L = nf._lib
    rb = L.RingBuf(ring.data.ctypes.data, ring.size - 1, ring.prod.ctypes.data, ring.cons.ctypes.data)
    dst = np.zeros(max(cap, 1) * 144, dtype=np.uint8)
    n, sk = C.c_size_t(0), C.c_size_t(0)
    errs = np.zeros(256, dtype=np.uint64)
    rc = L.lib.nfagg_ringbuf_drain(C.byref(rb), dst.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(sk), errs.ctypes.data_as(C.c_void_p))
    recs = [dst[k * 144:(k + 1) * 144].tobytes() for k in range(n.value)]
    return rc, recs, sk.value, errs


def test_drain_matches_per_sample_reader(nf, O):
    rng = np.random.default_rng(7)
    recs = O.gen_stream(400, seed=9, n_keys=50, variant=1)
    raw = [r.tobytes() for r in recs]
    # start near the end of the data area so that samples wrap around it
    ring = Ring(1 << 14, start_pos=(1 << 14) * 1000 - 152 * 2 - 40)
    busy_pos = None
    k = 0
    total_out = []
    for rnd in range(12):
        # fill
        while k < len(raw) and int(ring.prod[0]) - int(ring.cons[0]) + 200 <= ring.size:
            kind = rng.integers(0, 10)
            if kind == 0:
                ring.push(bytes(rng.integers(0, 256, int(rng.choice([8, 24, 100, 152])), dtype=np.uint8)))   # wrong-size sample
            elif kind == 1:
                ring.push(raw[k], discard=True); k += 1
            elif kind == 2 and busy_pos is None:
                busy_pos = ring.push(raw[k], busy=True); k += 1                                       # reserved, not yet committed
            else:
                ring.push(raw[k]); k += 1
        cap = int(rng.choice([1, 3, 17, 1000]))
        want, want_cons, why = ref_drain(ring, cap)
        rc, got, skipped, errs = drain(nf, ring, cap)
        assert rc == nf.OK
        assert got == want, (rnd, len(got), len(want))
        assert int(ring.cons[0]) == want_cons
        assert int(errs.sum()) == len(got) and all(errs[g[40 + 57]] > 0 for g in got)
        total_out += got
        if busy_pos is not None and why == "busy":
            ring.commit(busy_pos); busy_pos = None                                                    # the kernel commits it
    assert len(total_out) > 150


def test_drain_empty_full_and_truncated(nf, O):
    ring = Ring(1 << 12)
    rc, got, skipped, _ = drain(nf, ring, 10)
    assert (rc, got, skipped) == (nf.OK, [], 0)
    rec = O.gen_stream(1, seed=1, n_keys=1)[0].tobytes()
    for _ in range(5):
        ring.push(rec)
    rc, got, _, _ = drain(nf, ring, 2)                       # destination full: the rest stays in the ring
    assert rc == nf.OK and len(got) == 2 and int(ring.prod[0]) - int(ring.cons[0]) == 3 * 152
    # producer position inside a sample: io.ErrUnexpectedEOF in the reference, EINVAL here, nothing consumed past it
    ring.prod[0] -= 100
    before = int(ring.cons[0])
    rc, got, _, _ = drain(nf, ring, 10)
    assert rc == nf._lib.EINVAL and len(got) == 2 and int(ring.cons[0]) == before + 2 * 152


@pytest.mark.parametrize("pool", [None, 3, 16])
def test_bulk_drain_runs_of_plain_samples_split_over_threads(nf, O, pool):
    """Runs of >= 32768 plain 144-byte samples take the multi-part path of nfagg_ringbuf_drain (each part verifies the headers of its
    range; the run ends at the first header that is not "144 bytes, committed"): same result as the per-sample reader, with a
    discarded sample, a wrong-length sample and a busy sample placed inside / between the runs, and the ring wrapping. pool: the
    process's copy workers (csrc/nfagg_hostpool.h) as they are (none before the first nfagg_create: the caller copies alone), or
    re-shaped to 3 / 16 unbound workers (nfagg_host_threads)."""
    if pool is not None:
        assert nf.host_threads(pool, -1) == pool
        info = nf.host_info()
        assert info["workers"] == pool and 1 <= info["parts"] <= pool + 1 and not info["bound"]
    rng = np.random.default_rng(5)
    n = 150_000
    recs = O.gen_stream(n, seed=21, n_keys=5000, variant=1)
    raw = recs.view(np.uint8).reshape(n, 144)
    size = 1 << 25
    ring = Ring(size, start_pos=size * 3 - 152 * 40_000)              # the first run wraps around the data area

    def push_many(lo, hi):                                            # vectorised bpf_ringbuf_reserve + submit
        blob = np.zeros((hi - lo, 152), dtype=np.uint8)
        blob[:, 0] = 144
        blob[:, 8:] = raw[lo:hi]
        flat = blob.reshape(-1)
        p = int(ring.prod[0])
        pos = p & (size - 1)
        first = min(len(flat), size - pos)
        ring.data[pos:pos + first] = flat[:first]
        ring.data[:len(flat) - first] = flat[first:]
        ring.prod[0] = p + len(flat)

    push_many(0, 70_000)
    ring.push(raw[70_000].tobytes(), discard=True)                    # ends the first run
    push_many(70_001, 110_000)
    ring.push(bytes(24))                                              # wrong length
    push_many(110_000, 149_000)
    busy = ring.push(raw[149_000].tobytes(), busy=True)               # reserved, not committed: the drain stops here
    push_many(149_001, 150_000)
    shadow = Ring(size); shadow.data, shadow.prod, shadow.cons = ring.data, ring.prod.copy(), ring.cons.copy()
    want, want_cons, why = ref_drain(shadow, n)
    rc, got, skipped, errs = drain(nf, ring, n)
    assert rc == nf.OK and why == "busy" and len(want) == 148_999
    assert int(ring.cons[0]) == want_cons and skipped == 2
    assert got == want
    assert int(errs.sum()) == len(got)
    ring.commit(busy)
    shadow.cons = ring.cons.copy()
    want2, want_cons2, _ = ref_drain(shadow, 500)                     # a small destination: stops when it is full
    rc, got2, _, _ = drain(nf, ring, 500)
    assert rc == nf.OK and got2 == want2 and len(got2) == 500 and int(ring.cons[0]) == want_cons2
