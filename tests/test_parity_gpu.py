"""Parity tests proper: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs — bit-exact on every byte of every evicted
flow_record_t. Mirrors the reference's own tests where they exist
(pkg/flow/account_test.go) and widens to the cases SURVEY.md §8(c) lists as
unpinned (order-dependent fields, wrap-around, evict-on-full inside a batch)."""
import queue
import threading
import time

import numpy as np
import pytest

from conftest import assert_records_equal
from test_oracle_kat import K1, K2, K3, mk

pytestmark = pytest.mark.gpu

PN = dict(start="start_mono_time_ts", end="end_mono_time_ts")


def drive_product(tab, records, batch):
    """Feed `records` in arrival order in batches; evict on NFAGG_FULL exactly as
    account.go:85-94; final eviction as on close (:73-80)."""
    nf = __import__("netobserv_ebpf_agent_amd")
    out, off, n = [], 0, len(records)
    while off < n:
        hi = min(n, off + batch)
        while off < hi:
            rc, c = tab.ingest(records[off:hi])
            off += c
            if rc == nf.FULL:
                out.append(("full", nf.sort_by_key(tab.evict(nf.REASON_FULL))))
    out.append(("closing", nf.sort_by_key(tab.evict(nf.REASON_CLOSING))))
    return out


def check_against_oracle(nf, O, records, max_entries, batch, **table_kw):
    want = O.run_accounter(records, max_entries)
    with nf.FlowTable(max_entries=max_entries, **table_kw) as tab:
        got = drive_product(tab, records.view(nf.FLOW_RECORD), batch)
    assert [r for r, _ in got] == [r for r, _ in want], "eviction sequence differs"
    for k, ((_, g), (_, w)) in enumerate(zip(got, want)):
        assert_records_equal(g, w, f"eviction #{k}")
    return want


# ---------------------------------------------------------------- reference tests through the C ABI
def test_evict_max_entries(nf):
    """pkg/flow/account_test.go:47-128 TestEvict_MaxEntries, via the host mirror of Accounter.Account."""
    now = 1661272402 * 10**9
    acc = nf.NewAccounter(2, 3600.0, lambda: now, lambda: 1000, nf.NoOp())
    inputs, evictor = queue.Queue(), queue.Queue()
    th = threading.Thread(target=acc.Account, args=(inputs, evictor), daemon=True)
    th.start()
    assert evictor.empty()
    R = nf.FLOW_RECORD
    inputs.put(mk(R, K1, PN, bytes=123, packets=1, start=123, end=123, flags=1))
    inputs.put(mk(R, K2, PN, bytes=456, packets=1, start=456, end=456, flags=1))
    inputs.put(mk(R, K1, PN, bytes=321, packets=1, start=789, end=789, flags=1))
    time.sleep(0.5)
    assert evictor.empty()                                   # nothing evicted until the maximum is surpassed
    inputs.put(mk(R, K3, PN, bytes=111, packets=1, start=888, end=888, flags=1))
    r = evictor.get(timeout=10)
    assert len(r) == 2
    by_port = {int(x.ID["src_port"]): x for x in r}
    a, b = by_port[333], by_port[12]
    m = a.Metrics
    assert (m["bytes"], m["packets"], m["start_mono_time_ts"], m["end_mono_time_ts"], m["flags"]) == (444, 2, 123, 789, 1)
    assert a.TimeFlowStart == now - (1000 - 123) and a.TimeFlowEnd == now - (1000 - 789)
    assert [(i.Interface, i.Direction) for i in a.Interfaces] == [("[namer unset] 0", 0)]
    m = b.Metrics
    assert (m["bytes"], m["packets"], m["start_mono_time_ts"], m["end_mono_time_ts"], m["flags"]) == (456, 1, 456, 456, 1)
    assert b.TimeFlowStart == now - (1000 - 456) and b.TimeFlowEnd == now - (1000 - 456)
    time.sleep(0.2)
    assert evictor.empty()
    assert acc.metrics.evictions_total == {("accounter", "full"): 1}
    assert acc.metrics.evicted_flows_total == {("accounter", "full"): 2}
    inputs.put(nf.CLOSE)
    r = evictor.get(timeout=10)                              # closing flushes k3 (account.go:73-80)
    assert len(r) == 1 and int(r[0].ID["dst_port"]) == 443 and int(r[0].Metrics["bytes"]) == 111
    th.join(timeout=10)
    acc.close()


def test_evict_period(nf):
    """pkg/flow/account_test.go:130-217 TestEvict_Period (timeouts scaled up: a GPU box pages the HIP module in)."""
    now = 1661272402 * 10**9
    acc = nf.NewAccounter(200, 0.4, lambda: now, lambda: 1000, nf.NoOp())
    len(acc.table)                                           # warm the device before the clock matters
    inputs, evictor = queue.Queue(), queue.Queue()
    th = threading.Thread(target=acc.Account, args=(inputs, evictor), daemon=True)
    th.start()
    R = nf.FLOW_RECORD
    for t in (123, 456, 789):
        inputs.put(mk(R, K1, PN, bytes=10, packets=1, start=t, end=t, flags=1))
    time.sleep(0.8)                                          # forcing at least one eviction here
    for t in (1123, 1456):
        inputs.put(mk(R, K1, PN, bytes=10, packets=1, start=t, end=t, flags=1))
    r = evictor.get(timeout=10)
    assert len(r) == 1
    m = r[0].Metrics
    assert (m["bytes"], m["packets"], m["start_mono_time_ts"], m["end_mono_time_ts"], m["flags"]) == (30, 3, 123, 789, 1)
    assert r[0].TimeFlowStart == now - 1000 + 123 and r[0].TimeFlowEnd == now - 1000 + 789
    r = evictor.get(timeout=10)
    assert len(r) == 1
    m = r[0].Metrics
    assert (m["bytes"], m["packets"], m["start_mono_time_ts"], m["end_mono_time_ts"], m["flags"]) == (20, 2, 1123, 1456, 1)
    time.sleep(0.9)
    assert evictor.empty()                                   # no more flows are evicted (empty table: no eviction, :64-66)
    inputs.put(nf.CLOSE)
    assert evictor.get(timeout=10) == []                     # closing always sends, even an empty batch
    th.join(timeout=10)
    acc.close()


def test_batching_policy_keeps_tick_and_closing_semantics(nf):
    """The shim gathers records until BATCH_RECORDS wait or the oldest has waited BATCH_TIMEOUT (accounter.py; INTEGRATION.md §3:
    the reference's channel delivers ONE record per operation, pkg/agent/agent.go:408, and a call costs ~80 us whatever it
    holds). What must survive (account.go:61-96): records received before a tick are in the eviction the tick makes; closing
    accounts what was received, then evicts; an eviction on full happens at the record that finds the map full, whatever the
    batch boundaries; and the calls are few."""
    now = 1661272402 * 10**9
    R = nf.FLOW_RECORD
    # (1) a long batch timeout: the records wait in the shim; the TICK flushes them first, and its eviction holds them
    acc = nf.NewAccounter(100, 0.5, lambda: now, lambda: 1000, nf.NoOp(), batch_records=1000, batch_timeout=30.0)
    len(acc.table)
    inputs, evictor = queue.Queue(), queue.Queue()
    th = threading.Thread(target=acc.Account, args=(inputs, evictor), daemon=True)
    th.start()
    for t in (123, 456, 789):
        inputs.put(mk(R, K1, PN, bytes=10, packets=1, start=t, end=t, flags=1))
    r = evictor.get(timeout=10)                               # the timeout eviction, although no batch was ever "ready"
    assert len(r) == 1 and (int(r[0].Metrics["bytes"]), int(r[0].Metrics["packets"])) == (30, 3)
    assert acc.calls == 1                                     # three channel operations, one call
    # (2) closing: what was received is accounted, then evicted
    inputs.put(mk(R, K2, PN, bytes=7, packets=1, start=5, end=5, flags=1))
    inputs.put(mk(R, K2, PN, bytes=8, packets=1, start=6, end=6, flags=1))
    inputs.put(nf.CLOSE)
    r = evictor.get(timeout=10)
    assert len(r) == 1 and (int(r[0].Metrics["bytes"]), int(r[0].Metrics["packets"])) == (15, 2)
    th.join(timeout=10)
    assert acc.calls == 2
    acc.close()
    # (3) a batch of a given size: flushed when it is full, not before the timeout; the eviction on full falls where the
    # reference's does (TestEvict_MaxEntries' records, batch of 4: the fourth record finds the map of 2 full)
    acc = nf.NewAccounter(2, 3600.0, lambda: now, lambda: 1000, nf.NoOp(), batch_records=4, batch_timeout=30.0)
    inputs, evictor = queue.Queue(), queue.Queue()
    th = threading.Thread(target=acc.Account, args=(inputs, evictor), daemon=True)
    th.start()
    inputs.put(mk(R, K1, PN, bytes=123, packets=1, start=123, end=123, flags=1))
    inputs.put(mk(R, K2, PN, bytes=456, packets=1, start=456, end=456, flags=1))
    inputs.put(mk(R, K1, PN, bytes=321, packets=1, start=789, end=789, flags=1))
    time.sleep(0.3)
    assert evictor.empty() and acc.calls == 0
    inputs.put(mk(R, K3, PN, bytes=111, packets=1, start=888, end=888, flags=1))
    r = evictor.get(timeout=10)
    assert sorted(int(x.Metrics["bytes"]) for x in r) == [444, 456]
    assert acc.calls == 1 and acc.metrics.evictions_total == {("accounter", "full"): 1}
    inputs.put(nf.CLOSE)
    r = evictor.get(timeout=10)
    assert len(r) == 1 and int(r[0].Metrics["bytes"]) == 111
    th.join(timeout=10)
    acc.close()
    # (4) the default policy under a stream of single records: ~1 ms batches, far fewer calls than records, nothing lost
    acc = nf.NewAccounter(5000, 3600.0, lambda: now, lambda: 1000, nf.NoOp())
    len(acc.table)
    inputs, evictor = queue.Queue(), queue.Queue()
    th = threading.Thread(target=acc.Account, args=(inputs, evictor), daemon=True)
    th.start()
    n = 20_000
    for k in range(n):
        inputs.put(mk(R, K1 if k % 2 else K2, PN, bytes=1, packets=1, start=1 + k, end=1 + k, flags=1))
    inputs.put(nf.CLOSE)
    r = evictor.get(timeout=30)
    assert len(r) == 2 and sum(int(x.Metrics["packets"]) for x in r) == n
    th.join(timeout=10)
    assert acc.calls < n // 10, acc.calls
    acc.close()


def test_timeout_eviction_fires_under_sustained_load(nf):
    """account.go:61-71: the ticker arm of the select is served although `in` never runs dry (the host mirror
    used to starve it). Records keep arriving for ~5 evict periods; 'timeout' evictions must happen meanwhile."""
    acc = nf.NewAccounter(1000, 0.25, lambda: 0, lambda: 1000, nf.NoOp())
    len(acc.table)
    inputs, evictor = queue.Queue(), queue.Queue()
    th = threading.Thread(target=acc.Account, args=(inputs, evictor), daemon=True)
    th.start()
    R = nf.FLOW_RECORD
    t_end = time.monotonic() + 1.3
    fed = 0
    while time.monotonic() < t_end:
        while inputs.qsize() < 50:                           # the queue is never empty
            inputs.put(mk(R, K1, PN, bytes=1, packets=1, start=1 + fed, end=1 + fed, flags=1))
            fed += 1
        time.sleep(0.001)
    during = acc.metrics.evictions_total.get(("accounter", "timeout"), 0)
    inputs.put(nf.CLOSE)
    th.join(timeout=20)
    assert during >= 3, f"{during} timeout evictions in 5 periods of sustained load"
    total = 0
    while not evictor.empty():
        total += sum(int(r.Metrics["packets"]) for r in evictor.get())
    assert total == fed                                      # nothing lost across the evictions
    acc.close()


# ---------------------------------------------------------------- seeded streams vs the oracle
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("ingest_variant", [0, 1, 3, 4, 5, 7, 10, 11, 17])
@pytest.mark.parametrize("batch", [1 << 30, 1000, 257])
def test_stream_parity(nf, O, variant, ingest_variant, batch):
    th = O.zipf_thresholds(3000, 1.1)
    recs = O.gen_stream(40000, seed=100 + variant, n_keys=3000, thresholds=th, variant=variant)
    want = check_against_oracle(nf, O, recs, 1 << 16, batch, ingest_variant=ingest_variant)
    assert len(want) == 1 and len(want[0][1]) > 1500


def test_single_record_batches(nf, O):
    recs = O.gen_stream(600, seed=5, n_keys=40, variant=1)
    check_against_oracle(nf, O, recs, 1000, 1)


def test_config1_10k_records_1k_keys(nf, O):
    """BASELINE configs[0] on the GPU path (the CPU-only leg is tests/test_host_logic.py)."""
    recs = O.gen_stream(10000, seed=1, n_keys=1000)
    want = check_against_oracle(nf, O, recs, 5000, 1 << 30)
    assert len(want[0][1]) == len(np.unique(recs["id"]["src_port"]))


@pytest.mark.parametrize("ingest_variant", [0, 1, 3, 4, 5, 7, 10, 11, 17])
def test_hot_key_stream(nf, O, ingest_variant):
    """BASELINE configs[4] shape: 90 % of the records are one flow (LDS / atomic contention)."""
    th = O.zipf_thresholds(5000, 1.1)
    recs = O.gen_stream(60000, seed=5, n_keys=5000, thresholds=th, hot_permille=900, variant=1)
    check_against_oracle(nf, O, recs, 1 << 16, 1 << 30, ingest_variant=ingest_variant)


@pytest.mark.parametrize("ingest_variant", [0, 1, 7, 10, 17])
def test_wraparound_and_zero_fields(nf, O, ingest_variant):
    """u64 bytes / u32 packets wrap (flow_content.go:42-43) and all-zero optional fields."""
    recs = O.gen_stream(3000, seed=9, n_keys=3, variant=1)   # ~1000 records per key, wrap values injected
    check_against_oracle(nf, O, recs, 100, 1 << 30, ingest_variant=ingest_variant)
    z = np.zeros(10, dtype=O.FLOW_RECORD)                     # ten all-zero records: one flow, everything zero
    check_against_oracle(nf, O, z, 100, 1 << 30, ingest_variant=ingest_variant)


@pytest.mark.parametrize("ingest_variant", [0, 1, 7, 10, 17])
@pytest.mark.parametrize("max_entries,batch", [(2, 1 << 30), (100, 1 << 30), (100, 333), (1, 50), (999, 4096), (3000, 1 << 30)])
def test_evict_on_full_inside_batches(nf, O, max_entries, batch, ingest_variant):
    """account.go:85-94: the arrival of the (maxEntries+1)-th distinct key flushes what was
    accumulated so far — the split point must be found inside a parallel batch."""
    th = O.zipf_thresholds(1000, 1.1)
    recs = O.gen_stream(20000, seed=77, n_keys=1000, thresholds=th, variant=1)
    want = check_against_oracle(nf, O, recs, max_entries, batch, ingest_variant=ingest_variant)
    if max_entries < 500:
        assert sum(1 for r, _ in want if r == "full") >= 2
        assert all(len(b) == max_entries for r, b in want if r == "full")


def test_empty_and_ragged_inputs(nf, O):
    with nf.FlowTable(max_entries=16) as tab:
        assert tab.ingest(np.zeros(0, dtype=nf.FLOW_RECORD)) == (nf.OK, 0)
        assert len(tab) == 0
        assert len(tab.evict(nf.REASON_TIMEOUT)) == 0
        s = tab.stats()
        assert list(s.evictions) == [0, 0, 0]                 # an empty table does not count a timeout eviction (:64-66)
        assert len(tab.evict(nf.REASON_CLOSING)) == 0
        assert list(tab.stats().evictions) == [0, 0, 1]       # closing does (:78)
    for n in (1, 63, 64, 65, 255, 256, 257, 1023, 1025, 2049):            # ragged tile tails
        recs = O.gen_stream(n, seed=n, n_keys=50, variant=1)
        for v in (0, 1, 3, 7, 10):
            check_against_oracle(nf, O, recs, 1000, 1 << 30, ingest_variant=v)


def test_epochs_are_independent(nf, O):
    """After an eviction the table starts from nothing: two ticks = two independent folds."""
    a = O.gen_stream(5000, seed=1, n_keys=300, variant=1)
    b = O.gen_stream(5000, j0=5000, seed=1, n_keys=300, variant=1)
    with nf.FlowTable(max_entries=1000) as tab:
        for part in (a, b):
            assert tab.ingest(part.view(nf.FLOW_RECORD)) == (nf.OK, len(part))
            got = nf.sort_by_key(tab.evict(nf.REASON_TIMEOUT))
            want = O.run_accounter(part, 1000)[0][1]
            assert_records_equal(got, want)
            assert len(tab) == 0


def test_truncated_evict_keeps_state(nf, O):
    import ctypes as C
    recs = O.gen_stream(2000, seed=3, n_keys=100)
    with nf.FlowTable(max_entries=1000) as tab:
        tab.ingest(recs.view(nf.FLOW_RECORD))
        n = len(tab)
        out = np.zeros(10, dtype=nf.FLOW_RECORD)
        got = C.c_size_t(0)
        rc = nf._lib.lib.nfagg_evict(tab._h, nf.REASON_TIMEOUT, out.ctypes.data_as(C.c_void_p), 10, C.byref(got))
        assert rc == nf.TRUNCATED and got.value == n and len(tab) == n
        assert_records_equal(nf.sort_by_key(tab.evict()), O.run_accounter(recs, 1000)[0][1])


def test_staging_acquire_commit(nf, O):
    recs = O.gen_stream(5000, seed=21, n_keys=400, variant=1)
    with nf.FlowTable(max_entries=1000, staging_records=2048) as tab:
        off = 0
        while off < len(recs):
            buf = tab.staging_acquire()
            m = min(len(buf), len(recs) - off)
            buf[:m] = recs[off:off + m].view(nf.FLOW_RECORD)
            rc, c = tab.staging_commit(m)
            assert (rc, c) == (nf.OK, m)
            off += m
        assert_records_equal(nf.sort_by_key(tab.evict()), O.run_accounter(recs, 1000)[0][1])


def test_small_staging_ring_many_chunks(nf, O):
    recs = O.gen_stream(30000, seed=22, n_keys=2000, variant=1)
    check_against_oracle(nf, O, recs, 4096, 1 << 30, staging_records=1000)


@pytest.mark.parametrize("ingest_variant", [0, 1, 7, 10, 17])
@pytest.mark.parametrize("n_shards", [2, 8])
def test_sharded_tables_cover_the_stream(nf, O, n_shards, ingest_variant):
    """Records shard by flow-key hash; every shard folds only its own keys; the
    concatenation of the shards' evictions equals the unsharded result."""
    th = O.zipf_thresholds(2000, 1.1)
    recs = O.gen_stream(30000, seed=8, n_keys=2000, thresholds=th, variant=1)
    parts, skipped = [], 0
    for s in range(n_shards):
        with nf.FlowTable(max_entries=4096, n_shards=n_shards, shard_id=s, ingest_variant=ingest_variant) as tab:
            assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
            ev = tab.evict()
            assert all(nf.shard_of(r["id"].tobytes(), n_shards) == s for r in ev[:50])
            skipped += tab.stats().records_skipped
            parts.append(ev)
    assert skipped == (n_shards - 1) * len(recs)
    got = nf.sort_by_key(np.concatenate(parts))
    assert_records_equal(got, O.run_accounter(recs, 1 << 20)[0][1])


def test_idempotent_refold_of_evicted_records(nf, O):
    """Round trip: an evicted batch fed back in (one record per key) evicts unchanged."""
    recs = O.gen_stream(20000, seed=31, n_keys=1500, variant=1)
    with nf.FlowTable(max_entries=4096) as tab:
        tab.ingest(recs.view(nf.FLOW_RECORD))
        first = nf.sort_by_key(tab.evict())
        tab.ingest(first)
        again = nf.sort_by_key(tab.evict())
        assert_records_equal(again, first)


def _np_key_hash(words):
    """DESIGN.md §5 key hash, vectorised (uint64 wrap-around arithmetic)."""
    mul, seed = np.uint64(0x9E3779B97F4A7C15), np.uint64(0x6E66616767206B31)
    h = np.full(len(words), seed, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(5):
            h = (((h << np.uint64(27)) | (h >> np.uint64(37))) ^ words[:, i]) * mul
        h ^= h >> np.uint64(33); h *= np.uint64(0xff51afd7ed558ccd)
        h ^= h >> np.uint64(33); h *= np.uint64(0xc4ceb9fe1a85ec53)
        h ^= h >> np.uint64(33)
    return h


def test_partition_skew_overflows_the_spill_queues(nf, O):
    """Two-pass ingest, adversarial case: thousands of flows whose key hashes fall into ONE spill
    partition. The partition queue overflows into the overflow list (third pass), staging groups
    fill up within a tile (carry path) — the result must still be bit-exact."""
    c = np.arange(6_000_000, dtype=np.uint64)
    ids = np.zeros((len(c), 5), dtype=np.uint64)
    ids[:, 0] = c * np.uint64(0x10001) + np.uint64(7)
    ids[:, 3] = c ^ np.uint64(0xabcdef)
    ids[:, 4] = c & np.uint64(0xffffff)                      # byte 39 stays zero
    h = _np_key_hash(ids)
    assert int(h[12345]) == nf.key_hash(ids[12345].tobytes())
    sel = ids[((h >> np.uint64(29)) & np.uint64(2047)) == 7][:2500]
    assert len(sel) == 2500
    n = 200_000
    th = O.zipf_thresholds(2500, 0.3)                         # nearly uniform: most flows miss the LDS caches
    recs = O.gen_stream(n, seed=41, n_keys=2500, thresholds=th, variant=1)
    recs["id"]["pad"] = 0                                     # variant 1 dirties byte 39; it is not part of the identity
    raw = np.ascontiguousarray(recs["id"]).view(np.dtype((np.void, 40)))
    _, inverse = np.unique(raw, return_inverse=True)
    recs["id"] = sel[inverse.ravel()].view(O.FLOW_ID).ravel()
    want = O.run_accounter(recs, 1 << 20)[0][1]
    assert len(want) > 2400
    with nf.FlowTable(max_entries=1 << 20, ingest_variant=10) as tab:
        assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, n)
        st = tab.stats()
        assert st.records_bypassed > 2 * (2 * n // 2048 + 1024), "the stream was meant to overflow one partition queue"
        assert_records_equal(nf.sort_by_key(tab.evict()), want)


@pytest.mark.parametrize("mode", ["accounter", "dedup"])
def test_large_table_evicts_in_slot_order(nf, O, mode):
    """A table of 2^24 slots or more with 65 536+ live flows takes the slot-sorted eviction (launch_sort_slots):
    same flows, bit-exact, two epochs so that the zeroing is checked too."""
    th = O.zipf_thresholds(150_000, 0.6)
    recs = O.gen_stream(400_000, seed=31, n_keys=150_000, thresholds=th, variant=2 if mode == "dedup" else 1)
    omode = 1 if mode == "dedup" else 0
    want = O.run_accounter(recs, 1 << 23, omode)[0][1]
    assert len(want) >= 1 << 16
    kw = dict(mode=nf.MODE_KERNEL_DEDUP) if mode == "dedup" else {}
    with nf.FlowTable(max_entries=1 << 23, **kw) as tab:
        assert tab.stats().table_slots >= 1 << 24
        for _ in range(2):
            assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
            got = nf.sort_by_key(tab.evict(nf.REASON_TIMEOUT))
            assert_records_equal(got, want, mode)
            assert len(tab) == 0


def test_epoch_tags_wrap_after_65535_evictions(nf, O):
    """Eviction does not clear the table: slots carry their epoch in 16 tag bits, and the tags are cleared once when the
    epoch counter wraps. Flows left in the table by epoch e must not reappear in epoch e + 65 535."""
    a = O.gen_stream(3000, seed=71, n_keys=200, variant=1)
    b = O.gen_stream(3000, seed=72, n_keys=200, variant=1)
    with nf.FlowTable(max_entries=1000) as tab:
        assert tab.ingest(a.view(nf.FLOW_RECORD)) == (nf.OK, len(a))
        assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)), O.run_accounter(a, 1000)[0][1])
        for _ in range(65534):
            assert len(tab.evict(nf.REASON_CLOSING, cap=1)) == 0          # closing evicts (and starts an epoch) even when empty
        for part in (b, a, b):                                             # epoch numbers 1 (again), 2, 3
            assert tab.ingest(part.view(nf.FLOW_RECORD)) == (nf.OK, len(part))
            assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)), O.run_accounter(part, 1000)[0][1])


@pytest.mark.parametrize("ingest_variant", [10, 17])
def test_spilled_records_gathered_in_five_six_or_seven_units(nf, O, ingest_variant):
    """Pass 2 of the two-pass fold gathers only the 16-byte units of a spilled record that the fold needs (csrc/nfagg_ingest_part.hip
    queue_entry): the seventh (bytes 96..111) only when dscp is non-zero, the sixth (80..95) only when sampling is non-zero or
    dst_mac's first two bytes are zero while its last four are not — every combination here, with the MAC shapes the seeded streams
    never produce (00:00:5e:.. — VRRP: zero first bytes, set last ones; all zero; only the first byte set), so that "first non-zero
    dst_mac", "last non-zero sampling / dscp" (pkg/model/flow_content.go:48-59) are decided by records read in every one of the forms."""
    n, keys = 300_000, 60_000
    recs = O.gen_stream(n, seed=77, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), variant=0)
    rng = np.random.default_rng(ingest_variant)
    m = recs["metrics"]
    shapes = np.array([[0, 0, 0, 0, 0, 0], [0, 0, 0x5e, 0, 1, 7], [2, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 9], [0x0a, 0x0b, 0x0c, 0x0d, 0x0e, 0x0f]], dtype=np.uint8)
    m["dst_mac"] = shapes[rng.integers(0, len(shapes), n)]
    m["src_mac"] = shapes[rng.integers(0, len(shapes), n)]
    m["sampling"] = rng.choice(np.array([0, 0, 0, 50, 7], dtype=np.uint32), n)
    m["dscp"] = rng.choice(np.array([0, 0, 0, 46, 10], dtype=np.uint8), n)
    want = check_against_oracle(nf, O, recs, 1 << 17, 1 << 30, ingest_variant=ingest_variant)
    ev = want[0][1]["metrics"]
    assert (ev["sampling"] != 0).any() and (ev["dscp"] != 0).any() and (ev["dst_mac"][:, :2] == 0).all(axis=1).any()
