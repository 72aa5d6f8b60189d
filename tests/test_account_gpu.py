"""nfagg_account / nfagg_account_device (include/nfagg.h): the record arm of Accounter.Account WITH its evictions on "full"
(pkg/flow/account.go:81-96) in one call — for small CACHE_MAX_FLOWS the whole loop runs on the device: calls of more than a few
epochs have their epochs found first (csrc/nfagg_epoch_par.hip, tests/test_account_par_gpu.py), the others — most of the calls
here — take the kernel chain (csrc/nfagg_epoch_chain.hip). Every eviction must be bit-identical, in order, to the oracle's Accounter driven the way
the reference's TestEvict_MaxEntries drives it (pkg/flow/account_test.go:47-128: the (maxEntries+1)-th distinct key flushes
exactly maxEntries flows)."""
import numpy as np
import pytest

from conftest import assert_records_equal

pytestmark = pytest.mark.gpu


def _stream(O, n, keys, seed, hot=0, variant=1):
    return O.gen_stream(n, seed=seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=hot, variant=variant)


def _check(nf, O, tab, recs, max_entries, batches, mode=0):
    """Feed recs in `batches` through tab.account; the evictions on full + the closing eviction = the oracle's, one by one."""
    want = O.run_accounter(recs, max_entries, mode)
    got = []
    off = 0
    view = recs.view(nf.FLOW_RECORD)
    for b in batches:
        hi = min(len(recs), off + b)
        while off < hi:
            rc, c, epochs = tab.account(view[off:hi])
            got += [("full", nf.sort_by_key(e)) for e in epochs]
            off += c
            assert rc in (nf.OK, nf.TRUNCATED)
        if off >= len(recs):
            break
    assert off == len(recs)
    got.append(("closing", nf.sort_by_key(tab.evict(nf.REASON_CLOSING))))
    assert [r for r, _ in got] == [r for r, _ in want], (len(got), len(want))
    for k, ((_, g), (_, w)) in enumerate(zip(got, want)):
        assert_records_equal(g, w, "eviction %d of %d" % (k, len(want)))
    return len(want)


@pytest.mark.parametrize("variant", [0, 30])      # 0: the default (by call size: epochs found first / the kernel chain); 30: the kernel chain always
@pytest.mark.parametrize("max_entries,keys,n", [(5000, 100_000, 400_000), (100, 3_000, 60_000), (2, 50, 3_000), (20_000, 400_000, 500_000),
                                                 (5000, 4_000, 100_000)])
def test_account_equals_the_reference_loop(nf, O, max_entries, keys, n, variant):
    """Whole stream in one call (the chain's windows are 16 384 records; epochs from 3 records to longer than a window;
    a map that never fills)."""
    recs = _stream(O, n, keys, seed=7 + max_entries)
    with nf.FlowTable(max_entries=max_entries, ingest_variant=variant) as tab:
        n_ev = _check(nf, O, tab, recs, max_entries, [n])
        if keys > max_entries:
            assert n_ev > 3
        st = tab.stats()
        assert st.records_ingested == n and st.evictions[nf.REASON_FULL] == n_ev - 1


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_account_ragged_batches_hot_flows_and_sketches(nf, O, seed):
    """Batch cuts anywhere (an epoch spans calls: its slots are finalized by one call and evicted by a later one), a hot flow,
    the sketches folded along (they see every record exactly once, whatever the epochs)."""
    rng = np.random.default_rng(seed)
    max_entries = int(rng.choice([7, 300, 5000]))
    recs = _stream(O, 250_000, int(rng.choice([2_000, 80_000])), seed=90 + seed, hot=int(rng.choice([0, 700])))
    batches = [int(rng.choice([1, 5, 999, 16_384, 16_385, 40_000, 100_000])) for _ in range(400)]
    with nf.FlowTable(max_entries=max_entries, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=12, hll_p=8,
                      ingest_variant=30 if seed == 4 else 0) as tab:
        _check(nf, O, tab, recs, max_entries, batches)
        cs, cd, hs, hd = O.sketches(recs, 4, 12, 8)
        assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cs) and np.array_equal(tab.sketch_snapshot(nf.CM_DST), cd)
        assert np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs) and np.array_equal(tab.sketch_snapshot(nf.HLL_DST), hd)


def test_account_mixes_with_ingest_and_evict(nf, O):
    """nfagg_ingest / nfagg_evict and nfagg_account on one handle: the epoch in progress is the same object for both."""
    max_entries = 1000
    recs = _stream(O, 120_000, 30_000, seed=3)
    want = O.run_accounter(recs, max_entries)
    view = recs.view(nf.FLOW_RECORD)
    got = []
    with nf.FlowTable(max_entries=max_entries) as tab:
        off = 0
        k = 0
        while off < len(recs):
            hi = min(len(recs), off + 7_000)
            if k % 2 == 0:
                while off < hi:
                    rc, c, epochs = tab.account(view[off:hi])
                    got += [nf.sort_by_key(e) for e in epochs]
                    off += c
            else:
                while off < hi:
                    rc, c = tab.ingest(view[off:hi])
                    off += c
                    if rc == nf.FULL:
                        got.append(nf.sort_by_key(tab.evict(nf.REASON_FULL)))
            k += 1
        got.append(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)))
    assert len(got) == len(want)
    for g, (_, w) in zip(got, want):
        assert_records_equal(g, w)


def test_account_stops_when_the_output_is_full_and_resumes(nf, O):
    max_entries = 500
    recs = _stream(O, 60_000, 20_000, seed=11)
    want = O.run_accounter(recs, max_entries)
    view = recs.view(nf.FLOW_RECORD)
    got, off, calls = [], 0, 0
    with nf.FlowTable(max_entries=max_entries) as tab:
        while off < len(recs):
            rc, c, epochs = tab.account(view[off:], out_cap=3 * max_entries + 10, max_epochs=64)     # room for three evictions per call
            assert len(epochs) <= 3 and (rc == nf.TRUNCATED or off + c == len(recs))
            got += [nf.sort_by_key(e) for e in epochs]
            off += c
            calls += 1
        got.append(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)))
    assert calls > 5 and len(got) == len(want)
    for g, (_, w) in zip(got, want):
        assert_records_equal(g, w)


def test_account_device_and_sharded_handle(nf, O):
    """Device-resident variant; a handle that filters by shard skips the other shards' records inside the kernel."""
    import torch
    max_entries = 2000
    recs = _stream(O, 200_000, 50_000, seed=5)
    d = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).cuda()
    for n_shards, shard in ((1, 0), (3, 1)):
        mine = recs if n_shards == 1 else recs[nf.distributed.shard_ids(recs.view(nf.FLOW_RECORD), n_shards) == shard]
        want = O.run_accounter(mine, max_entries)
        with nf.FlowTable(max_entries=max_entries, n_shards=n_shards, shard_id=shard) as tab:
            out = torch.zeros((len(recs) + max_entries) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            rc, c, ends = tab.account_device(d.data_ptr(), len(recs), out.data_ptr(), len(recs) + max_entries, 4096)
            assert (rc, c) == (nf.OK, len(recs)) and len(ends) == len(want) - 1
            ev = out.cpu().numpy()
            lo = 0
            for e, (_, w) in zip(ends, want):
                assert_records_equal(nf.sort_by_key(ev[lo * 144:e * 144].view(nf.FLOW_RECORD)), w)
                lo = e
            assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)), want[-1][1])


def test_account_dedup_mode_and_large_tables_take_the_host_loop(nf, O):
    """Modes the persistent kernel does not cover go through the library's own ingest / evict loop: same contract."""
    from conftest import dedup_stream
    recs = dedup_stream(O, 80_000, seed=8, n_keys=9_000, thresholds=O.zipf_thresholds(9_000, 1.1), style=1)
    with nf.FlowTable(max_entries=1500, mode=nf.MODE_KERNEL_DEDUP) as tab:
        _check(nf, O, tab, recs, 1500, [33_000] * 5, mode=1)
    recs = _stream(O, 300_000, 200_000, seed=13)
    with nf.FlowTable(max_entries=40_000) as tab:
        _check(nf, O, tab, recs, 40_000, [120_000] * 5)


def test_page_locked_caller_buffers_take_the_direct_path(nf, O):
    """Records in / evictions into page-locked memory (nfagg_host_alloc): DMA straight from / into the caller's buffers, no copy
    through the staging ring — nfagg_account, nfagg_ingest and nfagg_evict deliver what they deliver from pageable arrays."""
    max_entries = 3000
    recs = _stream(O, 300_000, 60_000, seed=21)
    want = O.run_accounter(recs, max_entries)
    with nf.PinnedRecords(len(recs)) as pin, nf.PinnedRecords(len(recs) + max_entries) as pout:
        pin.records[:] = recs.view(nf.FLOW_RECORD)
        with nf.FlowTable(max_entries=max_entries, staging_records=50_000) as tab:
            rc, c, epochs = tab.account(pin.records, out=pout.records)
            assert (rc, c) == (nf.OK, len(recs)) and len(epochs) == len(want) - 1
            for e, (_, w) in zip(epochs, want):
                assert_records_equal(nf.sort_by_key(e.copy()), w)
            last = tab.evict(nf.REASON_CLOSING, out=pout.records)
            assert_records_equal(nf.sort_by_key(last.copy()), want[-1][1])
        # nfagg_ingest + a large nfagg_evict (above the 4 MiB the bounce path starts at) through the same buffers
        want1 = O.run_accounter(recs, 1 << 20)[0][1]
        with nf.FlowTable(max_entries=1 << 20, staging_records=50_000) as tab:
            assert tab.ingest(pin.records) == (nf.OK, len(recs))
            got = tab.evict(nf.REASON_TIMEOUT, out=pout.records)
            assert_records_equal(nf.sort_by_key(got.copy()), want1)
            got2 = tab.evict(nf.REASON_TIMEOUT)                    # nothing since
            assert len(got2) == 0
