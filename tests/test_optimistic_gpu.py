"""The optimistic fold (csrc/nfagg_api.hip: fold_optimistic): a batch with live + batch > max_entries is folded whole and
checked afterwards; a batch that did cross max_entries (account.go:85-94) is rolled back and split at the exact record.
Everything against the oracle's sequential Accounter: same eviction sequence, every evicted record bit-exact."""
import numpy as np
import pytest

from conftest import assert_records_equal, dedup_stream
from test_parity_gpu import check_against_oracle, drive_product
from test_dedup_gpu import check_dedup

pytestmark = pytest.mark.gpu


def _zipf(O, n, keys, seed, **kw):
    return O.gen_stream(n, seed=seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), variant=1, **kw)


@pytest.mark.parametrize("max_entries,batch", [(30_000, 1 << 30), (30_000, 700_001), (5_000, 1 << 30), (40_000, 250_000)])
def test_large_batches_split_exactly(nf, O, max_entries, batch):
    """2 M records over 100 k flows with a table for fewer: several evictions on full inside every call."""
    recs = _zipf(O, 2_000_000, 100_000, seed=21)
    with nf.FlowTable(max_entries=max_entries) as tab:
        got = drive_product(tab, recs.view(nf.FLOW_RECORD), batch)
        st = tab.stats()
    want = O.run_accounter(recs, max_entries)
    assert [r for r, _ in got] == [r for r, _ in want]
    assert sum(1 for r, _ in want if r == "full") >= 3
    for k, ((_, g), (_, w)) in enumerate(zip(got, want)):
        assert_records_equal(g, w, f"eviction #{k}")
    assert st.optimistic_rollbacks >= 3 and st.optimistic_folds >= st.optimistic_rollbacks


def test_no_overflow_is_one_fold(nf, O):
    """live + batch > max_entries but the distinct keys fit: ONE optimistic fold for the whole host call (four staging buffers folded
    without a look at the device in between, the proof — n_live <= max_entries — read once after the last: round 5), no rollback,
    table as the oracle's; the same call from device memory is one launch and one fold."""
    recs = _zipf(O, 4_000_000, 50_000, seed=22)
    with nf.FlowTable(max_entries=60_000, profile=True) as tab:
        assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        st = tab.stats()
        assert (st.optimistic_folds, st.optimistic_rollbacks, st.ingest_launches) == (1, 0, 4)
        got = nf.sort_by_key(tab.evict(nf.REASON_CLOSING))
    assert_records_equal(got, O.run_accounter(recs, 60_000)[0][1])


def test_rollback_restores_the_flows_that_were_live(nf, O):
    """Epoch state before the overflowing batch (every order-dependent field, sequence tags) must survive the rollback."""
    a = _zipf(O, 300_000, 20_000, seed=23)
    b = _zipf(O, 900_000, 200_000, seed=24)
    recs = np.concatenate([a, b])
    for me in (25_000, 60_000):
        want = O.run_accounter(recs, me)
        with nf.FlowTable(max_entries=me, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=14, hll_p=10) as tab:
            assert tab.ingest(a.view(nf.FLOW_RECORD)) == (nf.OK, len(a))
            got = drive_product(tab, b.view(nf.FLOW_RECORD), 1 << 30)
            assert tab.stats().optimistic_rollbacks >= 1
            cm = tab.sketch_snapshot(nf.CM_SRC)
            hll = tab.sketch_snapshot(nf.HLL_DST)
        assert [r for r, _ in got] == [r for r, _ in want]
        for k, ((_, g), (_, w)) in enumerate(zip(got, want)):
            assert_records_equal(g, w, f"max_entries {me}, eviction #{k}")
        cs, _, _, hd = O.sketches(recs, 4, 14, 10)          # every record counted exactly once, rolled-back folds not at all
        assert np.array_equal(cm, cs) and np.array_equal(hll, hd)


def test_more_new_keys_than_the_table_has_slots(nf, O):
    """max_entries 100 -> a 65 536-slot table; 400 k distinct keys in one call: the kernels refuse claims beyond the
    claim limit (aborted), the batch is rolled back and retried shorter until the split is found."""
    n = 400_000
    recs = O.gen_stream(n, seed=25, n_keys=n, variant=1)
    recs["id"]["src_port"] = np.arange(n) & 0xffff
    recs["id"]["dst_port"] = np.arange(n) >> 16          # all keys distinct
    with nf.FlowTable(max_entries=100) as tab:
        rc, c = tab.ingest(recs.view(nf.FLOW_RECORD))
        assert (rc, c) == (nf.FULL, 100)
        ev = tab.evict(nf.REASON_FULL)
        assert len(ev) == 100
        assert_records_equal(nf.sort_by_key(ev), O.run_accounter(recs[:100], 1000)[0][1])
        assert tab.stats().optimistic_rollbacks >= 2
        rc, c = tab.ingest(recs[100:].view(nf.FLOW_RECORD))
        assert (rc, c) == (nf.FULL, 100)


@pytest.mark.parametrize("style", [1, 2])
def test_dedup_mode_large_batches_split_exactly(nf, O, style):
    th = O.zipf_thresholds(60_000, 1.1)
    recs = dedup_stream(O, 1_000_000, seed=26 + style, n_keys=60_000, thresholds=th, style=style)
    want = check_dedup(nf, O, recs, 20_000, 1 << 30)
    assert sum(1 for r, _ in want if r == "full") >= 2


def test_epoch_goes_on_past_the_sequence_window(nf, O):
    """An epoch has no maximum length (account.go:58-100). The slots' sequence tags are 32 bits wide and window-relative; round 2
    ended the epoch with NFAGG_FULL when 2^32-16 records had gone by — now the window moves (tests/test_sequence_window_gpu.py
    has the thorough cases) and the eviction is exactly one Accounter's over all the records."""
    recs = _zipf(O, 50_000, 500, seed=27)
    with nf.FlowTable(max_entries=1 << 16) as tab:
        assert tab.ingest(recs[:10_000].view(nf.FLOW_RECORD)) == (nf.OK, 10_000)
        tab.debug_skip_sequence(0xFFFFFFF0 - 10_000 - 1 - 25_000)          # 25 000 sequence numbers left in the window
        assert tab.ingest(recs[10_000:].view(nf.FLOW_RECORD)) == (nf.OK, 40_000)
        assert tab.stats().sequence_rebases == 1
        assert_records_equal(nf.sort_by_key(tab.evict()), O.run_accounter(recs, 1 << 16)[0][1])


def test_two_pass_fold_with_more_new_keys_than_the_table_has_slots(nf, O):
    """A call large enough for the two-pass fold (>= 768 Ki records) of all-distinct keys into a 65 536-slot table: pass 2 must
    count its claims on the spot here (deferred, per-workgroup counting could fill a table this small), claims beyond the
    limit are refused, the call is rolled back and retried shorter until the split is found."""
    n = 1_500_000
    recs = O.gen_stream(n, seed=41, n_keys=n, variant=0)
    recs["id"]["src_port"] = np.arange(n) & 0xffff
    recs["id"]["dst_port"] = np.arange(n) >> 16
    with nf.FlowTable(max_entries=5000) as tab:
        rc, c = tab.ingest(recs.view(nf.FLOW_RECORD))
        assert (rc, c) == (nf.FULL, 5000)
        assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_FULL)), O.run_accounter(recs[:5000], 1 << 20)[0][1])


def test_stream_that_keeps_stopping_on_full_uses_its_epoch_length(nf, O):
    """After the first stop the library sizes its chunks by the length of the last epoch: most folds fit, one short chunk per
    epoch crosses max_entries and is rolled back."""
    recs = _zipf(O, 3_000_000, 400_000, seed=42)
    with nf.FlowTable(max_entries=50_000) as tab:
        got = drive_product(tab, recs.view(nf.FLOW_RECORD), 1 << 30)
        st = tab.stats()
    want = O.run_accounter(recs, 50_000)
    assert [(r, len(b)) for r, b in got] == [(r, len(b)) for r, b in want]
    for k, ((_, g), (_, w)) in enumerate(zip(got, want)):
        assert_records_equal(g, w, f"eviction #{k}")
    n_full = sum(1 for r, _ in want if r == "full")
    assert n_full >= 5 and st.optimistic_rollbacks <= n_full + 2 and st.optimistic_folds >= 2 * n_full - 2
