"""ingest_variant 31 of nfagg_account[_device] (opt-in; written at the end of round 4: these tests and one timing run are all the
GPU time it has had) — the evict-on-full loop of Accounter.Account (pkg/flow/account.go:81-96) with its epochs found first
(previous-occurrence links, tests/test_epoch_boundaries.py) and folded together (csrc/nfagg_epoch_par.hip,
csrc/nfagg_account_par.inc). Same contract as the default path: every eviction bit-identical, in order, to the oracle's."""
import numpy as np
import pytest

from conftest import assert_records_equal
from test_account_gpu import _check, _stream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_entries,keys,n", [(5000, 100_000, 600_000), (100, 3_000, 150_000), (2, 50, 80_000), (20_000, 400_000, 900_000),
                                                 (5000, 4_000, 300_000)])
def test_epochs_found_first_equal_the_reference_loop(nf, O, max_entries, keys, n):
    recs = _stream(O, n, keys, seed=7 + max_entries)
    with nf.FlowTable(max_entries=max_entries, ingest_variant=31) as tab:
        n_ev = _check(nf, O, tab, recs, max_entries, [n])
        if keys > max_entries:
            assert n_ev > 3
        st = tab.stats()
        assert st.records_ingested == n and st.evictions[nf.REASON_FULL] == n_ev - 1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_epochs_that_span_calls_and_a_hot_flow(nf, O, seed):
    rng = np.random.default_rng(seed)
    max_entries = int(rng.choice([7, 300, 5000]))
    recs = _stream(O, 700_000, int(rng.choice([2_000, 80_000])), seed=90 + seed, hot=int(rng.choice([0, 700])))
    batches = [int(rng.choice([1, 999, 90_000, 200_000, 300_000])) for _ in range(400)]
    with nf.FlowTable(max_entries=max_entries, ingest_variant=31) as tab:
        _check(nf, O, tab, recs, max_entries, batches)


def test_device_resident_call_and_small_output_room(nf, O):
    import torch
    max_entries = 2000
    recs = _stream(O, 500_000, 50_000, seed=5)
    want = O.run_accounter(recs, max_entries)
    d = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).cuda()
    with nf.FlowTable(max_entries=max_entries, ingest_variant=31) as tab:
        out = torch.zeros((len(recs) + max_entries) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        rc, c, ends = tab.account_device(d.data_ptr(), len(recs), out.data_ptr(), len(recs) + max_entries, 4096)
        assert (rc, c) == (nf.OK, len(recs)) and len(ends) == len(want) - 1
        ev = out.cpu().numpy()
        lo = 0
        for e, (_, w) in zip(ends, want):
            assert_records_equal(nf.sort_by_key(ev[lo * 144:e * 144].view(nf.FLOW_RECORD)), w)
            lo = e
        assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)), want[-1][1])
    # room for five evictions per call: NFAGG_TRUNCATED, the caller drains and calls again
    view = recs.view(nf.FLOW_RECORD)
    got, off = [], 0
    with nf.FlowTable(max_entries=max_entries, ingest_variant=31) as tab:
        while off < len(recs):
            rc, c, epochs = tab.account(view[off:], out_cap=5 * max_entries + 10, max_epochs=64)
            assert len(epochs) <= 5 and (rc == nf.TRUNCATED or off + c == len(recs))
            got += [nf.sort_by_key(e) for e in epochs]
            off += c
        got.append(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)))
    assert len(got) == len(want)
    for g, (_, w) in zip(got, want):
        assert_records_equal(g, w)
