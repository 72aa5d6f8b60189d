"""The sequence window (csrc/nfagg_rebase.hip). The reference's Accounter has no maximum epoch length (pkg/flow/account.go:58-100);
the parallel fold orders records by sequence numbers that the slots carry as 32-bit, window-relative tags. When ~2^32 records
of an epoch have gone by the window MOVES — the tags in the table are rebased — instead of the epoch being evicted early. The
order-dependent fields (flow_content.go:45-59: last non-zero eth/dscp/sampling, first non-zero MACs; account.go:95: first
record stored whole; bpf/flows.c:76-143 in kernel-dedup mode) must come out exactly as from ONE sequential Accounter over all
the records, wherever the window boundaries fall. nfagg_debug_skip_sequence moves the position without feeding 600 GB."""
import numpy as np
import pytest

from conftest import assert_records_equal, dedup_stream

pytestmark = pytest.mark.gpu
WINDOW = 0xFFFFFFF0


def _stream(O, n, keys, seed, hot=0):
    return O.gen_stream(n, seed=seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=hot, variant=1)


@pytest.mark.parametrize("mode,batch", [(0, 7_000), (0, 400_000), (1, 9_000), (1, 90_000)])
def test_one_handle_across_several_windows(nf, O, mode, batch):
    """Records before, across and after three window moves; every ingest kernel family by batch size (direct / cached /
    two-pass; kernel-dedup direct / cached). Scrambled records: every order-dependent field is exercised."""
    n = 900_000 if batch > 100_000 else 120_000
    if mode == 0:
        recs = _stream(O, n, 20_000, seed=61, hot=300)
    else:
        recs = dedup_stream(O, n, seed=62, n_keys=6_000, thresholds=O.zipf_thresholds(6_000, 1.1), style=2)
    view = recs.view(nf.FLOW_RECORD)
    with nf.FlowTable(max_entries=1 << 17, mode=mode) as tab:
        off, skipped = 0, 0
        jumps = {n // 5: WINDOW - 1000, n // 2: 3 * WINDOW + 12345, 4 * n // 5: WINDOW - 7}      # where the position leaps
        while off < n:
            hi = min(n, off + batch)
            for at in sorted(jumps):
                if off < at < hi:
                    hi = at
            assert tab.ingest(view[off:hi]) == (nf.OK, hi - off)
            off = hi
            if off in jumps:
                tab.debug_skip_sequence(jumps[off])
                skipped += jumps[off]
        st = tab.stats()
        assert st.sequence_rebases >= 3 and st.epoch_seq == n + skipped
        assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)), O.run_accounter(recs, 1 << 20, mode)[0][1])


def test_window_boundary_inside_a_batch_and_with_evictions_on_full(nf, O):
    """The boundary falls inside a batch (the library cuts nothing: it moves the window before the batch would cross it), the
    stream also stops on full now and then; account (persistent epoch kernel) and ingest/evict alternate."""
    recs = _stream(O, 300_000, 30_000, seed=63)
    want = O.run_accounter(recs, 4_000)
    view = recs.view(nf.FLOW_RECORD)
    got = []
    with nf.FlowTable(max_entries=4_000) as tab:
        off, k = 0, 0
        while off < len(recs):
            hi = min(len(recs), off + 11_000)
            tab.debug_skip_sequence(WINDOW - 5_000)                    # every batch begins 5000 numbers before a boundary
            while off < hi:
                if k % 2:
                    rc, c, epochs = tab.account(view[off:hi])
                    got += [nf.sort_by_key(e) for e in epochs]
                else:
                    rc, c = tab.ingest(view[off:hi])
                    if rc == nf.FULL:
                        got.append(nf.sort_by_key(tab.evict(nf.REASON_FULL)))
                off += c
            k += 1
        got.append(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)))
    assert len(got) == len(want)
    for g, (_, w) in zip(got, want):
        assert_records_equal(g, w)


@pytest.mark.parametrize("n_members", [2, 8])
def test_local_fold_group_past_the_window(nf, O, n_members):
    """More than 2^32 sequence numbers in ONE epoch of a local-fold group: the members' flows are brought together at their
    owners, the common window moves, folding goes on on every member — and the eviction is bit-identical to ONE Accounter over
    all the records."""
    import torch
    recs = _stream(O, 480_000, 25_000, seed=64, hot=400)
    with nf.FlowGroup([0] * n_members, max_entries=1 << 18, local_fold=True) as grp:
        rng = np.random.default_rng(n_members)
        off, keep, part = 0, [], len(recs) // 6
        for phase in range(6):
            end = len(recs) if phase == 5 else off + part
            while off < end:
                c = min(end - off, int(rng.choice([1, 500, 9_000, 40_000])))
                d = torch.from_numpy(recs[off:off + c].view(np.uint8).reshape(-1).copy()).cuda()
                keep.append(d)
                m = int(rng.integers(0, n_members))
                assert grp.ingest_device(m, d.data_ptr(), c) == (nf.OK, c)
                off += c
            if phase in (1, 3):                                        # leap: the next chunk does not fit the window any more
                grp.members[0].sync()
                grp.debug_skip_sequence(WINDOW - 3_000)
        want = O.run_accounter(recs, 1 << 20)[0][1]
        got = nf.sort_by_key(grp.evict(nf.REASON_TIMEOUT))
        assert_records_equal(got, want)
        assert sum(m.stats().sequence_rebases for m in grp.members) >= 2 * n_members
        assert grp.ingest(recs.view(nf.FLOW_RECORD)[:10]) == (nf.OK, 10)   # next epoch: sequence and window start over


def test_ranks_restart_their_window_together(nf, O):
    """One process per GPU (the handles stand for ranks): an externally sequenced handle refuses to fold past its window
    (NFAGG_ERANGE); export-all -> exchange -> nfagg_window_restart_device moves every rank's window without an eviction."""
    import torch
    n_ranks = 4
    recs = _stream(O, 240_000, 12_000, seed=65)
    half = len(recs) // 2
    per = half // n_ranks
    tabs = [nf.FlowTable(max_entries=1 << 16, table_log2_slots=18) for _ in range(n_ranks)]
    exp = [torch.zeros((1 << 16) * 24, dtype=torch.int64, device="cuda") for _ in range(n_ranks)]
    torch.cuda.synchronize()
    keep = []

    def fold(r, lo, hi, seq):
        d = torch.from_numpy(recs[lo:hi].view(np.uint8).reshape(-1).copy()).cuda()
        keep.append(d)
        tabs[r].set_sequence(seq)
        return tabs[r].ingest_device(d.data_ptr(), hi - lo)
    try:
        for r in range(n_ranks):
            assert fold(r, r * per, (r + 1) * per, r * per) == (nf.OK, per)
        leap = WINDOW - 100                                           # the job's position leaps: no rank's window has room any more
        with pytest.raises(nf.NfaggError) as ei:
            fold(0, half, half + per, leap)
        assert ei.value.code == -6 and "nfagg_window_restart_device" in str(ei.value)
        # every rank: export ALL its flows grouped by owner; "exchange" (one device here); restart at the common position
        counts = []
        for r in range(n_ranks):
            rc, c, total = tabs[r].partials_export_device(n_ranks, 0xFFFFFFFF, exp[r].data_ptr(), 1 << 16)
            assert rc == nf.OK
            counts.append(c)
        for owner in range(n_ranks):
            segs = [exp[src][sum(counts[src][:owner]) * 24:(sum(counts[src][:owner]) + counts[src][owner]) * 24] for src in range(n_ranks)]
            mine = torch.cat(segs)
            keep.append(mine)
            torch.cuda.synchronize()
            tabs[owner].window_restart_device(n_ranks, owner, mine.data_ptr(), mine.numel() // 24, leap)
            assert 0 < len(tabs[owner]) <= mine.numel() // 24           # one slot per flow this rank owns, however many ranks had seen it
        for r in range(n_ranks):
            assert fold(r, half + r * per, half + (r + 1) * per, leap + r * per) == (nf.OK, per)
        # the tick: export (own flows stay) -> merge -> evict owned
        counts = []
        for r in range(n_ranks):
            rc, c, total = tabs[r].partials_export_device(n_ranks, r, exp[r].data_ptr(), 1 << 16)
            assert rc == nf.OK
            counts.append(c)
        for owner in range(n_ranks):
            for src in range(n_ranks):
                if src != owner and counts[src][owner]:
                    tabs[owner].partials_merge_device(n_ranks, owner, exp[src].data_ptr() + sum(counts[src][:owner]) * 192, counts[src][owner])
        out = []
        for r in range(n_ranks):
            rc, need = tabs[r].evict_owned_device(n_ranks, r, 0, 0)
            buf = torch.zeros(max(need, 1) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            if need:
                assert tabs[r].evict_owned_device(n_ranks, r, buf.data_ptr(), need) == (nf.OK, need)
            out.append(buf.cpu().numpy()[: need * 144].view(nf.FLOW_RECORD))
        used = np.concatenate([recs[:per * n_ranks], recs[half:half + per * n_ranks]])
        assert_records_equal(nf.sort_by_key(np.concatenate(out)), O.run_accounter(used, 1 << 20)[0][1])
    finally:
        for t in tabs:
            t.close()
