"""Kernel-dedup mode (bpf/flows.c:76-143) under LOCAL FOLD across GPUs — BASELINE configs[4] as it is written ("dedup on, 8 GPU").

What a table counts for a flow depends on the flow's first interface F (flows.c:100-126), and when several GPUs fold parts of ONE
stream F is the interface of the earliest record ANYWHERE. Ranks of a local-fold job (nfagg_config.local_fold) therefore key
their tables by the SUB-FLOW (flow, interface), exchange sub-flow partials (256 bytes) at the tick and the owner of a flow JOINS its
sub-flows (csrc/nfagg_dedup_join.hip). Rehearsed on ONE GPU: several handles / group members on device 0 stand for the GPUs. The
union of the evictions must be bit-identical to ONE sequential kernel-dedup table over the whole stream — the oracle in mode 1,
itself pinned to the reference's own C (oracle/_ref, tests/test_oracle_ref.py)."""
import numpy as np
import pytest

from conftest import assert_records_equal, dedup_stream
from test_parity_gpu import drive_product

pytestmark = pytest.mark.gpu


def _want(O, recs):
    return O.run_accounter(recs, 1 << 22, mode=1)[0][1]


# ---------------------------------------------------------------- one sub-flow table alone: fold + join = one kernel-dedup table
@pytest.mark.parametrize("ingest_variant", [1, 10])   # 1 = direct per-record passes, 10 = streaming + partition passes
@pytest.mark.parametrize("style", [0, 1, 2, 3])
@pytest.mark.parametrize("batch", [1 << 30, 1000, 257])
def test_subflow_table_alone_equals_the_flow_table(nf, O, style, batch, ingest_variant):
    """Every style of tests/test_dedup_gpu.py through a handle created with local_fold: sub-flow slots accumulate over the
    batches, the eviction joins them. (max_entries counts (flow, interface) pairs there: large enough not to fill.)"""
    th = O.zipf_thresholds(300, 1.1)
    recs = dedup_stream(O, 40000, seed=200 + style, n_keys=300, thresholds=th, style=style)
    with nf.FlowTable(max_entries=1 << 16, mode=nf.MODE_KERNEL_DEDUP, local_fold=True, ingest_variant=ingest_variant) as tab:
        assert tab.partial_bytes == 256
        got = drive_product(tab, recs.view(nf.FLOW_RECORD), batch)
    assert [r for r, _ in got] == ["closing"]
    assert_records_equal(got[0][1], _want(O, recs), "style %d" % style)


def test_subflow_table_many_flows_and_epochs(nf, O):
    """More sub-flows than the workgroup caches hold, three epochs on one handle (the join table is reused by its epoch tag)."""
    th = O.zipf_thresholds(30000, 1.1)
    with nf.FlowTable(max_entries=1 << 18, mode=nf.MODE_KERNEL_DEDUP, local_fold=True) as tab:
        for epoch, style in enumerate((2, 1, 2)):
            recs = dedup_stream(O, 300000, seed=17 + epoch, n_keys=30000, thresholds=th, style=style)
            view = recs.view(nf.FLOW_RECORD)
            for lo in range(0, len(recs), 100_000):
                assert tab.ingest(view[lo:lo + 100_000]) == (nf.OK, min(100_000, len(recs) - lo))
            assert len(tab) >= len(_want(O, recs))                              # (flow, interface) pairs: an upper bound of the flows
            assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_TIMEOUT)), _want(O, recs), "epoch %d" % epoch)
            assert len(tab) == 0
        assert len(tab.evict(nf.REASON_TIMEOUT)) == 0                           # nothing folded since: the timeout arm does not evict


def test_subflow_table_full_stop_conserves_every_record(nf, O):
    """NFAGG_FULL comes when a NEW (flow, interface) pair finds max_entries of them — earlier than a flow count would — and every
    eviction still equals one kernel-dedup table over exactly the records consumed since the last one."""
    th = O.zipf_thresholds(400, 1.1)
    recs = dedup_stream(O, 12000, seed=77, n_keys=400, thresholds=th, style=2)
    view = recs.view(nf.FLOW_RECORD)
    for max_entries, batch in ((50, 333), (250, 4096), (300, 1 << 30)):
        with nf.FlowTable(max_entries=max_entries, mode=nf.MODE_KERNEL_DEDUP, local_fold=True) as tab:
            off, start, epochs = 0, 0, 0
            while off < len(recs):
                rc, c = tab.ingest(view[off:off + batch])
                off += c
                if rc == nf.FULL:
                    assert len(tab) == max_entries
                    got = nf.sort_by_key(tab.evict(nf.REASON_FULL))
                    assert_records_equal(got, _want(O, recs[start:off]), "max_entries %d, epoch %d" % (max_entries, epochs))
                    start, epochs = off, epochs + 1
            assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)), _want(O, recs[start:]))
            assert epochs >= 2


def test_subflow_table_truncated_eviction_is_repeatable(nf, O):
    recs = dedup_stream(O, 30000, seed=31, n_keys=900, style=1)
    want = _want(O, recs)
    with nf.FlowTable(max_entries=1 << 14, mode=nf.MODE_KERNEL_DEDUP, local_fold=True) as tab:
        assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        small = np.zeros(7, dtype=nf.FLOW_RECORD)
        n = __import__("ctypes").c_size_t(0)
        from netobserv_ebpf_agent_amd import _lib as L
        assert L.lib.nfagg_evict(tab._h, nf.REASON_TIMEOUT, small.ctypes.data, 7, __import__("ctypes").byref(n)) == nf.TRUNCATED
        assert n.value == len(want)                                              # flows (known from the join), not sub-flows
        # more records after the refused eviction: the join that was made is void
        assert tab.ingest(recs.view(nf.FLOW_RECORD)[:5000]) == (nf.OK, 5000)
        got = nf.sort_by_key(tab.evict(nf.REASON_TIMEOUT))
        assert_records_equal(got, _want(O, np.concatenate([recs, recs[:5000]])))


# ---------------------------------------------------------------- ranks: nfagg_partials_* on sub-flow tables
class Ranks:
    """N unsharded kernel-dedup handles on cuda:0 (local_fold) + the buffers a rank of bench.py holds."""

    def __init__(self, nf, n, max_entries=1 << 18, **kw):
        import torch
        self.nf, self.n, self.torch = nf, n, torch
        self.tabs = [nf.FlowTable(max_entries=max_entries, table_log2_slots=20, mode=nf.MODE_KERNEL_DEDUP, local_fold=True, **kw) for _ in range(n)]
        self.pb = self.tabs[0].partial_bytes
        assert self.pb == 256
        self.exp = [torch.zeros(max_entries * self.pb // 8, dtype=torch.int64, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        self.keep = []

    def close(self):
        for t in self.tabs:
            t.close()

    def fold(self, rank, recs, seq):
        d = self.torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(-1).copy()).cuda()
        self.torch.cuda.synchronize()            # the upload runs on torch's stream, the fold on the library's
        self.keep.append(d)                      # the fold is asynchronous
        self.tabs[rank].set_sequence(seq)
        rc, c = self.tabs[rank].ingest_device(d.data_ptr(), len(recs))
        assert (rc, c) == (self.nf.OK, len(recs))

    def tick(self, reason=None):
        nf, n, torch = self.nf, self.n, self.torch
        reason = nf.REASON_TIMEOUT if reason is None else reason
        counts = []
        for r in range(n):
            rc, c, total = self.tabs[r].partials_export_device(n, r, self.exp[r].data_ptr(), self.exp[r].numel() * 8 // self.pb)
            assert rc == nf.OK and total == sum(c) and c[r] == 0
            counts.append(c)
        for owner in range(n):
            for src in range(n):
                if src == owner or not counts[src][owner]:
                    continue
                off = sum(counts[src][:owner])
                self.tabs[owner].partials_merge_device(n, owner, self.exp[src].data_ptr() + off * self.pb, counts[src][owner])
        out = []
        for r in range(n):
            rc, need = self.tabs[r].evict_owned_device(n, r, 0, 0, reason)         # cap 0: the number of FLOWS comes back
            buf = torch.zeros(max(need, 1) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            if need:
                assert rc == nf.TRUNCATED
                rc, got = self.tabs[r].evict_owned_device(n, r, buf.data_ptr(), need, reason)
                assert (rc, got) == (nf.OK, need)
            ev = buf.cpu().numpy()[: need * 144].view(nf.FLOW_RECORD)
            assert np.all(nf.distributed.shard_ids(ev, n) == r)                    # rank r delivered exactly the flows it owns
            out.append(ev)
        self.keep.clear()
        return nf.sort_by_key(np.concatenate(out)), counts


@pytest.mark.parametrize("n_ranks,hot,style", [(1, 0, 2), (2, 0, 1), (2, 900, 2), (4, 900, 1), (8, 0, 2), (8, 900, 1), (8, 999, 2), (3, 500, 0), (5, 0, 3)])
def test_ranks_with_contiguous_slices_equal_one_dedup_table(nf, O, n_ranks, hot, style):
    """bench.py --dedup's layout: rank r holds arrival positions [r n, (r+1) n) of the one stream. With hot = 900 one flow takes 90 %
    of the records and alternates over the interfaces (BASELINE configs[4]): every rank folds a share of it, and which interface is
    COUNTED is decided by the rank that saw the flow's earliest record."""
    th = O.zipf_thresholds(50_000, 1.1)
    recs = dedup_stream(O, 480_000, seed=21 + style, n_keys=50_000, thresholds=th, hot_permille=hot, style=style)
    per = len(recs) // n_ranks
    R = Ranks(nf, n_ranks)
    try:
        for epoch in range(2):                                             # every eviction restarts the sequence at 0
            for r in range(n_ranks):
                R.fold(r, recs[r * per:(r + 1) * per], r * per)
            got, counts = R.tick()
            assert_records_equal(got, _want(O, recs[: per * n_ranks]), "epoch %d" % epoch)
            if n_ranks > 1:
                assert sum(map(sum, counts)) > 0
    finally:
        R.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_ranks_with_interleaved_ragged_chunks(nf, O, seed):
    """Chunks of any size land on any rank (direct kernels for the small ones, streaming + partition passes for the large ones);
    a flow's records — and its interfaces — are spread over the ranks at random. Gaps in the numbering do not matter."""
    rng = np.random.default_rng(seed)
    n_ranks = int(rng.integers(2, 7))
    keys = int(rng.choice([300, 40_000]))
    recs = dedup_stream(O, 300_000, seed=50 + seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1),
                        hot_permille=int(rng.choice([0, 600, 900])), style=int(rng.choice([1, 2])))
    R = Ranks(nf, n_ranks)
    try:
        off, gap = 0, 0
        while off < len(recs):
            c = min(len(recs) - off, int(rng.choice([1, 63, 4_000, 30_000, 90_000])))
            R.fold(int(rng.integers(0, n_ranks)), recs[off:off + c], off + gap)
            off += c
            gap += int(rng.choice([0, 0, 5, 1000]))
        got, _ = R.tick(nf.REASON_CLOSING)
        assert_records_equal(got, _want(O, recs))
    finally:
        R.close()


def test_first_interface_is_decided_by_the_earliest_record_anywhere(nf, O):
    """The case that rules out merging flow-keyed kernel-dedup slots: rank 1 sees the flow's records on interface 3 BEFORE any on
    interface 2, rank 0 the other way round, and the flow's earliest record (interface 2) is on rank 0. Rank 1's bytes on interface
    2 must be counted, its bytes on interface 3 must not."""
    recs = dedup_stream(O, 4000, seed=9, n_keys=1, style=1)                      # ONE flow
    m = recs["metrics"]
    m["if_index_first_seen"][:2000] = np.where(np.arange(2000) < 1000, 2, 3)       # rank 0: interface 2 first, then 3
    m["if_index_first_seen"][2000:] = np.where(np.arange(2000) < 1000, 3, 2)       # rank 1: interface 3 first, then 2
    want = _want(O, recs)
    assert len(want) == 1 and want["metrics"]["if_index_first_seen"][0] == 2
    R = Ranks(nf, 2)
    try:
        R.fold(0, recs[:2000], 0)
        R.fold(1, recs[2000:], 2000)
        got, _ = R.tick()
        assert_records_equal(got, want)
        assert got["metrics"]["bytes"][0] == recs["metrics"]["bytes"][m["if_index_first_seen"] == 2].sum()
    finally:
        R.close()


def test_export_states_and_errors(nf, O):
    import torch
    recs = dedup_stream(O, 60_000, seed=4, n_keys=5_000, style=1)
    with nf.FlowTable(max_entries=1 << 16, table_log2_slots=18, mode=nf.MODE_KERNEL_DEDUP, local_fold=True) as tab:
        d = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).cuda()
        torch.cuda.synchronize()
        assert tab.ingest_device(d.data_ptr(), len(recs)) == (nf.OK, len(recs))
        n_sub = len(tab)
        buf = torch.zeros(n_sub * 32, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
        rc, counts, total = tab.partials_export_device(4, 0xFFFFFFFF, buf.data_ptr(), n_sub)      # NFAGG_SHARD_NONE: every sub-flow leaves
        assert rc == nf.OK and total == n_sub == sum(counts) and all(counts)
        raw = buf.cpu().numpy().view(np.uint8).reshape(-1, 256)
        # a partial = one (flow, interface): the sixth key word rides in the hot line's `end` word, owner = owner of the FLOW
        kx = raw[:, 56:64].copy().view(np.uint64).reshape(-1)
        assert np.all(kx >> 32 == 1) and set((kx & 0xFFFFFFFF).tolist()) <= {2, 3}
        keys = np.ascontiguousarray(raw[:, 8:48])
        pairs = {(k.tobytes(), int(i)) for k, i in zip(keys, kx & 0xFFFFFFFF)}
        assert len(pairs) == n_sub                                                               # one partial per pair
        bounds = np.cumsum([0] + counts)
        idx = np.arange(len(keys))[:: max(1, len(keys) // 400)]
        owner = np.array([nf.shard_of(keys[i].tobytes(), 4) for i in idx])
        assert np.array_equal(owner, np.searchsorted(bounds, idx, side="right") - 1)
        with pytest.raises(nf.NfaggError) as ei:
            tab.window_restart_device(4, 0, buf.data_ptr(), counts[0], 1 << 20)
        assert "does not move" in str(ei.value)
    # flow-keyed kernel-dedup slots do not merge across tables: a handle without local_fold refuses
    with nf.FlowTable(max_entries=1 << 12, mode=nf.MODE_KERNEL_DEDUP) as tab:
        assert tab.partial_bytes == 192
        with pytest.raises(nf.NfaggError) as ei:
            tab.partials_export_device(2, 0, 0, 0)
        assert "local_fold" in str(ei.value)


# ---------------------------------------------------------------- the in-process group: NFAGG_GROUP_LOCAL_FOLD in kernel-dedup mode
@pytest.mark.parametrize("n_members,hot,style", [(1, 0, 2), (2, 900, 1), (3, 900, 2), (8, 0, 1), (8, 900, 2)])
def test_group_local_fold_equals_one_dedup_table(nf, O, n_members, hot, style):
    th = O.zipf_thresholds(40_000, 1.1)
    recs = dedup_stream(O, 500_000, seed=77, n_keys=40_000, thresholds=th, hot_permille=hot, style=style)
    with nf.FlowGroup([0] * n_members, max_entries=1 << 20, mode=nf.MODE_KERNEL_DEDUP, local_fold=True, sketches=nf.SKETCH_CM | nf.SKETCH_HLL,
                      cm_log2_width=14, hll_p=10, staging_records=37_000) as grp:
        assert grp.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        want = _want(O, recs)
        assert len(grp) >= len(want)
        grp.merge_sketches()
        cs, cd, hs, hd = O.sketches(recs, 4, 14, 10)
        for m in grp.members:
            assert np.array_equal(m.sketch_snapshot(nf.CM_SRC), cs) and np.array_equal(m.sketch_snapshot(nf.HLL_DST), hd)
        got = nf.sort_by_key(grp.evict(nf.REASON_TIMEOUT))
        assert_records_equal(got, want)
        assert len(grp) == 0
        assert len(grp.evict(nf.REASON_TIMEOUT)) == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_group_local_fold_random_splits_and_epochs(nf, O, seed):
    import torch
    rng = np.random.default_rng(seed)
    n_members = int(rng.integers(2, 6))
    keys = int(rng.choice([500, 60_000]))
    recs = dedup_stream(O, 600_000, seed=100 + seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=int(rng.choice([0, 900])), style=2)
    with nf.FlowGroup([0] * n_members, max_entries=1 << 20, mode=nf.MODE_KERNEL_DEDUP, local_fold=True) as grp:
        off = 0
        for epoch in range(3):
            end = len(recs) if epoch == 2 else off + int(rng.integers(1, 250_000))
            start = off
            keep = []
            while off < end:
                c = min(end - off, int(rng.choice([1, 77, 5_000, 20_000, 120_000])))
                d = torch.from_numpy(recs[off:off + c].view(np.uint8).reshape(-1).copy()).cuda()
                torch.cuda.synchronize()
                keep.append(d)
                rc, took = grp.ingest_device(int(rng.integers(0, n_members)), d.data_ptr(), c)
                assert (rc, took) == (nf.OK, c)
                off += c
            got = nf.sort_by_key(grp.evict(nf.REASON_TIMEOUT))
            assert_records_equal(got, _want(O, recs[start:end]), "seed %d epoch %d" % (seed, epoch))


def test_group_local_fold_truncated_eviction_is_repeatable(nf, O):
    import ctypes as C
    import torch
    from netobserv_ebpf_agent_amd import _lib as L
    recs = dedup_stream(O, 200_000, seed=5, n_keys=20_000, hot_permille=500, style=1)
    n_members = 4
    with nf.FlowGroup([0] * n_members, max_entries=1 << 20, mode=nf.MODE_KERNEL_DEDUP, local_fold=True, staging_records=30_000) as grp:
        assert grp.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        want = _want(O, recs)
        bufs = [torch.zeros(144 * 16, dtype=torch.uint8, device="cuda") for _ in range(n_members)]
        torch.cuda.synchronize()
        p = (C.c_void_p * n_members)(*[b.data_ptr() for b in bufs])
        cap = (C.c_size_t * n_members)(*([16] * n_members))
        need = (C.c_size_t * n_members)()
        assert L.lib.nfagg_group_evict_device(grp._g, nf.REASON_TIMEOUT, p, cap, need) == nf.TRUNCATED
        assert grp.ingest(recs.view(nf.FLOW_RECORD)[:10]) == (nf.FULL, 0)
        need = [int(x) for x in need]
        assert sum(need) == len(want)                                            # flows, each at its owner
        bufs = [torch.zeros(144 * max(k, 1), dtype=torch.uint8, device="cuda") for k in need]
        torch.cuda.synchronize()
        counts = grp.evict_device([b.data_ptr() for b in bufs], need)
        assert counts == need
        got = np.concatenate([b.cpu().numpy()[: 144 * k].view(nf.FLOW_RECORD) for b, k in zip(bufs, counts)])
        assert np.array_equal(nf.distributed.shard_ids(got, n_members), np.repeat(np.arange(n_members), counts))
        assert_records_equal(nf.sort_by_key(got), want)
        assert grp.ingest(recs.view(nf.FLOW_RECORD)[:10]) == (nf.OK, 10)


# ---------------------------------------------------------------- bench.py --gpus N --dedup: configs[4] as it is written
def test_bench_gpus_2_dedup_runs_the_common_stream_rehearsed_on_one_gpu(nf, O):
    """`python bench.py --gpus 2 --dedup --hot-permille 900` as a PLAIN process (it spawns its ranks), both ranks on cuda:0 over
    gloo: ONE common stream — no pre-sharded escape any more — whose hot flow alternates over two interfaces and is folded by
    both ranks; sub-flow partials travel to the owners of their flows, the owners join and evict. The number of evicted flows
    must be the number of distinct keys of the common stream (one kernel-dedup table over it: the oracle)."""
    from test_partials_gpu import _bench
    from netobserv_ebpf_agent_amd import synth
    n, keys = 500_000, 30_000
    j = _bench("--gpus", "2", "--dedup", "--hot-permille", "900", "--no-sketches", "--same-device", "--backend", "gloo", "--records", str(n),
               "--flows", str(keys), "--steps", "2", "--warmup", "1")
    assert j["n_gpus"] == 2 and j["value"] > 0
    c = j["config"]
    assert "configs[4]" in c["workload"] and "local fold" in c["parallelism"] and "REHEARSAL" in c["parallelism"]
    assert c["mode"] == "kernel_dedup" and c["stream_variant"] == 2 and c["hot_permille"] == 900
    assert c["member_records_folded"] == [3 * n, 3 * n]                      # warm-up + 2 timed steps, both ranks, nothing skipped or routed
    ex = c["exchange"]
    assert ex["partial_bytes"] == 256 and ex["partials_sent"] > 0 and ex["partials_received"] > 0 and ex["all_to_all_ms"] > 0
    th = synth.zipf_thresholds(2 * keys, 1.1)
    whole = synth.stream_host(2 * n, seed=2, n_keys=2 * keys, thresholds=th, hot_permille=900, variant=2)
    want = O.run_accounter(whole, 1 << 22, mode=1)[0][1]
    assert c["evicted_flows_per_step"] == len(want)
    # the hot flow is seen on two interfaces by both ranks: it really is one flow with an observed interface in the oracle's eviction
    hot = want[np.argmax(want["metrics"]["packets"])]
    assert hot["metrics"]["nb_observed_intf"] >= 1


# ---------------------------------------------------------------- the limits of a sub-flow table
@pytest.mark.parametrize("style", [1, 2])
def test_subflow_table_large_batches_split_exactly_on_full(nf, O, style):
    """Batches large enough for the optimistic fold (folded whole, rolled back and split when the table filled): NFAGG_FULL comes
    at the record whose NEW (flow, interface) pair finds max_entries of them; every eviction equals one kernel-dedup table over
    exactly the records consumed since the last one, nothing is folded twice or lost across the rollbacks."""
    th = O.zipf_thresholds(60_000, 1.1)
    recs = dedup_stream(O, 1_000_000, seed=26 + style, n_keys=60_000, thresholds=th, style=style)
    view = recs.view(nf.FLOW_RECORD)
    with nf.FlowTable(max_entries=40_000, mode=nf.MODE_KERNEL_DEDUP, local_fold=True) as tab:
        off, start, fulls = 0, 0, 0
        while off < len(recs):
            rc, c = tab.ingest(view[off:])
            off += c
            if rc == nf.FULL:
                assert len(tab) == 40_000
                got = nf.sort_by_key(tab.evict(nf.REASON_FULL))
                assert_records_equal(got, _want(O, recs[start:off]), "eviction %d" % fulls)
                start, fulls = off, fulls + 1
        assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)), _want(O, recs[start:]))
        st = tab.stats()
        assert fulls >= 2 and st.optimistic_rollbacks >= 1 and st.records_ingested == len(recs)


def test_subflow_tables_end_the_epoch_when_the_sequence_window_is_used_up(nf, O):
    """The 32-bit window of a sub-flow table does not move (the order BETWEEN the slots of one flow matters until the join): the
    batch that would cross it gets NFAGG_FULL with nothing consumed, the caller evicts — exactly the records folded so far — and
    resubmits. One handle and a local-fold group."""
    th = O.zipf_thresholds(2000, 1.1)
    recs = dedup_stream(O, 60_000, seed=71, n_keys=2000, thresholds=th, style=1)
    view = recs.view(nf.FLOW_RECORD)
    with nf.FlowTable(max_entries=1 << 16, mode=nf.MODE_KERNEL_DEDUP, local_fold=True) as tab:
        assert tab.ingest(view[:20_000]) == (nf.OK, 20_000)
        tab.debug_skip_sequence(0xFFFFFFF0 - 20_000 - 5_000)                 # 5000 sequence numbers left in the window
        assert tab.ingest(view[20_000:]) == (nf.FULL, 0)
        assert tab.stats().sequence_rebases == 0
        assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_FULL)), _want(O, recs[:20_000]))
        assert tab.ingest(view[20_000:]) == (nf.OK, 40_000)                  # a new epoch: a fresh window
        assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)), _want(O, recs[20_000:]))
        with pytest.raises(nf.NfaggError):
            tab.window_restart_device(2, 0, 0, 0, 1 << 20)
    with nf.FlowGroup([0, 0, 0], max_entries=1 << 16, mode=nf.MODE_KERNEL_DEDUP, local_fold=True, staging_records=7_000) as grp:
        assert grp.ingest(view[:20_000]) == (nf.OK, 20_000)
        grp.debug_skip_sequence(0xFFFFFFF0 - 20_000 - 5_000)
        rc, c = grp.ingest(view[20_000:])
        assert rc == nf.FULL and c < 7_000                                  # the chunk that would cross the window is refused
        assert_records_equal(nf.sort_by_key(grp.evict(nf.REASON_FULL)), _want(O, recs[:20_000 + c]))
        assert grp.ingest(view[20_000 + c:]) == (nf.OK, 40_000 - c)
        assert_records_equal(nf.sort_by_key(grp.evict(nf.REASON_CLOSING)), _want(O, recs[20_000 + c:]))
