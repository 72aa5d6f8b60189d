"""Local-fold mode of the multi-GPU group (NFAGG_GROUP_LOCAL_FOLD; csrc/nfagg_group.inc, csrc/nfagg_combine.hip) on ONE GPU:
several members on device 0 fold the chunks that arrive at them — no routing, a flow lives on several members — and the
eviction merges the members' raw slots into their owners. The result must be bit-identical to ONE sequential Accounter over
the same records (oracle), whatever the split of the stream over the members."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stream(O, n, keys, seed, hot=0):
    return O.gen_stream(n, seed=seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=hot, variant=1)


@pytest.mark.parametrize("n_members,hot", [(1, 0), (2, 0), (3, 900), (8, 0), (8, 999)])
def test_local_fold_equals_one_accounter(nf, O, n_members, hot):
    """Chunks go round robin over the members (staging_records = chunk size); one eviction = the oracle's single map.
    hot = 900/999 permille: one flow takes most records (BASELINE configs[4]) and is folded by every member."""
    recs = _stream(O, 500_000, 40_000, seed=77, hot=hot)
    with nf.FlowGroup([0] * n_members, max_entries=1 << 20, local_fold=True, sketches=nf.SKETCH_CM | nf.SKETCH_HLL,
                      cm_log2_width=14, hll_p=10, staging_records=37_000) as grp:
        assert grp.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        want = O.run_accounter(recs, 1 << 20)[0][1]
        assert len(grp) >= len(want)                           # upper bound: a flow counts once per member that saw it
        grp.merge_sketches()
        cs, cd, hs, hd = O.sketches(recs, 4, 14, 10)
        for m in grp.members:
            assert np.array_equal(m.sketch_snapshot(nf.CM_SRC), cs) and np.array_equal(m.sketch_snapshot(nf.HLL_DST), hd)
        got = nf.sort_by_key(grp.evict(nf.REASON_TIMEOUT))
        assert got.tobytes() == want.tobytes()
        assert len(grp) == 0
        assert len(grp.evict(nf.REASON_TIMEOUT)) == 0          # nothing folded since: the timeout arm does not evict


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_local_fold_random_splits_and_epochs(nf, O, seed):
    """Several epochs, ragged chunk sizes, members fed directly (ingest_device from the test's own buffers) in random
    order: every eviction = the oracle's Accounter over the records of that epoch."""
    import torch
    rng = np.random.default_rng(seed)
    n_members = int(rng.integers(2, 6))
    recs = _stream(O, 600_000, int(rng.choice([500, 60_000])), seed=100 + seed, hot=int(rng.choice([0, 500])))
    with nf.FlowGroup([0] * n_members, max_entries=1 << 20, local_fold=True) as grp:
        off = 0
        for epoch in range(3):
            end = len(recs) if epoch == 2 else off + int(rng.integers(1, 250_000))
            start = off
            keep = []                  # the folds are asynchronous: the buffers stay alive until the eviction synchronised
            while off < end:
                c = min(end - off, int(rng.choice([1, 77, 5_000, 20_000, 120_000])))
                d = torch.from_numpy(recs[off:off + c].view(np.uint8).reshape(-1).copy()).cuda()
                keep.append(d)
                rc, took = grp.ingest_device(int(rng.integers(0, n_members)), d.data_ptr(), c)
                assert (rc, took) == (nf.OK, c)
                off += c
            want = O.run_accounter(recs[start:end], 1 << 20)[0][1]
            got = nf.sort_by_key(grp.evict(nf.REASON_TIMEOUT))
            assert got.tobytes() == want.tobytes(), (seed, epoch)


def test_local_fold_full_stop_conserves_every_record(nf, O):
    """max_entries bounds every member's table: the group stops when one member is full, the eviction delivers what was
    folded up to there — equal to one Accounter (large enough not to fill) over exactly the consumed records — and the
    stream goes on."""
    recs = _stream(O, 300_000, 50_000, seed=9)
    view = recs.view(nf.FLOW_RECORD)
    with nf.FlowGroup([0] * 3, max_entries=6_000, local_fold=True, staging_records=25_000) as grp:
        off, epochs = 0, 0
        while off < len(recs):
            start = off
            rc = nf.OK
            while off < len(recs) and rc == nf.OK:
                rc, c = grp.ingest(view[off:])
                off += c
            got = nf.sort_by_key(grp.evict(nf.REASON_FULL if rc == nf.FULL else nf.REASON_CLOSING))
            want = O.run_accounter(recs[start:off], 1 << 22)[0][1]
            assert got.tobytes() == want.tobytes(), epochs
            assert max(len(m) for m in grp.members) == 0
            epochs += 1
        assert epochs > 3
        st = [m.stats() for m in grp.members]
        assert sum(s.records_ingested for s in st) == len(recs)


def test_local_fold_truncated_eviction_is_repeatable(nf, O):
    """nfagg_group_evict_device with buffers that are too small: NFAGG_TRUNCATED and the needed counts, nothing lost;
    ingest is refused until the eviction is repeated; the repeated eviction delivers the merged flows once."""
    import torch
    from netobserv_ebpf_agent_amd import _lib as L
    recs = _stream(O, 200_000, 20_000, seed=5)
    n_members = 4
    with nf.FlowGroup([0] * n_members, max_entries=1 << 20, local_fold=True, staging_records=30_000) as grp:
        assert grp.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        want = O.run_accounter(recs, 1 << 20)[0][1]
        bufs = [torch.zeros(144 * 16, dtype=torch.uint8, device="cuda") for _ in range(n_members)]
        torch.cuda.synchronize()                               # torch fills them on ITS stream, the library writes on its own
        with pytest.raises(nf.NfaggError) as ei:
            grp.evict_device([b.data_ptr() for b in bufs], [16] * n_members)
        assert ei.value.code == nf.TRUNCATED
        assert grp.ingest(recs.view(nf.FLOW_RECORD)[:10]) == (nf.FULL, 0)
        p = (C.c_void_p * n_members)(*[b.data_ptr() for b in bufs])
        cap = (C.c_size_t * n_members)(*([16] * n_members))
        need = (C.c_size_t * n_members)()
        assert L.lib.nfagg_group_evict_device(grp._g, nf.REASON_TIMEOUT, p, cap, need) == nf.TRUNCATED
        need = [int(x) for x in need]
        assert sum(need) == len(want)
        bufs = [torch.zeros(144 * max(k, 1), dtype=torch.uint8, device="cuda") for k in need]
        torch.cuda.synchronize()
        counts = grp.evict_device([b.data_ptr() for b in bufs], need)
        assert counts == need
        got = np.concatenate([b.cpu().numpy()[: 144 * k].view(nf.FLOW_RECORD) for b, k in zip(bufs, counts)])
        # member j delivered exactly the flows it owns
        shard = nf.distributed.shard_ids(got, n_members)
        assert np.array_equal(shard, np.repeat(np.arange(n_members), counts))
        assert nf.sort_by_key(got).tobytes() == want.tobytes()
        assert grp.ingest(recs.view(nf.FLOW_RECORD)[:10]) == (nf.OK, 10)


def test_local_fold_accepts_kernel_dedup_mode(nf):
    """Round 4: the members of a kernel-dedup local-fold group are sub-flow tables (tests/test_dedup_local_fold_gpu.py)."""
    with nf.FlowGroup([0, 0], mode=nf.MODE_KERNEL_DEDUP, local_fold=True) as grp:
        assert [m.partial_bytes for m in grp.members] == [256, 256]
    with nf.FlowGroup([0, 0], mode=nf.MODE_KERNEL_DEDUP) as grp:       # routed: flow-keyed members
        assert [m.partial_bytes for m in grp.members] == [192, 192]
