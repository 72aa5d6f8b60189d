"""nfagg_account[_device] on calls of more than a few epochs: the evict-on-full loop of Accounter.Account
(pkg/flow/account.go:81-96) with its epochs FOUND FIRST (previous-occurrence links, tests/test_epoch_boundaries.py) and every
complete epoch folded on its own from the sorted call — csrc/nfagg_epoch_par.hip, csrc/nfagg_account_par.inc; the default path of
such calls. `variant` 30 forces the kernel chain (its fallback) over the same streams. Same contract either way: every eviction
bit-identical, in order, to the oracle's."""
import numpy as np
import pytest

from conftest import assert_records_equal
from test_account_gpu import _check, _stream

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _fmix(x):
    x ^= x >> 33; x = (x * 0xff51afd7ed558ccd) & M64
    x ^= x >> 33; x = (x * 0xc4ceb9fe1a85ec53) & M64
    return x ^ (x >> 33)


def _key_hash(words):
    """csrc/nfagg_hash.h key_hash over the five little-endian words of a key, byte 39 zero."""
    h = 0x6E66616767206B31
    for w in words:
        h = ((_rotl(h, 27) ^ w) * 0x9E3779B97F4A7C15) & M64
    return _fmix(h)


def _colliding_keys(rng, k):
    """k distinct 40-byte keys with ONE 64-bit key hash: the inner state of the hash after two words can be steered to any value by
    the second word (the step is invertible), so keys that differ in their first two words and share the rest collide."""
    rest = [int(x) for x in rng.integers(0, 1 << 62, 3)]
    rest[2] &= (1 << 56) - 1                                             # byte 39 is not part of the key
    w0 = int(rng.integers(0, 1 << 62)); w1 = int(rng.integers(0, 1 << 62))
    seed = 0x6E66616767206B31
    s1 = ((_rotl(seed, 27) ^ w0) * 0x9E3779B97F4A7C15) & M64
    target = _rotl(s1, 27) ^ w1                                          # (rotl(state after word 0) ^ word 1) is what must be equal
    keys = [[w0, w1, *rest]]
    while len(keys) < k:
        v0 = int(rng.integers(0, 1 << 62))
        t1 = ((_rotl(seed, 27) ^ v0) * 0x9E3779B97F4A7C15) & M64
        keys.append([v0, _rotl(t1, 27) ^ target, *rest])
    hs = {_key_hash(w) for w in keys}
    assert len(hs) == 1 and len({tuple(w) for w in keys}) == k
    return np.array(keys, dtype=np.uint64)


@pytest.mark.parametrize("variant", [0, 30])
@pytest.mark.parametrize("max_entries,keys,n", [(5000, 100_000, 600_000), (100, 3_000, 150_000), (2, 50, 80_000), (20_000, 400_000, 900_000),
                                                 (5000, 4_000, 300_000), (1, 40, 70_000), (3, 200, 300_000), (5000, 1_000_000, 3_000_000)])
def test_epochs_found_first_equal_the_reference_loop(nf, O, max_entries, keys, n, variant):
    """Epochs of 2 records to epochs longer than a 16 Ki-record block of the cut walk; more cuts than one launch lists (65 535);
    a map that never fills; walks in one part (below 512 Ki records) and in two to four (the epochs a part completes are folded
    while the next part walks on)."""
    if variant == 30 and n > 200_000 and max_entries < 100:
        pytest.skip("the chain takes 30 us per epoch: covered at smaller sizes by test_account_gpu.py")
    recs = _stream(O, n, keys, seed=7 + max_entries)
    with nf.FlowTable(max_entries=max_entries, ingest_variant=variant) as tab:
        n_ev = _check(nf, O, tab, recs, max_entries, [n])
        if keys > max_entries:
            assert n_ev > 3
        st = tab.stats()
        assert st.records_ingested == n and st.evictions[nf.REASON_FULL] == n_ev - 1
        assert st.account_epochs_first >= 1 if variant == 0 else (st.account_epochs_first == 0 and st.account_chain >= 1)   # (a short rest of a call is the chain's)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_epochs_that_span_calls_a_hot_flow_and_the_sketches(nf, O, seed):
    """Calls of every size (the short ones take the chain: both paths work on one epoch in progress), a flow with 70 % of the
    records (segments of thousands of records: the wave-per-segment fold), the sketches fed along by whichever path folds."""
    rng = np.random.default_rng(seed)
    max_entries = int(rng.choice([7, 300, 5000]))
    recs = _stream(O, 700_000, int(rng.choice([2_000, 80_000])), seed=90 + seed, hot=int(rng.choice([0, 700])))
    batches = [int(rng.choice([1, 999, 90_000, 200_000, 300_000])) for _ in range(400)]
    with nf.FlowTable(max_entries=max_entries, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=12, hll_p=8) as tab:
        _check(nf, O, tab, recs, max_entries, batches)
        cs, cd, hs, hd = O.sketches(recs, 4, 12, 8)
        assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cs) and np.array_equal(tab.sketch_snapshot(nf.CM_DST), cd)
        assert np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs) and np.array_equal(tab.sketch_snapshot(nf.HLL_DST), hd)


@pytest.mark.parametrize("variant", [0, 30])
@pytest.mark.parametrize("n_colliding,max_entries,hot", [(2, 5000, False), (40, 300, False), (200, 5000, False), (2, 5000, True), (40, 300, True), (600, 300, True)])
def test_flows_that_share_their_key_hash(nf, O, n_colliding, max_entries, hot, variant):
    """Distinct keys with ONE 64-bit key hash (crafted: the hash is public and invertible step by step). The sort of the
    epochs-found-first path groups by hash bits only: previous-occurrence links and segment folds compare full keys and take such
    flows apart — on that path, no fallback (hot = False: the colliding flows are cold ones, their records a few positions apart among
    equal hash bits). hot = True: half of them are the stream's hottest flows; with 600 of them the run of equal hash bits holds
    70 000 records and 1 776 records have more than 4096 records of OTHER flows between themselves and their previous occurrence — more
    than the link search walks over: that call is the kernel chain's, and says so in the stats. Same evictions either way."""
    rng = np.random.default_rng(n_colliding)
    recs = _stream(O, 400_000, 20_000, seed=31 + n_colliding)
    ids = np.ascontiguousarray(recs["id"]).view(np.uint64).reshape(len(recs), 5).copy()
    _, inv, counts = np.unique(ids, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    order = np.argsort(-counts)
    if hot:
        chosen = np.concatenate([order[:n_colliding // 2], order[2000:2000 + n_colliding - n_colliding // 2]])
    else:
        chosen = order[3000:3000 + n_colliding]                        # a few records each
    crafted = _colliding_keys(rng, n_colliding)
    for k, f in enumerate(chosen):
        ids[inv == f] = crafted[k]
    recs.view(np.uint8).reshape(len(recs), 144)[:, :40] = ids.view(np.uint8).reshape(len(recs), 40)
    with nf.FlowTable(max_entries=max_entries, ingest_variant=variant) as tab:
        _check(nf, O, tab, recs, max_entries, [len(recs)])
        st = tab.stats()
        if variant == 30:
            assert st.account_epochs_first == 0 and st.account_chain >= 1
        elif hot:                                                         # (whether a walk really exceeds the bound depends on where the cold records fall)
            assert st.account_epochs_first + st.account_chain >= 1 and (st.account_declined == 0 or st.account_chain >= 1)
            if n_colliding >= 600:
                assert st.account_declined >= 1
        else:
            assert st.account_epochs_first >= 1 and st.account_declined == 0 and st.account_chain == 0


@pytest.mark.parametrize("max_entries,keys,n,batches", [
    (10_000, 1_000_000, 3_000_000, None),                  # scripts/agent.yml:35-36 deploys CACHE_MAX_FLOWS = 10 000
    (40_000, 1_000_000, 3_000_000, None),                  # the first size beyond the kernel chain's (32 768)
    (100_000, 1_000_000, 4_000_000, None),                 # pkg/flow/tracer_map_bench_test.go:64-111 brackets 1 k / 10 k / 100 k
    (100_000, 1_000_000, 4_000_000, [1_048_576] * 4),      # ... in calls of the shim's batch size: an epoch spans two calls
    (100_000, 300_000, 2_500_000, [131_072, 50_000, 400_000, 7, 900_000, 2_000_000]),   # calls on both sides of the path's entry bar
    (250_000, 1_000_000, 4_000_000, None),
    (100_000, 1_000_000, 6_000_000, [6_000_000, 1]),       # hot=700 below: segments of more than 4096 records (a workgroup per segment)
    (100_000, 60_000, 1_000_000, None),                    # a map that never fills
])
def test_table_sizes_the_reference_benchmarks_and_deploys(nf, O, max_entries, keys, n, batches):
    """CACHE_MAX_FLOWS beyond the kernel chain's 32 768: the epochs are found first for every call of at least 128 Ki records that
    may fill the map (round 5 sent these sizes through the optimistic fold of nfagg_ingest: fold, find the split, roll back, fold
    again). Epochs of ~600 k records: the ranking runs over tiles of the call, not one workgroup per epoch."""
    recs = _stream(O, n, keys, seed=11 + max_entries, hot=700 if n == 6_000_000 else 0)
    with nf.FlowTable(max_entries=max_entries) as tab:
        n_ev = _check(nf, O, tab, recs, max_entries, batches or [n])
        st = tab.stats()
        assert st.records_ingested == n and st.evictions[nf.REASON_FULL] == n_ev - 1
        if keys > 2 * max_entries:
            assert n_ev >= 2 and st.account_epochs_first >= 1 and st.account_chain == 0


def _collision_runs_stream(O, runs=12, per_run=2_000, max_entries=9_000):
    """lead (so that epochs END before the crafted part: only complete epochs are segment-folded) + `runs` runs of `per_run` crafted
    flows with one key hash per run, every flow twice: [first occurrences][second occurrences] — no record has more than
    2 x per_run < 4096 records of other flows between itself and its previous occurrence (the link search's bound) — + filler."""
    rng = np.random.default_rng(77)
    lead = _stream(O, 40_000, 40_000, seed=124)
    body = _stream(O, runs * per_run * 2, 50_000, seed=123)
    ids = np.ascontiguousarray(body["id"]).view(np.uint64).reshape(len(body), 5).copy()
    at = 0
    for r in range(runs):
        crafted = _colliding_keys(rng, per_run)
        for rep in range(2):
            ids[at:at + per_run] = crafted[rng.permutation(per_run)]
            at += per_run
    body.view(np.uint8).reshape(len(body), 144)[:, :40] = ids.view(np.uint8).reshape(len(body), 40)
    tail = _stream(O, 40_000, 40_000, seed=125)
    return np.concatenate([lead, body, tail]), max_entries


def test_many_runs_of_flows_that_share_their_key_hash(nf, O):
    """Several runs of crafted flows with one key hash each, one record per flow and epoch on average: every head with 16 positions
    of its run behind it is listed as a long segment — about G entries from G records per run, more than the list of long segments
    was sized for (call / 16) when the runs together exceed 1/16 of the call. The list is bounded now: what does not fit is folded
    by the listing lane itself (round-5 advisor finding: an out-of-bounds device write by construction)."""
    recs, max_entries = _collision_runs_stream(O)
    with nf.FlowTable(max_entries=max_entries) as tab:
        _check(nf, O, tab, recs, max_entries, [len(recs)])
        st = tab.stats()
        assert st.account_epochs_first >= 1 and st.account_declined == 0


def test_device_resident_call_and_small_output_room(nf, O):
    import torch
    max_entries = 2000
    recs = _stream(O, 500_000, 50_000, seed=5)
    want = O.run_accounter(recs, max_entries)
    d = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).cuda()
    with nf.FlowTable(max_entries=max_entries) as tab:
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
    with nf.FlowTable(max_entries=max_entries) as tab:
        while off < len(recs):
            rc, c, epochs = tab.account(view[off:], out_cap=5 * max_entries + 10, max_epochs=64)
            assert len(epochs) <= 5 and (rc == nf.TRUNCATED or off + c == len(recs))
            got += [nf.sort_by_key(e) for e in epochs]
            off += c
        got.append(nf.sort_by_key(tab.evict(nf.REASON_CLOSING)))
    assert len(got) == len(want)
    for g, (_, w) in zip(got, want):
        assert_records_equal(g, w)
