"""NFAGG_MODE_KERNEL_DEDUP parity: the HIP dedup merge (csrc/nfagg_dedup.hip, csrc/nfagg_dedup_cached.hip),
called through the C ABI, against the oracle's sequential restatement of
bpf/flows.c:76-143 (update_existing_flow + add_observed_intf) — bit-exact on all
144 bytes. The reference has no unit test for this merge; the
oracle follows the source text line by line and is itself pinned to the reference's
own C compiled in place (oracle/_ref, tests/test_oracle_ref.py)."""
import numpy as np
import pytest

from conftest import assert_records_equal, dedup_stream
from test_parity_gpu import drive_product

pytestmark = pytest.mark.gpu


def check_dedup(nf, O, records, max_entries, batch, **kw):
    want = O.run_accounter(records, max_entries, mode=1)
    with nf.FlowTable(max_entries=max_entries, mode=nf.MODE_KERNEL_DEDUP, **kw) as tab:
        got = drive_product(tab, records.view(nf.FLOW_RECORD), batch)
    assert [r for r, _ in got] == [r for r, _ in want], "eviction sequence differs"
    for k, ((_, g), (_, w)) in enumerate(zip(got, want)):
        assert_records_equal(g, w, f"dedup eviction #{k}")
    return want


@pytest.mark.parametrize("ingest_variant", [1, 10, 16])   # 1 = direct per-record passes, 10 = LDS-cached passes (0 picks by batch size), 16 = the same with the partition pass sorting its items first
@pytest.mark.parametrize("style", [0, 1, 2, 3])
@pytest.mark.parametrize("batch", [1 << 30, 1000, 257])
def test_dedup_stream_parity(nf, O, style, batch, ingest_variant):
    th = O.zipf_thresholds(300, 1.1)
    recs = dedup_stream(O, 40000, seed=200 + style, n_keys=300, thresholds=th, style=style)
    want = check_dedup(nf, O, recs, 1 << 16, batch, ingest_variant=ingest_variant)
    ev = want[0][1]
    assert len(ev) > 150
    if style in (0, 2):
        assert (ev["metrics"]["nb_observed_intf"] == 6).any() and (ev["metrics"]["observed_direction"] == 3).any()


def test_dedup_differs_from_accounter_mode(nf, O):
    """The two modes must not be the same function: bytes are only counted on the first interface."""
    recs = dedup_stream(O, 5000, seed=3, n_keys=20, style=1)
    a = O.run_accounter(recs, 1000, mode=0)[0][1]
    d = check_dedup(nf, O, recs, 1000, 1 << 30)[0][1]
    assert len(a) == len(d) and (a["metrics"]["bytes"] != d["metrics"]["bytes"]).any()
    assert (d["metrics"]["nb_observed_intf"] >= 1).all()


def test_dedup_single_record_batches(nf, O):
    recs = dedup_stream(O, 500, seed=5, n_keys=12, style=2)
    check_dedup(nf, O, recs, 1000, 1)


@pytest.mark.parametrize("ingest_variant", [1, 10])
def test_dedup_hot_key(nf, O, ingest_variant):
    """BASELINE configs[4]: 90 % of the records are one flow, dedup on, interfaces alternating."""
    th = O.zipf_thresholds(2000, 1.1)
    recs = dedup_stream(O, 60000, seed=6, n_keys=2000, thresholds=th, hot_permille=900, style=1)
    check_dedup(nf, O, recs, 1 << 16, 1 << 30, ingest_variant=ingest_variant)
    recs = dedup_stream(O, 60000, seed=7, n_keys=2000, thresholds=th, hot_permille=900, style=2)
    check_dedup(nf, O, recs, 1 << 16, 1 << 30, ingest_variant=ingest_variant)


def test_dedup_cached_passes_with_many_flows(nf, O):
    """More sub-flows than the workgroup caches hold: cached and direct records of one flow must merge exactly."""
    th = O.zipf_thresholds(30000, 1.1)
    recs = dedup_stream(O, 300000, seed=17, n_keys=30000, thresholds=th, style=2)
    check_dedup(nf, O, recs, 1 << 17, 100_000)             # 0: batches >= 65536 records take the cached passes
    check_dedup(nf, O, recs, 1 << 17, 1 << 30, ingest_variant=10)


@pytest.mark.parametrize("ingest_variant", [10, 12, 16])   # 12: the partition pass makes no retry rounds (misses merged item by item); 16: sorted first
def test_dedup_partitions_with_more_subflows_than_cache_entries(nf, O, ingest_variant):
    """~2.4 M sub-flows over 2048 partitions (1024 cache entries each): exported entries and spilled records of one flow meet in
    the partition pass, sub-flows that find no entry are claimed at once and folded in retry rounds — the first interface of a
    flow must still be the one of its earliest record in the whole batch. Two batches: the second meets claimed slots."""
    recs = dedup_stream(O, 3_000_000, seed=23, n_keys=300_000, style=2)
    check_dedup(nf, O, recs, 1 << 20, 2_000_000, ingest_variant=ingest_variant)


@pytest.mark.parametrize("ingest_variant", [0, 10, 16])
@pytest.mark.parametrize("max_entries,batch", [(2, 1 << 30), (50, 333), (1, 50), (250, 4096)])
def test_dedup_evict_on_full_inside_batches(nf, O, max_entries, batch, ingest_variant):
    th = O.zipf_thresholds(400, 1.1)
    recs = dedup_stream(O, 12000, seed=77, n_keys=400, thresholds=th, style=2)
    want = check_dedup(nf, O, recs, max_entries, batch, ingest_variant=ingest_variant)
    assert sum(1 for r, _ in want if r == "full") >= 2


def test_dedup_epochs_and_ragged_sizes(nf, O):
    for n in (1, 63, 65, 257, 1025):
        check_dedup(nf, O, dedup_stream(O, n, seed=n, n_keys=9, style=2), 1000, 1 << 30)
    a = dedup_stream(O, 4000, seed=11, n_keys=50, style=0)
    b = dedup_stream(O, 4000, seed=12, n_keys=50, style=2)
    with nf.FlowTable(max_entries=1000, mode=nf.MODE_KERNEL_DEDUP) as tab:
        for part in (a, b, a):
            assert tab.ingest(part.view(nf.FLOW_RECORD)) == (nf.OK, len(part))
            got = nf.sort_by_key(tab.evict(nf.REASON_TIMEOUT))
            assert_records_equal(got, O.run_accounter(part, 1000, mode=1)[0][1])
            assert len(tab) == 0


@pytest.mark.parametrize("ingest_variant", [0, 10])    # 20 000 records: 0 = the direct kernels, 10 = streaming + partition passes
@pytest.mark.parametrize("n_shards", [2, 8])
def test_dedup_sharded(nf, O, n_shards, ingest_variant):
    th = O.zipf_thresholds(500, 1.1)
    recs = dedup_stream(O, 20000, seed=8, n_keys=500, thresholds=th, style=2)
    parts = []
    for s in range(n_shards):
        with nf.FlowTable(max_entries=4096, mode=nf.MODE_KERNEL_DEDUP, n_shards=n_shards, shard_id=s, ingest_variant=ingest_variant) as tab:
            assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
            parts.append(tab.evict())
    assert_records_equal(nf.sort_by_key(np.concatenate(parts)), O.run_accounter(recs, 1 << 20, mode=1)[0][1])


def test_dedup_with_sketches(nf, O):
    """Small batch (the direct kernels): the sketches run as their own kernel and see every record."""
    recs = dedup_stream(O, 20000, seed=9, n_keys=300, style=1)
    with nf.FlowTable(max_entries=4096, mode=nf.MODE_KERNEL_DEDUP, sketches=nf.SKETCH_CM | nf.SKETCH_HLL) as tab:
        assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        cm = tab.sketch_snapshot(nf.CM_SRC)
        hll = tab.sketch_snapshot(nf.HLL_DST)
        got = nf.sort_by_key(tab.evict())
    cs, _, _, hd = O.sketches(recs)
    assert np.array_equal(cm, cs) and np.array_equal(hll, hd)
    assert_records_equal(got, O.run_accounter(recs, 4096, mode=1)[0][1])


@pytest.mark.parametrize("style,hot,keys,log2_slots,variant,batch,n_shards", [
    (1, 0, 60_000, 0, 0, 1 << 30, 1), (2, 900, 60_000, 0, 0, 1 << 30, 1), (0, 0, 3_000, 0, 10, 70_001, 1), (3, 0, 60_000, 21, 0, 200_000, 1),
    (2, 0, 600_000, 22, 0, 1 << 30, 1),      # more sub-flows than the partitions' caches hold: retry rounds, each flushing its own entries
    (2, 0, 600_000, 22, 16, 1 << 30, 1),     # ... sorted first
    (1, 500, 60_000, 0, 12, 1 << 30, 1),     # no retry rounds: what the first cache cannot take is folded item by item
    (2, 0, 60_000, 0, 0, 1 << 30, 4),        # a shard filter: the sketches see the shard's records only
])
def test_dedup_sketches_fused_into_the_partition_flushes(nf, O, style, hot, keys, log2_slots, variant, batch, n_shards):
    """The streaming + partition passes feed the sketches themselves — one contribution per cache entry at its flush (round 5: a
    second pass over the batch, k_sketch_update: 8.1 of the 14.3 ms of a configs[4] step). The sketches count EVERY record's bytes
    (oracle/nfagg_oracle.c orc_sketch_ingest has no mode), also the ones the dedup merge does not count
    (bpf/flows.c:104-125: a record on another interface than the flow's first only moves end and flags): a cache entry's byte sum
    is the plain sum of its records — the counted / side decision falls at the merge. Count-Min counters, HLL registers and the
    records themselves, bit for bit; in one call and in several; grouped and two-phase flushes (tables with deferred claims)."""
    n = 900_000
    th = O.zipf_thresholds(keys, 1.1)
    recs = dedup_stream(O, n, seed=40 + style, n_keys=keys, thresholds=th, hot_permille=hot, style=style)
    view = recs.view(nf.FLOW_RECORD)
    kw = dict(table_log2_slots=log2_slots) if log2_slots else {}
    got, cms, hls = [], [], []
    for shard in range(n_shards):
        with nf.FlowTable(max_entries=1 << 20, mode=nf.MODE_KERNEL_DEDUP, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=14, hll_p=10,
                          ingest_variant=variant, n_shards=n_shards, shard_id=shard, **kw) as tab:
            for lo in range(0, n, batch):
                assert tab.ingest(view[lo:lo + batch]) == (nf.OK, len(view[lo:lo + batch]))
            cms.append([tab.sketch_snapshot(nf.CM_SRC), tab.sketch_snapshot(nf.CM_DST)])
            hls.append([tab.sketch_snapshot(nf.HLL_SRC), tab.sketch_snapshot(nf.HLL_DST)])
            got.append(tab.evict())
    cs, cd, hs, hd = O.sketches(recs, 4, 14, 10)
    assert np.array_equal(sum(c[0] for c in cms), cs) and np.array_equal(sum(c[1] for c in cms), cd)
    assert np.array_equal(np.maximum.reduce([h[0] for h in hls]), hs) and np.array_equal(np.maximum.reduce([h[1] for h in hls]), hd)
    assert_records_equal(nf.sort_by_key(np.concatenate(got)), O.run_accounter(recs, 1 << 21, mode=1)[0][1])


@pytest.mark.parametrize("style", [0, 1, 2, 3])
@pytest.mark.parametrize("batch", [1 << 30, 9000])
def test_dedup_grouped_first_flush_on_tables_with_deferred_claims(nf, O, style, batch):
    """Tables of 2^21 slots or more let the partition pass claim slots per workgroup (TableView.defer_claims): its first flush then
    brings the sub-flow entries of a flow together in LDS and the leader writes a NEW flow's whole slot once, with plain stores
    (csrc/nfagg_dedup_cached.hip parts_flush_grouped) — the state dedup_claim + dedup_merge would have left. Every style, one call
    and several (later calls meet existing slots: the two-phase path), flows with more interfaces than a leader takes (style 0 / 2:
    up to 14 per flow) included."""
    th = O.zipf_thresholds(3000, 1.1)
    recs = dedup_stream(O, 120_000, seed=300 + style, n_keys=3000, thresholds=th, style=style)
    check_dedup(nf, O, recs, 1 << 18, batch, ingest_variant=10, table_log2_slots=21)
    check_dedup(nf, O, recs, 1 << 18, batch, ingest_variant=16, table_log2_slots=21)     # sorted first: several grouped flushes per partition


def test_dedup_grouped_first_flush_hot_flow_and_epochs(nf, O):
    th = O.zipf_thresholds(50_000, 1.1)
    with nf.FlowTable(max_entries=1 << 19, mode=nf.MODE_KERNEL_DEDUP, table_log2_slots=21) as tab:
        for epoch, (style, hot) in enumerate(((1, 900), (2, 0), (1, 0))):
            recs = dedup_stream(O, 400_000, seed=40 + epoch, n_keys=50_000, thresholds=th, hot_permille=hot, style=style)
            view = recs.view(nf.FLOW_RECORD)
            for lo in range(0, len(recs), 150_000):
                assert tab.ingest(view[lo:lo + 150_000]) == (nf.OK, min(150_000, len(recs) - lo))
            assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_TIMEOUT)), O.run_accounter(recs, 1 << 22, mode=1)[0][1], "epoch %d" % epoch)


def test_dedup_sorted_first_with_dense_partitions(nf, O):
    """~2.4 M sub-flows over 2048 partitions — more than a cache holds per partition, ~150 per sub-partition: the sorted rounds flush
    several times per partition (fill-aware segments, grouped flushes on a table with deferred claims), a second batch meets the
    slots of the first, and a sub-flow table (local_fold) takes the same route with one slot per sub-flow."""
    recs = dedup_stream(O, 3_000_000, seed=29, n_keys=300_000, style=2)
    check_dedup(nf, O, recs, 1 << 20, 2_000_000, ingest_variant=16, table_log2_slots=21)
    want = O.run_accounter(recs, 1 << 22, mode=1)[0][1]
    with nf.FlowTable(max_entries=1 << 22, mode=nf.MODE_KERNEL_DEDUP, local_fold=True, ingest_variant=16) as tab:
        view = recs.view(nf.FLOW_RECORD)
        assert tab.ingest(view[:2_000_000]) == (nf.OK, 2_000_000)
        assert tab.ingest(view[2_000_000:]) == (nf.OK, 1_000_000)
        assert_records_equal(nf.sort_by_key(tab.evict(nf.REASON_TIMEOUT)), want)


@pytest.mark.parametrize("ingest_variant", [0, 16])
@pytest.mark.parametrize("n_keys", [1, 2])
def test_dedup_one_launch_whose_records_all_belong_to_one_partition(nf, O, n_keys, ingest_variant):
    """1.5 M records of one or two flows in ONE launch (found by tests/tools/soak.py): the records the streaming pass does not
    cache (the TLS ones) all spill to one partition, its staging group of four is full for nearly every one of them, and they
    go to the overflow list one slot each (nfagg_spill.h overflow_push_one; four slots each overran the list: error 5)."""
    recs = O.gen_stream(1_500_000, seed=918135167, n_keys=n_keys, variant=1)
    want = O.run_accounter(recs, 1 << 12, mode=1)[0][1]
    with nf.FlowTable(max_entries=1 << 12, mode=nf.MODE_KERNEL_DEDUP, ingest_variant=ingest_variant, staging_records=1 << 21) as tab:
        assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        got = nf.sort_by_key(tab.evict(nf.REASON_CLOSING))
    assert_records_equal(got, want, "one-partition batch")
