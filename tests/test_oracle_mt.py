"""CPU only. The multi-core CPU baselines of bench.py's cpu_baseline (oracle/nfagg_oracle_mt.c; test infrastructure, not reference
paths: pkg/flow.Accounter is one goroutine, pkg/flow/account.go:58) against the one-core oracle: the local fold (thread-local folds
over contiguous slices, then a key-sharded merge by the same AccumulateBase, pkg/model/flow_content.go:28-61) must deliver the
one-core Accounter's flows bit for bit — every order-dependent field included — whatever the number of threads."""
import numpy as np
import pytest


@pytest.mark.parametrize("threads", [1, 3, 8, 64])
@pytest.mark.parametrize("n,keys,hot", [(60_000, 5_000, 0), (40_000, 300, 900), (5_000, 5_000, 0), (7, 3, 0)])
def test_local_fold_then_sharded_merge_equals_one_accounter(O, threads, n, keys, hot):
    recs = O.gen_stream(n, seed=n + keys + threads, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=hot, variant=1)
    want = O.run_accounter(recs, 1 << 20)
    assert len(want) == 1
    folded, flows, t_fold, t_merge, share, got = O.local_fold_mt(recs, threads, 1 << 20, want_flows=True)
    assert folded == n and flows == len(want[0][1])
    assert got.tobytes() == want[0][1].tobytes()
    assert 0.0 < share <= 1.0


def test_local_fold_reports_a_table_that_would_fill(O):
    recs = O.gen_stream(20_000, seed=3, n_keys=4_000, thresholds=O.zipf_thresholds(4_000, 1.1), variant=1)
    assert O.local_fold_mt(recs, 4, 100)[0] == 0                     # no eviction on "full" here: it says so instead of folding on
    folded, flows, *_ = O.partition_fold_mt(recs, 4, 1 << 20)
    assert folded == len(recs) and flows == O.local_fold_mt(recs, 4, 1 << 20)[1]


@pytest.mark.parametrize("chunk,threads", [(7_000, 3), (25_000, 8), (1, 2)])
def test_chunks_prefolded_then_merged_in_arrival_order_equal_one_accounter(O, chunk, threads):
    """How tests/test_full_size_gpu.py feeds the oracle 1 B records: every chunk folded by orc_local_fold_mt into one record per
    flow, those records folded into ONE Accounter in chunk order. AccumulateBase (pkg/model/flow_content.go:28-61) applied to
    partials in arrival order is the fold of the records themselves: sums, OR, min / max, last non-zero, first non-zero, and the
    first record's remaining fields travel with the first chunk's partial."""
    n = 60_000 if chunk > 1 else 300
    recs = O.gen_stream(n, seed=17 + chunk, n_keys=4_000, thresholds=O.zipf_thresholds(4_000, 1.1), hot_permille=300, variant=1)
    want = O.run_accounter(recs, 1 << 20)[0][1]
    acc = O.Accounter(1 << 20, 0)
    for lo in range(0, n, chunk):
        part = recs[lo:lo + chunk]
        folded, n_fl, _, _, _, flows = O.local_fold_mt(part, threads, 1 << 20, want_flows=True)
        assert folded == len(part) and n_fl == len(flows)
        assert acc.ingest(flows) == len(flows)
    got = acc.evict()
    acc.close()
    assert got.tobytes() == want.tobytes()
