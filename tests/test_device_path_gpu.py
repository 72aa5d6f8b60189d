"""Device-resident path (nfagg_ingest_device / nfagg_evict_device) with streams
generated in HBM: parity against the oracle at sizes the oracle finishes in
seconds, and size-independent properties at BASELINE.json's full size."""
import numpy as np
import pytest

from conftest import assert_records_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def dev_stream(torch, synth, n, **kw):
    buf = torch.empty(n * 144, dtype=torch.uint8, device="cuda")
    th, pi = kw.pop("thresholds", None), kw.pop("pop_index", None)
    d_th = torch.from_numpy(th.view(np.int64)).cuda() if th is not None else None
    d_pi = torch.from_numpy(pi.view(np.int64)).cuda() if pi is not None else None
    torch.cuda.synchronize()
    synth.stream_device(buf.data_ptr(), n, d_thresholds=d_th.data_ptr() if d_th is not None else 0,
                        d_pop_index=d_pi.data_ptr() if d_pi is not None else 0, **kw)
    torch.cuda.synchronize()
    return buf


def test_device_generator_matches_host_generator(nf, O, torch):
    synth = nf.synth
    th = synth.zipf_thresholds(5000, 1.1)
    pop = synth.shard_population(5000, 4, 2)
    for kw in (dict(variant=0), dict(variant=1, hot_permille=900), dict(variant=1, pop_index=pop), dict(variant=2, hot_permille=900)):
        d = dev_stream(torch, synth, 20000, j0=123, seed=9, n_keys=5000, thresholds=th, **dict(kw))
        h = O.gen_stream(20000, j0=123, seed=9, n_keys=5000, thresholds=th, **kw)
        assert d.cpu().numpy().tobytes() == h.tobytes(), kw


@pytest.mark.parametrize("ingest_variant", [0, 1, 3, 4, 5, 7, 10, 17])
def test_device_ingest_parity_2m_records(nf, O, torch, ingest_variant):
    n, keys = 2_000_000, 100_000
    th = nf.synth.zipf_thresholds(keys, 1.1)
    d = dev_stream(torch, nf.synth, n, seed=2, n_keys=keys, thresholds=th, variant=1)
    want = O.run_accounter(O.gen_stream(n, seed=2, n_keys=keys, thresholds=th, variant=1), 1 << 20)[0][1]
    with nf.FlowTable(max_entries=1 << 20, ingest_variant=ingest_variant) as tab:
        for off in range(0, n, 700_001 * 1):               # uneven chunks; 144-byte records keep 16-byte alignment
            m = min(700_001, n - off)
            assert tab.ingest_device(d.data_ptr() + off * 144, m) == (nf.OK, m)
        out = torch.empty(len(want) * 144 + 16, dtype=torch.uint8, device="cuda")
        got_n = tab.evict_device(out.data_ptr(), len(want))
        assert got_n == len(want)
        got = out[: got_n * 144].cpu().numpy().view(nf.FLOW_RECORD)
    assert_records_equal(nf.sort_by_key(got), want)


@pytest.mark.parametrize("hot_permille,sketches", [(0, False), (300, True)])
def test_default_routing_exact_parity_8m_records(nf, O, torch, hot_permille, sketches):
    """One 8 M-record call: the default routing takes the two-pass fold (>= 3 Mi records) with its admission filter
    and slot-index partitions; exact against the oracle, every order-dependent field scrambled; then a second epoch
    of mid-size calls (single-pass kernel) on the same handle, and the sketches of both epochs."""
    n, keys = 8_000_000, 300_000
    th = nf.synth.zipf_thresholds(keys, 1.1)
    d = dev_stream(torch, nf.synth, n, seed=11, n_keys=keys, thresholds=th, variant=1, hot_permille=hot_permille)
    host = O.gen_stream(n, seed=11, n_keys=keys, thresholds=th, variant=1, hot_permille=hot_permille)
    want = O.run_accounter(host, 1 << 24)[0][1]
    sk = (nf.SKETCH_CM | nf.SKETCH_HLL) if sketches else 0
    with nf.FlowTable(max_entries=1 << 24, sketches=sk, cm_log2_width=16, hll_p=12) as tab:
        out = torch.empty(len(want) * 144 + 16, dtype=torch.uint8, device="cuda")
        for chunk in (n, 1_000_000):
            for off in range(0, n, chunk):
                m = min(chunk, n - off)
                assert tab.ingest_device(d.data_ptr() + off * 144, m) == (nf.OK, m)
            assert tab.evict_device(out.data_ptr(), len(want)) == len(want)
            got = out[: len(want) * 144].cpu().numpy().view(nf.FLOW_RECORD)
            assert_records_equal(nf.sort_by_key(got), want, f"chunk {chunk}")
        assert tab.stats().records_bypassed > 0            # the cold tail did go through the partition queues
        if sketches:
            cm_s, cm_d, hs, hd = O.sketches(host, 4, 16, 12)
            assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), 2 * cm_s) and np.array_equal(tab.sketch_snapshot(nf.CM_DST), 2 * cm_d)
            assert np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs) and np.array_equal(tab.sketch_snapshot(nf.HLL_DST), hd)


def _i64(torch, buf, n):
    return buf.view(torch.int64).view(n, 18)


def test_full_size_properties_100m_records(nf, torch):
    """BASELINE configs[1]: 100 M-record Zipf(1.1) stream over 1 M flows, one GPU. The
    oracle cannot run this in seconds; check what the fold must conserve:
    distinct keys, wrapping sums of bytes and packets, min start / max end, OR of
    flags, and that re-folding the evicted records is the identity."""
    n, keys = 100_000_000, 1_000_000
    th = nf.synth.zipf_thresholds(keys, 1.1)
    d = dev_stream(torch, nf.synth, n, seed=2, n_keys=keys, thresholds=th, variant=0)
    q = _i64(torch, d, n)
    sum_bytes = int(q[:, 7].sum().item())                                    # int64 sum wraps like uint64
    sum_packets = int((q[:, 8] & 0xffffffff).sum().item()) & 0xffffffff
    min_start, max_end = int(q[:, 5].min().item()), int(q[:, 6].max().item())
    distinct = int(torch.unique(d.view(torch.int32).view(n, 36)[:, 3]).numel())   # src_ip[12:16] identifies the member
    with nf.FlowTable(max_entries=1 << 21, profile=True) as tab:
        assert tab.ingest_device(d.data_ptr(), n) == (nf.OK, n)
        assert len(tab) == distinct
        out = torch.empty(distinct * 144, dtype=torch.uint8, device="cuda")
        assert tab.evict_device(out.data_ptr(), distinct) == distinct
        assert len(tab) == 0
        e = _i64(torch, out, distinct)
        assert int(e[:, 7].sum().item()) == sum_bytes
        assert int((e[:, 8] & 0xffffffff).sum().item()) & 0xffffffff == sum_packets
        assert int(e[:, 5].min().item()) == min_start and int(e[:, 6].max().item()) == max_end
        assert int(torch.unique(out.view(torch.int32).view(distinct, 36)[:, 3]).numel()) == distinct   # one record per key
        flags = (e[:, 8] >> 48) & 0xffff
        assert int(flags.min().item()) == 0x10 and int(flags.max().item()) == 0x10
        # idempotence: one record per key in, the same records out
        assert tab.ingest_device(out.data_ptr(), distinct) == (nf.OK, distinct)
        out2 = torch.empty_like(out)
        assert tab.evict_device(out2.data_ptr(), distinct) == distinct
        a = nf.sort_by_key(out.cpu().numpy().view(nf.FLOW_RECORD))
        b = nf.sort_by_key(out2.cpu().numpy().view(nf.FLOW_RECORD))
        assert_records_equal(b, a)


@pytest.mark.parametrize("sketches", [False, True])
def test_partitions_with_more_flows_than_cache_entries_take_retry_rounds(nf, O, torch, sketches):
    """6 M flows in one two-pass call: ~2 900 flows per partition against 1 024 cache entries per pass-2 workgroup. The misses
    are retried by sub-partition (three more hash bits) instead of being merged one by one; exact against the oracle, every
    order-dependent field scrambled; then a second call into the same epoch (flows already in the table)."""
    n, keys = 12_000_000, 6_000_000
    d = dev_stream(torch, nf.synth, n, seed=51, n_keys=keys, variant=1)          # uniform over the population: every flow ~2 records
    host = d.cpu().numpy().view(O.FLOW_RECORD)
    sk = (nf.SKETCH_CM | nf.SKETCH_HLL) if sketches else 0
    with nf.FlowTable(max_entries=1 << 23, sketches=sk, cm_log2_width=16, hll_p=12) as tab:
        assert tab.ingest_device(d.data_ptr(), n) == (nf.OK, n)
        st = tab.stats()
        assert st.records_bypassed > n // 2                                       # nearly everything spills (no hot head) ...
        half = n // 2
        assert tab.ingest_device(d.data_ptr(), half) == (nf.OK, half)             # ... and a second batch meets the flows again
        want = O.run_accounter(np.concatenate([host, host[:half]]), 1 << 23)[0][1]
        out = torch.empty(len(want) * 144 + 16, dtype=torch.uint8, device="cuda")
        assert tab.evict_device(out.data_ptr(), len(want)) == len(want)
        got = out[: len(want) * 144].cpu().numpy().view(nf.FLOW_RECORD)
        assert_records_equal(nf.sort_by_key(got), want)
        if sketches:
            cm_s, _, hs, _ = O.sketches(np.concatenate([host, host[:half]]), 4, 16, 12)
            assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cm_s) and np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs)
