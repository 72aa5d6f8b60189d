"""GPU parity for the sketches (own spec; scalar oracle) and the per-CPU rollups
(pkg/tracer/tracer.go:1057-1146 + pkg/model/flow_content.go) through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ingest_variant", [0, 1, 3, 7, 10])
def test_count_min_and_hll_match_scalar_oracle(nf, O, ingest_variant):
    th = O.zipf_thresholds(20000, 1.1)
    recs = O.gen_stream(200000, seed=3, n_keys=20000, thresholds=th, variant=1)
    depth, log2w, p = 4, 16, 14
    with nf.FlowTable(max_entries=1 << 16, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_depth=depth, cm_log2_width=log2w, hll_p=p,
                      ingest_variant=ingest_variant) as tab:
        for off in range(0, len(recs), 50000):
            tab.ingest(recs[off:off + 50000].view(nf.FLOW_RECORD))
        cm_s, cm_d, hs, hd = O.sketches(recs, depth, log2w, p)
        assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cm_s)          # integer counters: bit-exact
        assert np.array_equal(tab.sketch_snapshot(nf.CM_DST), cm_d)
        assert np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs)
        assert np.array_equal(tab.sketch_snapshot(nf.HLL_DST), hd)
        for which, regs in ((nf.HLL_SRC, hs), (nf.HLL_DST, hd)):
            got, want = tab.hll_estimate(which), O.hll_estimate(regs, p)
            assert abs(got - want) <= np.spacing(want)                       # ±1 ULP (north_star)
            true = len(np.unique(recs["id"]["src_ip" if which == nf.HLL_SRC else "dst_ip"], axis=0))
            assert abs(got - true) / true < 0.05
        ip = recs[0]["id"]["src_ip"].tobytes()
        assert tab.cm_query(nf.CM_SRC, ip) == O.lib().orc_cm_query(cm_s.ctypes.data, depth, log2w, ip)
        tab.sketch_reset()
        assert not tab.sketch_snapshot(nf.CM_SRC).any() and tab.hll_estimate(nf.HLL_SRC) == 0.0


def test_sketches_follow_the_consumed_prefix_only(nf, O):
    """When ingest stops at a 'full' split, only the consumed records reach the sketches."""
    recs = O.gen_stream(3000, seed=4, n_keys=500, variant=1)
    for v in (0, 1):
      with nf.FlowTable(max_entries=100, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=12, hll_p=10, ingest_variant=v) as tab:
        rc, c = tab.ingest(recs.view(nf.FLOW_RECORD))
        assert rc == nf.FULL and 0 < c < len(recs)
        cm_s, _, hs, _ = O.sketches(recs[:c], 4, 12, 10)
        assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cm_s)
        assert np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs)


def _random_partials(O, kind, n_flows, n_cpu, rng):
    dt = O.KIND_DTYPES[O.KIND_INDEX[kind]]
    raw = rng.integers(0, 256, size=(n_flows * n_cpu, dt.itemsize), dtype=np.uint8)
    p = raw.view(dt).reshape(-1).copy()
    # realistic sparsity: most per-CPU slots are all-zero; small value ranges so that ties/maxima/dup metadata occur
    zero = rng.random(n_flows * n_cpu) < 0.6
    p.view(np.uint8).reshape(len(p), -1)[zero] = 0
    p["start"] = np.where(rng.random(len(p)) < 0.3, 0, rng.integers(1, 1000, len(p))).astype(np.uint64) * (~zero)
    p["end"] = rng.integers(0, 1000, len(p)).astype(np.uint64) * (~zero)
    p["eth_protocol"] = rng.choice([0, 0x0800, 0x86DD], len(p)) * (~zero)
    if kind == "additional":
        p["ipsec_ret"] = rng.integers(-2, 3, len(p)); p["ipsec_encrypted"] = rng.integers(0, 2, len(p)); p["flow_rtt"] = rng.integers(0, 50, len(p))
    if kind == "drops":
        p["bytes"] = rng.choice([0, 5, 40000, 65535], len(p)); p["packets"] = rng.choice([0, 1, 65000], len(p))
    if kind == "network_events":
        p["network_events_idx"] = rng.integers(0, 4, len(p))
        p["network_events"] = rng.integers(0, 3, size=(len(p), 4, 8)) * (rng.random((len(p), 4, 8)) < 0.2)
        p["packets"] = rng.integers(0, 2, size=(len(p), 4)) * rng.choice([1, 65535], size=(len(p), 4))
        p["bytes"] = rng.choice([0, 9, 65535], size=(len(p), 4))
    if kind == "xlat":
        for f in ("saddr", "daddr"):
            m = rng.integers(0, 3, len(p))
            a = p[f]
            a[m == 0] = 0
            a[m == 1] = [0] * 10 + [0xff, 0xff, 0, 0, 0, 0]
    if kind == "quic":
        p["version"] = rng.integers(0, 3, len(p))
    return p


@pytest.mark.parametrize("kind", ["additional", "dns", "drops", "network_events", "xlat", "quic"])
@pytest.mark.parametrize("n_cpu", [1, 8, 80])
def test_rollup_matches_oracle(nf, O, kind, n_cpu):
    rng = np.random.default_rng(hash((kind, n_cpu)) % 2**32)
    n_flows = 3000
    parts = _random_partials(O, kind, n_flows, n_cpu, rng)
    base = np.zeros(n_flows, dtype=O.FLOW_METRICS)
    base["start"] = np.where(rng.random(n_flows) < 0.5, 0, rng.integers(1, 1000, n_flows))
    base["end"] = rng.integers(0, 1000, n_flows)
    base["eth_protocol"] = rng.choice([0, 0x0800], n_flows)
    base["packets"] = rng.integers(0, 10, n_flows)
    wb, wf = O.rollup(kind, parts, n_cpu, base)
    with nf.FlowTable(max_entries=16) as tab:
        gb, gf = tab.rollup(kind, parts.view(np.uint8).reshape(-1).view(nf.ROLLUP_KINDS[kind]), n_cpu, base.view(nf.FLOW_METRICS))
    assert gb.tobytes() == wb.tobytes(), "base metrics differ"
    assert gf.tobytes() == wf.tobytes(), "folded partials differ"


def test_rollup_reference_vectors(nf):
    """flow_content_test.go:184-246 (RTT max + IPsec precedence) and :11-53 (DNS) as 1-flow rollups."""
    with nf.FlowTable(max_entries=16) as tab:
        base = np.zeros(1, dtype=nf.FLOW_METRICS)
        base["start_mono_time_ts"], base["end_mono_time_ts"], base["packets"] = 10, 20, 3
        a = np.zeros(4, dtype=nf.ADDITIONAL)
        a["start_mono_time_ts"] = [25, 30, 30, 30]; a["end_mono_time_ts"] = [25, 30, 30, 30]
        a["flow_rtt"] = [200, 1000, 800, 800]; a["ipsec_encrypted"] = [1, 0, 0, 0]; a["ipsec_encrypted_ret"] = [0, 0, 5, 0]
        b, f = tab.rollup("additional", a, 4, base)
        assert (b["start_mono_time_ts"][0], b["end_mono_time_ts"][0], b["packets"][0]) == (10, 30, 3)
        assert (f["start_mono_time_ts"][0], f["end_mono_time_ts"][0], f["flow_rtt"][0], f["ipsec_encrypted_ret"][0], f["ipsec_encrypted"][0]) == (25, 25, 1000, 5, 0)
        d = np.zeros(2, dtype=nf.DNS)
        d["start_mono_time_ts"] = [25, 30]; d["end_mono_time_ts"] = [25, 30]; d["latency"] = [1000, 2000]; d["id"] = [1, 1]; d["flags"] = [0b11, 0b1001]
        b, f = tab.rollup("dns", d, 2, base)
        assert (b["start_mono_time_ts"][0], b["end_mono_time_ts"][0]) == (10, 30)
        assert (f["start_mono_time_ts"][0], f["end_mono_time_ts"][0], f["latency"][0], f["id"][0], f["flags"][0]) == (25, 25, 2000, 1, 0b1011)


@pytest.mark.parametrize("k,log2w", [(1, 16), (10, 16), (100, 12), (5000, 8), (10**6, 16)])
def test_heavy_hitters_match_scalar_oracle(nf, O, k, log2w):
    """nfagg_cm_topk: estimates and order (estimate desc, address bytes asc) bit-exact vs the oracle. log2w = 8 collides
    almost every address with others: long runs of equal estimates across the k-th position exercise the tie rule."""
    import torch
    th = O.zipf_thresholds(30000, 1.1)
    recs = O.gen_stream(300000, seed=21, n_keys=30000, thresholds=th, variant=1)
    with nf.FlowTable(max_entries=1 << 16, sketches=nf.SKETCH_CM, cm_depth=3, cm_log2_width=log2w) as tab:
        tab.ingest(recs.view(nf.FLOW_RECORD))
        ev = tab.evict(nf.REASON_TIMEOUT)
        cm_s, cm_d, _, _ = O.sketches(recs, 3, log2w, 14)
        for which, cm, side in ((nf.CM_SRC, cm_s, 0), (nf.CM_DST, cm_d, 1)):
            want = O.cm_topk(cm, 3, log2w, ev.view(O.FLOW_RECORD), side, k)
            got = tab.cm_topk(which, ev, k)
            assert len(got) == len(want) == min(k, len(np.unique(ev["id"]["dst_ip" if side else "src_ip"], axis=0)))
            assert got.tobytes() == want.tobytes()
        d_ev = torch.from_numpy(ev.view(np.uint8).reshape(-1).copy()).cuda()
        assert tab.cm_topk(nf.CM_SRC, None, k, device_ptr=d_ev.data_ptr(), n=len(ev)).tobytes() == O.cm_topk(cm_s, 3, log2w, ev.view(O.FLOW_RECORD), 0, k).tobytes()
        assert len(tab.cm_topk(nf.CM_SRC, ev[:0], k)) == 0 and len(tab.cm_topk(nf.CM_SRC, ev, 0)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("hot,n", [(900, 200_000), (999, 70_001), (500, 65), (0, 100_000)])
def test_wave_combined_sketch_updates_on_hot_endpoints(nf, O, hot, n):
    """k_sketch_update (the sketch kernel of the direct and the dedup paths) sums the byte counts of the lanes that carry the
    same address across the wave (DPP reduction) before touching the Count-Min counters: hot endpoints, ragged tails, waves
    with one, two or many distinct addresses — counters and registers still exactly the scalar oracle's."""
    th = O.zipf_thresholds(3000, 1.1)
    recs = O.gen_stream(n, seed=90 + hot, n_keys=3000, thresholds=th, hot_permille=hot, variant=1)
    cs, cd, hs, hd = O.sketches(recs, 4, 12, 10)
    for kw in (dict(ingest_variant=1), dict(mode=nf.MODE_KERNEL_DEDUP)):          # direct kernel + k_sketch_update; dedup mode + k_sketch_update
        with nf.FlowTable(max_entries=1 << 16, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=12, hll_p=10, **kw) as tab:
            assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, n)
            assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cs) and np.array_equal(tab.sketch_snapshot(nf.CM_DST), cd)
            assert np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs) and np.array_equal(tab.sketch_snapshot(nf.HLL_DST), hd)


def test_misaligned_caller_owned_sketch_buffers_are_refused(nf):
    """HyperLogLog registers are one byte each and are raised by a CAS on the 32-bit word that holds them (csrc/nfagg_device.h
    sketch_add_side): a caller-owned register buffer (nfagg_config.ext_sketch) must be 4-byte aligned, the Count-Min counters 8-byte."""
    import torch
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    base = buf.data_ptr()
    for ext in ([0, 0, base + 1, 0], [0, 0, 0, base + 2], [base + 4, 0, 0, 0]):
        with pytest.raises(nf.NfaggError) as ei:
            nf.FlowTable(max_entries=100, sketches=nf.SKETCH_CM | nf.SKETCH_HLL, cm_log2_width=4, hll_p=4, ext_sketch=ext)
        assert "aligned" in str(ei.value)
    with nf.FlowTable(max_entries=100, sketches=nf.SKETCH_HLL, hll_p=8, ext_sketch=[0, 0, base + 4, base + 1024]) as tab:
        assert tab.sketch_snapshot(nf.HLL_SRC).sum() == 0
