"""Merge of the drained eBPF maps (FlowFetcher.LookupAndDeleteMap, pkg/tracer/tracer.go:1022-1146).

CPU leg: the oracle's join (oracle/nfagg_oracle_maps.c) against an independent composition in Python — a dict
keyed by the id, the per-kind orc_rollup chained in Go's walk order. GPU leg: nfagg_map_merge[_device] through
the C ABI, bit-exact against the oracle; chained into nfagg_encode_pb_content_device.
The reference has no unit test for this function (it needs live eBPF maps): parity of the join is unpinned,
the Accumulate* it applies are pinned by tests/test_oracle_kat.py."""
import numpy as np
import pytest

WALK = ("dns", "drops", "network_events", "xlat", "additional", "quic")      # tracer.go:1057-1110
HAS = {"dns": "has_dns", "drops": "has_drops", "network_events": "has_netev", "xlat": "has_xlat",
       "additional": "has_additional", "quic": "has_quic"}
PART = {"dns": "dns", "drops": "drops", "network_events": "netev", "xlat": "xlat", "additional": "additional", "quic": "quic"}


def make_maps(O, seed, n_pop, n_main, n_feat, n_cpu, kinds=WALK):
    """Main map over a random subset of a key population; every feature map over another random subset
    (so some flows exist only in feature maps); all struct bytes random, padding included."""
    rng = np.random.default_rng(seed)
    pop = np.zeros(n_pop, dtype=O.FLOW_ID)
    for i in range(n_pop):
        O.lib().orc_bench_flow_id(i * 7 + seed, pop[i:i + 1].ctypes.data)
    mi = pop[rng.permutation(n_pop)[:n_main]].copy()
    mv = np.zeros(n_main, dtype=O.FLOW_METRICS)
    mv.view(np.uint8).reshape(n_main, 104)[:] = rng.integers(0, 256, (n_main, 104), dtype=np.uint8)
    mv["start"][::3] = 0
    mv["eth_protocol"][::2] = 0
    feats = {}
    for kind in kinds:
        dt = O.KIND_DTYPES[O.KIND_INDEX[kind]]
        n = int(rng.integers(0, n_feat + 1)) if n_feat else 0
        fi = pop[rng.permutation(n_pop)[:n]].copy()
        n = len(fi)
        fv = np.zeros((n, n_cpu), dtype=dt)
        fv.view(np.uint8).reshape(n, n_cpu * dt.itemsize)[:] = rng.integers(0, 256, (n, n_cpu * dt.itemsize), dtype=np.uint8)
        z = rng.integers(0, 3, (n, n_cpu)) == 0                      # idle CPUs: all-zero partials
        fv.view(np.uint8).reshape(n, n_cpu, dt.itemsize)[z] = 0
        feats[kind] = (fi, fv)
    return mi, mv, feats


def python_join(O, mi, mv, feats, n_cpu):
    """dict + orc_rollup per kind in walk order."""
    flows = {}
    for i in range(len(mi)):
        k = mi[i].tobytes()[:39]
        if k not in flows:
            c = np.zeros((), dtype=O.CONTENT)
            c["base"] = mv[i]
            flows[k] = c
    for kind in WALK:
        if kind not in feats:
            continue
        fi, fv = feats[kind]
        seen = set()
        for i in range(len(fi)):
            k = fi[i].tobytes()[:39]
            if k in seen:
                continue
            seen.add(k)
            c = flows.setdefault(k, np.zeros((), dtype=O.CONTENT))
            base, folded = O.rollup(kind, fv[i:i + 1], n_cpu, np.array([c["base"]], dtype=O.FLOW_METRICS))
            c["base"], c[PART[kind]], c[HAS[kind]] = base[0], folded[0], 1
    keys = sorted(flows)
    ids = np.frombuffer(b"".join(k + b"\0" for k in keys), dtype=O.FLOW_ID) if keys else np.zeros(0, dtype=O.FLOW_ID)
    return ids, np.array([flows[k] for k in keys], dtype=O.CONTENT) if keys else np.zeros(0, dtype=O.CONTENT)


@pytest.mark.parametrize("seed,n_pop,n_main,n_feat,n_cpu", [(1, 50, 30, 40, 4), (2, 300, 100, 300, 1), (3, 10, 0, 10, 3), (4, 40, 40, 0, 2)])
def test_oracle_join_matches_python_composition(O, seed, n_pop, n_main, n_feat, n_cpu):
    mi, mv, feats = make_maps(O, seed, n_pop, n_main, n_feat, n_cpu)
    ids, contents = O.map_merge(mi, mv, feats, n_cpu)
    want_ids, want = python_join(O, mi, mv, feats, n_cpu)
    assert ids.tobytes() == want_ids.tobytes()
    for name in ("has_dns", "has_drops", "has_netev", "has_xlat", "has_additional", "has_quic"):
        assert np.array_equal(contents[name] != 0, want[name] != 0), name
    for name in ("base", "dns", "drops", "netev", "xlat", "additional", "quic"):
        assert contents[name].tobytes() == want[name].tobytes(), name


def test_oracle_join_duplicates_and_blank_byte(O):
    mi, mv, feats = make_maps(O, 9, 20, 10, 12, 2, kinds=("dns", "additional"))
    fi, fv = feats["dns"]
    if len(fi) < 2:
        pytest.skip("degenerate draw")
    fi2 = np.concatenate([fi, fi[:1]]); fv2 = np.concatenate([fv, fv[1:2]])   # id listed twice: the second row is ignored
    ids_a, c_a = O.map_merge(mi, mv, feats, 2)
    ids_b, c_b = O.map_merge(mi, mv, {"dns": (fi2, fv2), "additional": feats["additional"]}, 2)
    assert ids_a.tobytes() == ids_b.tobytes() and c_a.tobytes() == c_b.tobytes()
    mi3 = mi.copy(); mi3["pad"] = 0xAB                                        # byte 39 is not part of the Go key
    ids_c, c_c = O.map_merge(mi3, mv, feats, 2)
    assert ids_c.tobytes() == ids_a.tobytes() and c_c.tobytes() == c_a.tobytes()


# ------------------------------------------------------------------ GPU
def _sorted_product(nf, O, recs, present, parts):
    order = np.argsort([r.tobytes()[:40] for r in recs["id"]], kind="stable") if len(recs) else np.zeros(0, dtype=np.int64)
    return recs[order], present[order], {k: v[order] for k, v in parts.items()}, order


def _assert_matches_oracle(nf, O, got, mi, mv, feats, n_cpu):
    recs, present, parts, n_dup = got
    ids, contents = O.map_merge(mi, mv, feats, n_cpu)
    assert len(recs) == len(ids)
    srecs, spresent, sparts, _ = _sorted_product(nf, O, recs, present, parts)
    assert srecs["id"].tobytes() == ids.tobytes()
    assert srecs["metrics"].tobytes() == contents["base"].tobytes()
    want_present = np.zeros(len(ids), dtype=np.uint8)
    for kind, bit in (("additional", nf.FEAT_ADDITIONAL), ("dns", nf.FEAT_DNS), ("drops", nf.FEAT_DROPS),
                      ("network_events", nf.FEAT_NETWORK_EVENTS), ("xlat", nf.FEAT_XLAT), ("quic", nf.FEAT_QUIC)):
        want_present |= (contents[HAS[kind]] != 0).astype(np.uint8) * np.uint8(bit)
        assert sparts[kind].tobytes() == contents[PART[kind]].tobytes(), kind
    assert np.array_equal(spresent, want_present)
    return ids, contents


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_pop,n_main,n_feat,n_cpu", [
    (1, 50, 30, 40, 4), (2, 300, 100, 300, 1), (3, 10, 0, 10, 3), (4, 40, 40, 0, 2), (5, 1, 1, 1, 1),
    (6, 5000, 3000, 4000, 16), (7, 100_000, 60_000, 50_000, 8), (8, 2000, 1024, 1024, 64)])
def test_map_merge_matches_oracle(nf, O, seed, n_pop, n_main, n_feat, n_cpu):
    mi, mv, feats = make_maps(O, seed, n_pop, n_main, n_feat, n_cpu)
    with nf.FlowTable(max_entries=64) as tab:
        got = tab.map_merge(mi.view(nf.FLOW_ID), mv.view(nf.FLOW_METRICS), {k: (a.view(nf.FLOW_ID), b.view(np.uint8).reshape(-1).view(nf.ROLLUP_KINDS[k]))
                                                                           for k, (a, b) in feats.items()}, n_cpu)
    assert got[3] == 0
    _assert_matches_oracle(nf, O, got, mi, mv, feats, n_cpu)
    # order of first appearance: main map rows first, in their order
    recs = got[0]
    assert recs["id"][:len(mi)].tobytes() == mi.tobytes()
    # then ids new to each feature map in walk order
    seen = {m.tobytes() for m in mi}
    tail = []
    for kind in WALK:
        for i in feats[kind][0]:
            if i.tobytes() not in seen:
                seen.add(i.tobytes()); tail.append(i.tobytes())
    assert recs["id"][len(mi):].tobytes() == b"".join(tail)


@pytest.mark.gpu
def test_map_merge_duplicates_blank_byte_empty_and_truncated(nf, O):
    mi, mv, feats = make_maps(O, 11, 200, 120, 150, 4)
    conv = lambda f: {k: (a.view(nf.FLOW_ID), b.view(np.uint8).reshape(-1).view(nf.ROLLUP_KINDS[k])) for k, (a, b) in f.items()}
    with nf.FlowTable(max_entries=64) as tab:
        base = tab.map_merge(mi.view(nf.FLOW_ID), mv.view(nf.FLOW_METRICS), conv(feats), 4)
        # a key listed twice in one map (and in the main map): first row wins, rows counted
        fi, fv = feats["drops"]
        f2 = dict(feats); f2["drops"] = (np.concatenate([fi, fi[:5]]), np.concatenate([fv, fv[5:10]]))
        mi2, mv2 = np.concatenate([mi, mi[:3]]), np.concatenate([mv, mv[3:6]])
        dup = tab.map_merge(mi2.view(nf.FLOW_ID), mv2.view(nf.FLOW_METRICS), conv(f2), 4)
        assert dup[3] == 8
        assert dup[0].tobytes() == base[0].tobytes() and dup[1].tobytes() == base[1].tobytes()
        for k in base[2]:
            assert dup[2][k].tobytes() == base[2][k].tobytes(), k
        _assert_matches_oracle(nf, O, dup, mi2, mv2, f2, 4)
        # byte 39 of the id is not part of the key and comes out as zero
        mi3 = mi.copy(); mi3["pad"] = 0x5A
        pad = tab.map_merge(mi3.view(nf.FLOW_ID), mv.view(nf.FLOW_METRICS), conv(feats), 4)
        assert pad[0].tobytes() == base[0].tobytes() and pad[1].tobytes() == base[1].tobytes()
        # nothing at all; features only
        empty = tab.map_merge(mi[:0].view(nf.FLOW_ID), mv[:0].view(nf.FLOW_METRICS), {}, 4)
        assert len(empty[0]) == 0
        only = tab.map_merge(mi[:0].view(nf.FLOW_ID), mv[:0].view(nf.FLOW_METRICS), conv(feats), 4)
        _assert_matches_oracle(nf, O, only, mi[:0], mv[:0], feats, 4)
        # too small an output: the wrapper retries with the count the library reported (uniform 4-tuple)
        small = tab.map_merge(mi.view(nf.FLOW_ID), mv.view(nf.FLOW_METRICS), conv(feats), 4, cap=3)
        assert len(small) == 4 and small[0].tobytes() == base[0].tobytes() and small[1].tobytes() == base[1].tobytes()


@pytest.mark.gpu
def test_map_merge_device_then_encode(nf, O):
    """Drained maps -> merged flows -> pbflow bytes without leaving HBM."""
    import torch
    from test_pb_gpu import NAMES, AGENT4, frames
    n_cpu = 8
    mi, mv, feats = make_maps(O, 21, 20_000, 12_000, 9_000, n_cpu)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    d_mi, d_mv = dev(mi), dev(mv)
    d_f = {k: (dev(a), dev(b), len(a)) for k, (a, b) in feats.items()}
    total = len(mi) + sum(len(a) for a, _ in feats.values())
    sizes = {"records": 144, "present": 1, "additional": 32, "dns": 64, "drops": 32, "network_events": 72, "xlat": 56, "quic": 24}
    d_out = {k: torch.zeros(total * s + 16, dtype=torch.uint8, device="cuda") for k, s in sizes.items()}
    with nf.FlowTable(max_entries=64) as tab:
        rc, n, n_dup = tab.map_merge_device((d_mi.data_ptr(), d_mv.data_ptr(), len(mi)),
                                            {k: (i.data_ptr(), v.data_ptr(), m) for k, (i, v, m) in d_f.items()}, n_cpu,
                                            {k: t.data_ptr() for k, t in d_out.items()}, total)
        assert rc == nf.OK and n_dup == 0
        recs = d_out["records"][: n * 144].cpu().numpy().view(nf.FLOW_RECORD)
        present = d_out["present"][:n].cpu().numpy()
        parts = {k: d_out[k][: n * sizes[k]].cpu().numpy().view(nf.ROLLUP_KINDS[k]) for k in ("additional", "dns", "drops", "network_events", "xlat", "quic")}
        ids, contents = _assert_matches_oracle(nf, O, (recs, present, parts, 0), mi, mv, feats, n_cpu)
        now, mono = 1_720_000_000_000_000_000, 5 * 10**11
        d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        d_len = torch.empty(n, dtype=torch.int32, device="cuda")
        ptrs = {k: d_out[k].data_ptr() for k in ("additional", "dns", "drops", "xlat", "quic")}
        rc, need = tab.encode_pb_device(d_out["records"].data_ptr(), n, now, mono, AGENT4, nf.intf_table(NAMES), 0, 0, d_off.data_ptr(), d_len.data_ptr(),
                                        d_present=d_out["present"].data_ptr(), d_parts=ptrs)
        assert rc == nf.TRUNCATED
        d_pb = torch.empty(need + 16, dtype=torch.uint8, device="cuda")
        rc, wrote = tab.encode_pb_device(d_out["records"].data_ptr(), n, now, mono, AGENT4, nf.intf_table(NAMES), d_pb.data_ptr(), need,
                                         d_off.data_ptr(), d_len.data_ptr(), d_present=d_out["present"].data_ptr(), d_parts=ptrs)
        assert rc == nf.OK and wrote == need
        got = frames(d_pb[:need].cpu().numpy(), d_off.cpu().numpy().astype(np.uint64), d_len.cpu().numpy().astype(np.uint32))
    # the oracle encodes the flows in the product's order
    _, _, _, order = _sorted_product(nf, O, recs, present, parts)
    want_sorted = O.pb_encode_contents(ids, contents, O.pb_options(now, mono, AGENT4, O.intf_table(NAMES)))
    for pos, j in enumerate(order):
        assert got[j] == want_sorted[pos]
