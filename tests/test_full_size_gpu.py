"""Bit-exact parity AT BASELINE SIZE on the bench's own route: the 100 M-record streams of
configs[1], configs[2] and the configs[4] shape, folded by ONE nfagg_ingest_device call with bench.py's table
configuration, every evicted 144-byte record compared with the oracle's sequential fold of the same bytes (the device
stream is copied to the host for the oracle: 14.4 GB; the C oracle folds it in ~10 s)."""
import numpy as np
import pytest

import bench
from conftest import assert_records_equal
from test_device_path_gpu import dev_stream, torch  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

N, KEYS = 100_000_000, 1_000_000


def _fold_and_compare(nf, O, torch, d, host, mode, sketches, max_entries, N=N):
    want = O.run_accounter(host, max_entries, mode=mode)
    assert [r for r, _ in want] == ["closing"], "the bench configuration never evicts on full"
    want = want[0][1]
    sk = (nf.SKETCH_CM | nf.SKETCH_HLL) if sketches else 0
    with nf.FlowTable(max_entries=max_entries, sketches=sk, mode=mode, profile=True) as tab:
        assert tab.ingest_device(d.data_ptr(), N) == (nf.OK, N)
        st = tab.stats()
        assert st.ingest_launches == 1, "the bench's route is ONE fold call for the whole stream"
        out = torch.empty(len(want) * 144 + 16, dtype=torch.uint8, device="cuda")
        assert len(tab) == len(want)
        assert tab.evict_device(out.data_ptr(), len(want)) == len(want)
        got = nf.sort_by_key(out[: len(want) * 144].cpu().numpy().view(nf.FLOW_RECORD))
        assert_records_equal(got, want, "100 M-record fold vs oracle")
        if sketches:
            cm_s, cm_d, hs, hd = O.sketches(host)
            assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cm_s) and np.array_equal(tab.sketch_snapshot(nf.CM_DST), cm_d)
            assert np.array_equal(tab.sketch_snapshot(nf.HLL_SRC), hs) and np.array_equal(tab.sketch_snapshot(nf.HLL_DST), hd)
            for which, regs in ((nf.HLL_SRC, hs), (nf.HLL_DST, hd)):
                est, ref = tab.hll_estimate(which), O.hll_estimate(regs, 14)
                assert abs(est - ref) <= np.spacing(ref)             # north_star: +-1 ULP of a scalar HLL on the same registers
    return want


def test_configs1_bench_stream_bit_exact_100m(nf, O, torch):
    """configs[1] exactly as bench.py runs it: seed 2, stream variant 0, one call, bench's max_entries."""
    th = nf.synth.zipf_thresholds(KEYS, 1.1)
    d = dev_stream(torch, nf.synth, N, seed=2, n_keys=KEYS, thresholds=th, variant=0)
    host = d.cpu().numpy()
    want = _fold_and_compare(nf, O, torch, d, host, nf.MODE_ACCOUNTER, False, bench.DEFAULT_MAX_ENTRIES)
    assert 900_000 < len(want) <= KEYS


def test_configs2_scrambled_stream_with_sketches_bit_exact_100m(nf, O, torch):
    """configs[2]: the same shape with every order-dependent field varying per record (stream variant 1), CM + HLL on:
    records, Count-Min counters, HLL registers bit-exact, estimates within 1 ULP."""
    th = nf.synth.zipf_thresholds(KEYS, 1.1)
    d = dev_stream(torch, nf.synth, N, seed=3, n_keys=KEYS, thresholds=th, variant=1)
    host = d.cpu().numpy()
    _fold_and_compare(nf, O, torch, d, host, nf.MODE_ACCOUNTER, True, bench.DEFAULT_MAX_ENTRIES)


def test_configs4_hot_flow_dedup_bit_exact_100m(nf, O, torch):
    """configs[4] shape as `bench.py --dedup --hot-permille 900` runs it: 90 % of the records one flow, every flow seen on
    two interfaces (stream variant 2), kernel-dedup merge on."""
    th = nf.synth.zipf_thresholds(KEYS, 1.1)
    d = dev_stream(torch, nf.synth, N, seed=2, n_keys=KEYS, thresholds=th, variant=2, hot_permille=900)
    host = d.cpu().numpy()
    want = _fold_and_compare(nf, O, torch, d, host, nf.MODE_KERNEL_DEDUP, False, bench.DEFAULT_MAX_ENTRIES)
    assert (want["metrics"]["nb_observed_intf"] >= 1).any()


def test_uniform_singleton_heavy_stream_bit_exact_80m(nf, O, torch):
    """Scan-like traffic: 80 M records uniform over 8 M flows. Pass 1's cache holds next to nothing, every partition of pass 2
    gets ~39 k queue entries over ~3 900 flows — four times its cache: most records go through the retry list, sorted by the
    sub-partition bits their queue entries carry, and eight more rounds (csrc/nfagg_ingest_part.hip)."""
    n, keys = 80_000_000, 8_000_000
    d = dev_stream(torch, nf.synth, n, seed=11, n_keys=keys, variant=1)
    host = d.cpu().numpy()
    want = _fold_and_compare(nf, O, torch, d, host, nf.MODE_ACCOUNTER, False, 1 << 24, N=n)
    assert 7_900_000 < len(want) <= keys


def test_configs3_at_size_eight_ranks_rehearsed_on_one_gpu(nf, O, torch):
    """configs[3] at its own size in multi-handle form, rehearsed on ONE GPU: 8 unsharded handles stand for the ranks of
    `bench.py --gpus 8`; rank r folds the contiguous slice [12.5 M r, 12.5 M (r + 1)) of ONE 100 M-record Zipf(1.1) stream over
    10 M flows with job-global sequence numbers (one call per rank, CM + HLL on), then the tick: every rank's flows as 192-byte
    partials grouped by owner, device-to-device "exchange" (pointer arithmetic: the ranks share the device), merge, evict owned.
    The union must be bit-identical to ONE Accounter (pkg/flow/account.go:58-124; the oracle) over the whole stream, the merged
    Count-Min / HLL arrays equal to the oracle's over all records, the HLL estimates within 1 ULP."""
    n_ranks, n, keys = 8, 100_000_000, 10_000_000
    per = n // n_ranks
    th = nf.synth.zipf_thresholds(keys, 1.1)
    d = dev_stream(torch, nf.synth, n, seed=4, n_keys=keys, thresholds=th, variant=1)          # seed 4
    host = d.cpu().numpy()
    want = O.run_accounter(host, 1 << 25)[0][1]
    assert 4_500_000 < len(want) <= keys                                # 4.86 M of the 10 M flows appear in 100 M records
    sk = nf.SKETCH_CM | nf.SKETCH_HLL
    max_entries = 1 << 22                                               # a rank sees ~1.9 M of the flows; after the merge it holds what it owns too
    tabs = [nf.FlowTable(max_entries=max_entries, table_log2_slots=24, sketches=sk) for _ in range(n_ranks)]
    try:
        for r, tab in enumerate(tabs):
            tab.set_sequence(r * per)
            assert tab.ingest_device(d.data_ptr() + r * per * 144, per) == (nf.OK, per)
        seen = [len(t) for t in tabs]
        assert all(1_000_000 < s < max_entries for s in seen), seen
        # the per-tick collective, rehearsed: CM sum / HLL max over the ranks' arrays = the oracle's over all records
        cm_s, cm_d, hs, hd = O.sketches(host)
        for which, ref in ((nf.CM_SRC, cm_s), (nf.CM_DST, cm_d)):
            acc = np.zeros_like(ref)
            for t in tabs:
                acc += t.sketch_snapshot(which)
            assert np.array_equal(acc, ref)
        for which, ref in ((nf.HLL_SRC, hs), (nf.HLL_DST, hd)):
            acc = np.zeros_like(ref)
            for t in tabs:
                acc = np.maximum(acc, t.sketch_snapshot(which))
            assert np.array_equal(acc, ref)
            hist = np.bincount(acc, minlength=65).astype(np.uint32)
            est, want_est = nf.hll_estimate_from_histogram(hist, 14), O.hll_estimate(ref, 14)
            assert abs(est - want_est) <= np.spacing(want_est)
        # export by owner, exchange, merge, evict owned
        exp = [torch.empty(s * 24, dtype=torch.int64, device="cuda") for s in seen]
        torch.cuda.synchronize()
        counts = []
        for r, tab in enumerate(tabs):
            rc, c, total = tab.partials_export_device(n_ranks, r, exp[r].data_ptr(), seen[r])
            assert rc == nf.OK and total == sum(c) and c[r] == 0
            counts.append(c)
        for owner in range(n_ranks):
            for src in range(n_ranks):
                if src != owner and counts[src][owner]:
                    tabs[owner].partials_merge_device(n_ranks, owner, exp[src].data_ptr() + sum(counts[src][:owner]) * 192, counts[src][owner])
        got = []
        for r, tab in enumerate(tabs):
            rc, need = tab.evict_owned_device(n_ranks, r, 0, 0)
            assert rc == nf.TRUNCATED and need > 0
            out = torch.empty(need * 144 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            assert tab.evict_owned_device(n_ranks, r, out.data_ptr(), need) == (nf.OK, need)
            got.append(out[: need * 144].cpu().numpy().view(nf.FLOW_RECORD))
        assert sum(len(g) for g in got) == len(want)
        assert_records_equal(nf.sort_by_key(np.concatenate(got)), want, "configs[3] at size: union of 8 ranks vs ONE Accounter")
    finally:
        for t in tabs:
            t.close()


def test_configs4_at_size_eight_ranks_dedup_local_fold_rehearsed_on_one_gpu(nf, O, torch):
    """configs[4] as BASELINE.json writes it — "adversarial 90 % single-hot-flow stream, dedup on, 8 GPU" — at 100 M records,
    rehearsed on ONE GPU: 8 kernel-dedup handles created with local_fold (sub-flow tables) stand for the ranks of
    `bench.py --gpus 8 --dedup --hot-permille 900`; rank r folds the contiguous slice [12.5 M r, 12.5 M (r + 1)) of ONE stream whose
    hot flow alternates over two interfaces (stream variant 2) with job-global sequence numbers, CM + HLL on, then the tick: the
    sketch arrays summed / maxed over the ranks (bit-exact vs the oracle's over all records, estimates within 1 ULP), sub-flow partials
    (256 bytes) grouped by the owner of their FLOW, device-to-device exchange, merge, join, evict owned. Every rank folds ~11 M
    records of the hot flow; which interface is counted is decided by the rank that holds the stream's first record. The union
    must be bit-identical to ONE kernel-dedup table (bpf/flows.c:76-143; the oracle in mode 1, pinned to oracle/_ref)."""
    n_ranks, n, keys = 8, 100_000_000, 1_000_000
    per = n // n_ranks
    th = nf.synth.zipf_thresholds(keys, 1.1)
    d = dev_stream(torch, nf.synth, n, seed=5, n_keys=keys, thresholds=th, variant=2, hot_permille=900)    # seed 5
    host = d.cpu().numpy()
    want = O.run_accounter(host, 1 << 22, mode=1)[0][1]
    assert 400_000 < len(want) <= keys and (want["metrics"]["nb_observed_intf"] >= 1).any()
    # sketches ON, as `bench.py --gpus N --dedup` runs it (bench.py: sketches default to on at N > 1): fed by the partition pass's
    # flushes of the kernel-dedup fold (csrc/nfagg_dedup_cached.hip) — every record's bytes, counted by the merge or not
    sk = nf.SKETCH_CM | nf.SKETCH_HLL
    tabs = [nf.FlowTable(max_entries=1 << 21, mode=nf.MODE_KERNEL_DEDUP, local_fold=True, sketches=sk, profile=True) for _ in range(n_ranks)]
    try:
        for r, tab in enumerate(tabs):
            tab.set_sequence(r * per)
            assert tab.ingest_device(d.data_ptr() + r * per * 144, per) == (nf.OK, per)
        assert all(t.stats().sketch_launches == 0 for t in tabs), "the sketches are fused into the fold: no second pass over the batch"
        # the per-tick collective, rehearsed: CM sum / HLL max over the ranks' arrays = the oracle's over all records
        cm_s, cm_d, hs, hd = O.sketches(host)
        for which, ref in ((nf.CM_SRC, cm_s), (nf.CM_DST, cm_d)):
            acc = np.zeros_like(ref)
            for t in tabs:
                acc += t.sketch_snapshot(which)
            assert np.array_equal(acc, ref)
        for which, ref in ((nf.HLL_SRC, hs), (nf.HLL_DST, hd)):
            acc = np.zeros_like(ref)
            for t in tabs:
                acc = np.maximum(acc, t.sketch_snapshot(which))
            assert np.array_equal(acc, ref)
            hist = np.bincount(acc, minlength=65).astype(np.uint32)
            est, want_est = nf.hll_estimate_from_histogram(hist, 14), O.hll_estimate(ref, 14)
            assert abs(est - want_est) <= np.spacing(want_est)
        seen = [len(t) for t in tabs]                                     # (flow, interface) pairs per rank
        pb = tabs[0].partial_bytes
        assert pb == 256
        exp = [torch.empty(s * pb // 8, dtype=torch.int64, device="cuda") for s in seen]
        torch.cuda.synchronize()
        counts = []
        for r, tab in enumerate(tabs):
            rc, c, total = tab.partials_export_device(n_ranks, r, exp[r].data_ptr(), seen[r])
            assert rc == nf.OK and total == sum(c) and c[r] == 0
            counts.append(c)
        for owner in range(n_ranks):
            for src in range(n_ranks):
                if src != owner and counts[src][owner]:
                    tabs[owner].partials_merge_device(n_ranks, owner, exp[src].data_ptr() + sum(counts[src][:owner]) * pb, counts[src][owner])
        got = []
        for r, tab in enumerate(tabs):
            rc, need = tab.evict_owned_device(n_ranks, r, 0, 0)
            assert rc == nf.TRUNCATED and need > 0
            out = torch.empty(need * 144 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            assert tab.evict_owned_device(n_ranks, r, out.data_ptr(), need) == (nf.OK, need)
            got.append(out[: need * 144].cpu().numpy().view(nf.FLOW_RECORD))
        assert sum(len(g) for g in got) == len(want)
        assert_records_equal(nf.sort_by_key(np.concatenate(got)), want, "configs[4] at size: union of 8 dedup ranks vs ONE kernel-dedup table")
        hot = want[np.argmax(want["metrics"]["packets"])]
        assert hot["metrics"]["packets"] > 0.8 * n * 0.45                  # the hot flow: ~90 % of the records, counted on ONE of its two interfaces
    finally:
        for t in tabs:
            t.close()


def _key_hash64(torch, rec):
    """A 64-bit mix of the 40 key bytes of records viewed as int64 [m, 18] (byte 39 is Go's blank field: not part of the key).
    Only used to ORDER and to COUNT flows in torch; nothing to do with the library's hash."""
    h = rec[:, 0].clone()
    for c, mul in ((1, -7046029254386353131), (2, -4417276706812531889), (3, 1609587929392839161), (4, -8663945395140668459)):
        w = rec[:, c] if c < 4 else (rec[:, 4] & 0x00FFFFFFFFFFFFFF)
        h.mul_(mul).bitwise_xor_(w)
        h.bitwise_xor_(h >> 29)
    return h


def test_configs3_at_its_full_size_one_billion_records(nf, O, torch):
    """configs[3] at BASELINE.json's OWN size — 1 B records (144 GB, generated on the device: SURVEY §8(d) config 4, seed 4),
    10 M flows — on ONE GPU, compared with the ORACLE bit for bit: the device-generated records come down in 50 M-record chunks
    into ONE incremental oracle Accounter (pkg/flow/account.go:58-124 restated; every chunk pre-folded on the host's cores by
    orc_local_fold_mt, then merged in arrival order), and the union of the eight ranks' evictions must be that Accounter's closing
    eviction. Beside it, what does not depend on size:
      * two independent routes through the library deliver the SAME evictions, bit for bit: (A) one handle, four calls of
        540 M / 270 M / 140 M / 50 M records (one per form of the partition queues' entries); (B) eight unsharded handles standing for the ranks of `bench.py --gpus 8`, each folding its 125 M-record
        slice with job-global sequence numbers, then partials by owner, merge, evict owned (the route of the test above);
      * linearity against the stream itself: the evicted flows' bytes add up to the stream's (mod 2^64), their packets to the
        stream's (mod 2^32, the width of the field), their flags OR to the stream's, the latest end and the earliest non-zero
        start are the stream's;
      * the number of evicted flows is the number of distinct keys in the stream (counted in torch from a 64-bit mix of the keys)."""
    n_ranks, n, keys = 8, 1_000_000_000, 10_000_000
    torch.cuda.synchronize(); torch.cuda.empty_cache()                   # (what earlier tests' tensors left in torch's cache counts as used)
    free, _ = torch.cuda.mem_get_info()
    if free < (205 << 30):
        pytest.skip("needs ~195 GB of free HBM (144 GB of records + 8 tables + exports): %.0f GB free" % (free / 2**30))
    per = n // n_ranks
    th = nf.synth.zipf_thresholds(keys, 1.1)
    d = dev_stream(torch, nf.synth, n, seed=4, n_keys=keys, thresholds=th, variant=1)
    rec = d.view(torch.int64).view(n, 18)
    # ---- the oracle: ONE Accounter over the whole stream (pkg/flow/account.go:58-124 restated), on a host thread of its own while
    # the GPU routes run (ctypes releases the GIL). Chunk by chunk: the chunk's records come down, orc_local_fold_mt folds them on
    # the host's cores (bit-exact against the one-core Accounter: tests/test_oracle_mt.py) into one record per flow, and those are
    # folded into the Accounter in chunk order = arrival order — AccumulateBase is its own ordered merge of partials (sums, OR,
    # min / max, last-non-zero, first-non-zero, the first record's fields: pkg/model/flow_content.go:28-61). The one-core oracle
    # alone would take ~3 min for 1 B records; with a single host core this does too, and is still right.
    import os
    import threading
    oracle = {}
    CH = 50_000_000
    T = max(1, min(64, (os.cpu_count() or 2) // 2))

    def _oracle_worker():
        try:
            acc = O.Accounter(1 << 25, 0)
            took = 0
            for a in range(0, n, CH):
                b = min(n, a + CH)
                host = d[a * 144:b * 144].cpu().numpy()
                if T >= 2:
                    folded, n_fl, _, _, _, flows = O.local_fold_mt(host, T, 1 << 25, want_flows=True)
                    assert folded == b - a and n_fl == len(flows)
                    assert acc.ingest(flows) == len(flows), "the oracle's map filled up"
                else:
                    assert acc.ingest(host) == b - a
                took += b - a
                del host
            oracle["records"] = took
            oracle["flows"] = acc.evict()                               # sorted by key
            acc.close()
        except BaseException as exc:                                    # reported by the main thread
            oracle["error"] = exc

    worker = threading.Thread(target=_oracle_worker, daemon=True)
    worker.start()
    # ---- what the stream says
    h = _key_hash64(torch, rec)
    n_distinct = int(torch.unique(h).numel())
    del h
    assert 8_000_000 < n_distinct <= keys
    want_bytes = int(rec[:, 7].sum())
    want_packets = int((rec[:, 8] & 0xFFFFFFFF).sum()) & 0xFFFFFFFF
    fl = (rec[:, 8] >> 48) & 0xFFFF
    want_flags = 0
    for b in range(16):
        if bool(((fl >> b) & 1).any()):
            want_flags |= 1 << b
    del fl
    want_end = int(rec[:, 6].max())
    st = rec[:, 5]
    want_start = int(torch.where(st == 0, torch.full_like(st, (1 << 63) - 1), st).min())
    del st
    torch.cuda.synchronize(); torch.cuda.empty_cache()

    def properties(ev):                                                 # ev: int64 [m, 18] on the device
        flags = 0
        f = (ev[:, 8] >> 48) & 0xFFFF
        for b in range(16):
            if bool(((f >> b) & 1).any()):
                flags |= 1 << b
        s = ev[:, 5]
        return (int(ev[:, 7].sum()), int((ev[:, 8] & 0xFFFFFFFF).sum()) & 0xFFFFFFFF, flags, int(ev[:, 6].max()),
                int(torch.where(s == 0, torch.full_like(s, (1 << 63) - 1), s).min()))

    # ---- route B: eight ranks, local fold, partials to their owners
    max_entries = 1 << 23
    tabs = [nf.FlowTable(max_entries=max_entries, table_log2_slots=24) for _ in range(n_ranks)]
    outs = []
    try:
        for r, tab in enumerate(tabs):
            tab.set_sequence(r * per)
            assert tab.ingest_device(d.data_ptr() + r * per * 144, per) == (nf.OK, per)
        seen = [len(t) for t in tabs]
        assert all(3_000_000 < s < max_entries for s in seen), seen
        exp = [torch.empty(s * 24, dtype=torch.int64, device="cuda") for s in seen]
        torch.cuda.synchronize()
        counts = []
        for r, tab in enumerate(tabs):
            rc, c, total = tab.partials_export_device(n_ranks, r, exp[r].data_ptr(), seen[r])
            assert rc == nf.OK and total == sum(c) and c[r] == 0
            counts.append(c)
        for owner in range(n_ranks):
            for src in range(n_ranks):
                if src != owner and counts[src][owner]:
                    tabs[owner].partials_merge_device(n_ranks, owner, exp[src].data_ptr() + sum(counts[src][:owner]) * 192, counts[src][owner])
        for r, tab in enumerate(tabs):
            rc, need = tab.evict_owned_device(n_ranks, r, 0, 0)
            assert rc == nf.TRUNCATED and need > 0
            out = torch.empty(need * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            assert tab.evict_owned_device(n_ranks, r, out.data_ptr(), need) == (nf.OK, need)
            outs.append(out)
        del exp
    finally:
        for t in tabs:
            t.close()
    torch.cuda.synchronize()
    ev_b = torch.cat(outs).view(torch.int64).view(-1, 18)
    del outs
    assert ev_b.shape[0] == n_distinct
    # ---- route A: one handle, the stream in eight calls
    with nf.FlowTable(max_entries=1 << 24) as tab:
        # four calls, one per form of the partition queues' entries (csrc/nfagg_ingest_part.hip entry_idx_mask): 540 M records
        # (>= 2^29: plain 32-bit indices, no retry rounds), 270 M (>= 2^28: index + sub-partition bits), 140 M (>= 2^27: + the
        # "seventh unit needed" flag: pass 2 gathers six 16-byte units of a record whose dscp is zero), 50 M (+ the "sixth unit
        # needed" flag: five units when sampling is zero too and dst_mac's first two bytes say that it is set)
        off = 0
        for m in (540_000_000, 270_000_000, 140_000_000, 50_000_000):
            assert tab.ingest_device(d.data_ptr() + off * 144, m) == (nf.OK, m)
            off += m
        assert off == n and len(tab) == n_distinct
        out = torch.empty(n_distinct * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        assert tab.evict_device(out.data_ptr(), n_distinct) == n_distinct
    ev_a = out.view(torch.int64).view(-1, 18)
    # ---- the same flows, bit for bit (ordered by the key mix; a tie would need two keys with one 64-bit mix: 1e-6)
    a = ev_a[torch.argsort(_key_hash64(torch, ev_a))]
    b = ev_b[torch.argsort(_key_hash64(torch, ev_b))]
    assert torch.equal(a, b), "1 B records: one handle and eight ranks + merge disagree in %d of %d flows" % (int((a != b).any(dim=1).sum()), n_distinct)
    got = properties(ev_a)
    assert got == (want_bytes, want_packets, want_flags, want_end, want_start), (got, (want_bytes, want_packets, want_flags, want_end, want_start))
    # ---- ... and the oracle's Accounter over the same 1 B records: every evicted flow, bit for bit
    worker.join()
    if "error" in oracle:
        raise oracle["error"]
    assert oracle["records"] == n and len(oracle["flows"]) == n_distinct
    got_b = nf.sort_by_key(ev_b.cpu().numpy().view(np.uint8).reshape(-1).view(nf.FLOW_RECORD))
    assert_records_equal(got_b, oracle["flows"], "configs[3] at 1 B records: union of 8 ranks vs ONE oracle Accounter")
