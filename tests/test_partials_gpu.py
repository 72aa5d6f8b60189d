"""Local fold across GPUs with one process per GPU (include/nfagg.h: nfagg_set_sequence, nfagg_partials_export_device,
nfagg_partials_merge_device, nfagg_evict_owned_device), rehearsed on ONE GPU: several unsharded handles on device 0 stand for
the ranks of `bench.py --gpus N`; each folds the slices of ONE stream that "arrive" at it with job-global sequence numbers;
at the tick every rank's flows travel as 192-byte partials to their key-hash owners, the owners merge and evict. The union
of the ranks' evictions must be bit-identical to ONE sequential Accounter (pkg/flow/account.go:58-124; the oracle) over the
whole stream — and `python bench.py --gpus 2` (the command the driver would run) must go through exactly this path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_records_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(O, n, keys, seed, hot=0):
    return O.gen_stream(n, seed=seed, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=hot, variant=1)


class Ranks:
    """N unsharded handles on cuda:0 + the buffers a rank of bench.py holds."""

    def __init__(self, nf, n, max_entries=1 << 18, **kw):
        import torch
        self.nf, self.n, self.torch = nf, n, torch
        self.tabs = [nf.FlowTable(max_entries=max_entries, table_log2_slots=20, **kw) for _ in range(n)]
        self.exp = [torch.zeros(max_entries * 24, dtype=torch.int64, device="cuda") for _ in range(n)]
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
        """export -> exchange (here: pointer arithmetic on one device) -> merge -> evict owned. Returns the union, key-sorted."""
        nf, n, torch = self.nf, self.n, self.torch
        counts = []
        for r in range(n):
            rc, c, total = self.tabs[r].partials_export_device(n, r, self.exp[r].data_ptr(), self.exp[r].numel() // 24)
            assert rc == nf.OK and total == sum(c) and c[r] == 0
            counts.append(c)
        for owner in range(n):
            for src in range(n):
                if src == owner or not counts[src][owner]:
                    continue
                off = sum(counts[src][:owner])
                self.tabs[owner].partials_merge_device(n, owner, self.exp[src].data_ptr() + off * 192, counts[src][owner])
        out = []
        for r in range(n):
            rc, need = self.tabs[r].evict_owned_device(n, r, 0, 0, reason if reason is not None else nf.REASON_TIMEOUT)
            buf = torch.zeros(max(need, 1) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()   # torch fills on ITS stream
            if need:
                assert rc == nf.TRUNCATED
                rc, got = self.tabs[r].evict_owned_device(n, r, buf.data_ptr(), need, reason if reason is not None else nf.REASON_TIMEOUT)
                assert (rc, got) == (nf.OK, need)
            ev = buf.cpu().numpy()[: need * 144].view(nf.FLOW_RECORD)
            assert np.all(nf.distributed.shard_ids(ev, n) == r)             # rank r delivered exactly the flows it owns
            out.append(ev)
        self.keep.clear()
        return nf.sort_by_key(np.concatenate(out)), counts


@pytest.mark.parametrize("n_ranks,hot", [(1, 0), (2, 0), (4, 900), (8, 0), (8, 999)])
def test_ranks_with_contiguous_slices_equal_one_accounter(nf, O, n_ranks, hot):
    """bench.py's layout: rank r holds arrival positions [r n, (r+1) n) of the one stream."""
    recs = _stream(O, 480_000, 50_000, seed=21, hot=hot)
    per = len(recs) // n_ranks
    R = Ranks(nf, n_ranks)
    try:
        for epoch in range(2):                                             # every eviction restarts the sequence at 0
            for r in range(n_ranks):
                R.fold(r, recs[r * per:(r + 1) * per], r * per)
            got, counts = R.tick()
            want = O.run_accounter(recs[: per * n_ranks], 1 << 20)[0][1]
            assert_records_equal(got, want, "epoch %d" % epoch)
            if n_ranks > 1:
                assert sum(map(sum, counts)) > 0
    finally:
        R.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ranks_with_interleaved_ragged_chunks(nf, O, seed):
    """Chunks of any size land on any rank; the sequence number of a chunk's first record is its arrival position. Gaps in the
    numbering (a rank that reserved more than it used) do not matter: only the order does."""
    rng = np.random.default_rng(seed)
    n_ranks = int(rng.integers(2, 7))
    recs = _stream(O, 300_000, int(rng.choice([300, 40_000])), seed=50 + seed, hot=int(rng.choice([0, 600])))
    R = Ranks(nf, n_ranks)
    try:
        off, gap = 0, 0
        while off < len(recs):
            c = min(len(recs) - off, int(rng.choice([1, 63, 4_000, 30_000, 90_000])))
            R.fold(int(rng.integers(0, n_ranks)), recs[off:off + c], off + gap)
            off += c
            gap += int(rng.choice([0, 0, 5, 1000]))
        got, _ = R.tick(nf.REASON_CLOSING)
        assert_records_equal(got, O.run_accounter(recs, 1 << 20)[0][1])
    finally:
        R.close()


def test_export_states_and_errors(nf, O):
    import torch
    recs = _stream(O, 100_000, 8_000, seed=4)
    want = O.run_accounter(recs, 1 << 20)[0][1]
    with nf.FlowTable(max_entries=1 << 16, table_log2_slots=18) as tab:
        d = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).cuda()
        assert tab.ingest_device(d.data_ptr(), len(recs)) == (nf.OK, len(recs))
        # too small: nothing written, the size needed comes back, the handle still takes records
        rc, counts, need = tab.partials_export_device(4, 1, 0, 0)
        assert rc == nf.TRUNCATED and need == sum(counts) and 0 < need < len(want) and counts[1] == 0
        small = torch.zeros(16 * 24, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
        assert tab.partials_export_device(4, 1, small.data_ptr(), 16)[0] == nf.TRUNCATED
        assert tab.ingest_device(d.data_ptr(), 10) == (nf.OK, 10)
        want2 = O.run_accounter(np.concatenate([recs, recs[:10]]), 1 << 20)[0][1].view(nf.FLOW_RECORD)   # the first ten records folded twice
        # NFAGG_SHARD_NONE: every flow leaves, grouped by owner
        buf = torch.zeros(len(want) * 24, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
        rc, counts, total = tab.partials_export_device(4, 0xFFFFFFFF, buf.data_ptr(), len(want))
        assert rc == nf.OK and total == len(want) == sum(counts) and all(counts)
        raw = buf.cpu().numpy().view(np.uint8).reshape(-1, 192)
        keys = np.ascontiguousarray(raw[:, 8:48])
        owner = np.array([nf.shard_of(k.tobytes(), 4) for k in keys[:: max(1, len(keys) // 500)]])
        bounds = np.cumsum([0] + counts)
        idx = np.arange(len(keys))[:: max(1, len(keys) // 500)]
        assert np.array_equal(owner, np.searchsorted(bounds, idx, side="right") - 1)
        assert np.all(raw[:, :8] == 0)                                       # the tag word stays home
        # exported with a real self shard: only the owned eviction may follow
        rc, counts, total = tab.partials_export_device(4, 2, buf.data_ptr(), len(want))
        assert rc == nf.OK and counts[2] == 0
        assert tab.ingest_device(d.data_ptr(), 10) == (nf.FULL, 0)
        with pytest.raises(nf.NfaggError) as ei:
            tab.evict(nf.REASON_TIMEOUT)
        assert ei.value.code == -5 and "nfagg_evict_owned_device" in str(ei.value)
        out = torch.zeros(len(want) * 144, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        rc, n = tab.evict_owned_device(4, 2, out.data_ptr(), len(want), nf.REASON_TIMEOUT)
        ev = out.cpu().numpy()[: n * 144].view(nf.FLOW_RECORD)
        mine = want2[nf.distributed.shard_ids(want2, 4) == 2]
        assert_records_equal(nf.sort_by_key(ev), mine)
        assert len(tab) == 0 and tab.ingest_device(d.data_ptr(), 10) == (nf.OK, 10)   # next epoch
        # a partial of another shard is refused loudly
        tab.partials_merge_device(4, 3, buf.data_ptr(), counts[0])           # segment 0 holds shard 0's flows
        with pytest.raises(nf.NfaggError) as ei:
            tab.sync(); len(tab)
        assert "bailed out" in str(ei.value)


def test_sequence_must_not_go_backwards(nf):
    with nf.FlowTable(max_entries=1 << 12) as tab:
        tab.set_sequence(1000)
        with pytest.raises(nf.NfaggError):
            tab.set_sequence(999)
        tab.set_sequence(1000)


def test_dedup_mode_has_no_partials(nf):
    with nf.FlowTable(max_entries=1 << 12, mode=nf.MODE_KERNEL_DEDUP) as tab:
        with pytest.raises(nf.NfaggError) as ei:
            tab.partials_export_device(2, 0, 0, 0)
        assert "NFAGG_MODE_ACCOUNTER" in str(ei.value)


def _bench(*argv, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NFAGG_BENCH_WATCHDOG="500")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_as_a_plain_process_rehearsed_on_one_gpu(nf):
    """The command line the driver runs for the scaling curve, `python bench.py --gpus N`, as a PLAIN process: it spawns its
    ranks itself. Rehearsed with both ranks on cuda:0 over gloo (RCCL refuses two ranks on one device); the workload must be
    configs[3]'s: one common stream, sketches all-reduced and flows exchanged inside the timed step."""
    n, keys = 600_000, 30_000
    j = _bench("--gpus", "2", "--same-device", "--backend", "gloo", "--records", str(n), "--flows", str(keys), "--steps", "2", "--warmup", "1")
    assert j["n_gpus"] == 2 and j["metric"].startswith("flow-records/s") and j["value"] > 0
    c = j["config"]
    assert "configs[3]" in c["workload"] and "local fold" in c["parallelism"] and "REHEARSAL" in c["parallelism"]
    assert c["member_records_folded"] == [3 * n, 3 * n]                      # warm-up + 2 timed steps, both ranks, nothing skipped
    ex = c["exchange"]
    assert ex["partials_sent"] > 0 and ex["partials_received"] > 0 and ex["all_to_all_ms"] > 0 and ex["sketch_allreduce_ms"] > 0
    # one Accounter over the common stream: as many flows as the stream has distinct keys (a host mirror of the generator)
    from netobserv_ebpf_agent_amd import synth
    th = synth.zipf_thresholds(2 * keys, 1.1)
    whole = synth.stream_host(2 * n, seed=2, n_keys=2 * keys, thresholds=th)
    distinct = len(np.unique(np.ascontiguousarray(whole["id"]).view(np.uint8).reshape(len(whole), 40), axis=0))
    assert c["evicted_flows_per_step"] == distinct
    assert j["roofline"]["alg_bytes_per_record"] == 522 and j["roofline"]["launch_ms"] > 0


@pytest.mark.parametrize("overlap,dedup", [(True, False), (False, False), (True, True)])
def test_bench_gpus_2_evictions_equal_one_accounter(nf, O, overlap, dedup, tmp_path):
    """What `bench.py --gpus 2` EVICTS — with window w's tick (sketch all-reduce, export, all-to-all, merge, evict) running on a host
    thread beside window w + 1's fold on the rank's second table (the default since round 6), and with everything in sequence
    (--no-overlap) — against ONE oracle Accounter (pkg/flow/account.go:58-124) over the common stream: the union of the two ranks'
    last evictions, bit for bit. Two gloo ranks on one device. dedup: configs[4]'s line (--dedup --hot-permille 900: sub-flow
    tables, the sketches fed by the kernel-dedup fold's flushes) against ONE kernel-dedup table (bpf/flows.c:76-143)."""
    n, keys = 500_000, 40_000
    dump = str(tmp_path / "ev")
    argv = ["--gpus", "2", "--same-device", "--backend", "gloo", "--records", str(n), "--flows", str(keys), "--steps", "3", "--warmup", "1",
            "--dump-evictions", dump] + ([] if overlap else ["--no-overlap"]) + (["--dedup", "--hot-permille", "900"] if dedup else [])
    j = _bench(*argv)
    assert j["config"]["windows_overlapped"] is overlap
    from netobserv_ebpf_agent_amd import synth
    th = synth.zipf_thresholds(2 * keys, 1.1)
    whole = synth.stream_host(2 * n, seed=2, n_keys=2 * keys, thresholds=th, hot_permille=900 if dedup else 0, variant=2 if dedup else 0)
    want = O.run_accounter(whole, 1 << 22, mode=1 if dedup else 0)[0][1]
    got = np.concatenate([np.fromfile("%s.%d" % (dump, r), dtype=nf.FLOW_RECORD) for r in range(2)])
    assert len(got) == len(want) == j["config"]["evicted_flows_per_step"]
    assert_records_equal(nf.sort_by_key(got), want, "bench.py --gpus 2 (%s%s): union of the ranks' evictions vs ONE %s"
                         % ("overlapped" if overlap else "in sequence", ", kernel-dedup" if dedup else "", "kernel-dedup table" if dedup else "Accounter"))
    if overlap:
        ov = j["config"]["exchange"]["overlapped"]
        assert ov["wall_ms_per_window"] > 0 and ov["tick_ms_beside_a_fold"] > 0
    if dedup:
        assert j["roofline"]["sketch_launch_ms"] is None           # the sketches are fed by the fold itself: no second pass


def test_bench_presharded_line_still_runs(nf):
    j = _bench("--gpus", "2", "--same-device", "--backend", "gloo", "--presharded", "--no-sketches", "--records", "400000", "--flows", "20000",
               "--steps", "1", "--warmup", "1")
    assert j["n_gpus"] == 2 and "key-hash shards x2" in j["config"]["parallelism"]


def test_bench_n1_line_carries_the_extra_legs(nf):
    """N = 1 default route at reduced size: roofline + cpu_baseline + the bounded extra legs in ONE line."""
    j = _bench("--records", "3000000", "--flows", "100000", "--steps", "2", "--warmup", "1", "--cpu-sample", "500000")
    assert j["n_gpus"] == 1 and j["roofline"]["frac"] > 0 and j["cpu_baseline"]["kind"] == "port"
    ex = j["extra"]
    assert "error" not in ex, ex
    for k in ("configs2", "configs4_shape", "e2e", "cache_max_flows_5000", "cache_max_flows_10000", "cache_max_flows_100000", "shim_small_calls"):
        assert k in ex and "error" not in ex[k], (k, ex.get(k))
    assert j["roofline"]["frac"] <= 1.0 and j["roofline"]["hbm_read_stream_measured_GBs"] > 1000
    for M in ("10000", "100000"):
        leg = ex["cache_max_flows_" + M]
        assert leg["account_host_path"]["evicted_flows"] == leg["account_device_resident"]["evicted_flows"] == leg["account_host_path_page_locked"]["evicted_flows"]
    assert set(ex["shim_small_calls"]["by_call_size"]) >= {"1", "1024", "65536"}
    assert ex["configs2"]["alg_bytes_per_record"] == 522 and ex["configs2"]["Mrecords_per_s"] > 0
    assert ex["e2e"]["Mrecords_per_s"] > 0 and ex["cache_max_flows_5000"]["account_host_path"]["evictions"] > 10
    assert ex["cache_max_flows_5000"]["account_host_path"]["evicted_flows"] == ex["cache_max_flows_5000"]["account_device_resident"]["evicted_flows"]
