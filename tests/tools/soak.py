#!/usr/bin/env python3
"""Test infrastructure (lives under tests/ because it uses the oracle as its checker). Randomised soak of the ingest/evict
path against the oracle (run on the GPU box): random stream shapes, batch
splits, kernel routings, table sizes and modes; every eviction compared bit for bit. Usage: python tests/tools/soak.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import netobserv_ebpf_agent_amd as nf
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
rounds = recs_total = 0
while time.time() < t_end:
    n = int(rng.choice([50_000, 400_000, 1_500_000, 3_500_000, 5_000_000, 9_000_000]))
    keys = int(rng.choice([1, 50, 3_000, 100_000, 700_000, 3_000_000]))      # 3 M: pass-2 partitions overflow their caches (retry rounds)
    s = float(rng.choice([0.0, 0.8, 1.1, 1.6]))
    hot = int(rng.choice([0, 0, 500, 900, 999]))
    dedup = bool(rng.integers(0, 4) == 0)
    variant = int(rng.choice([1, 2] if dedup else [1, 1, 1, 0]))       # stream variant: scrambled fields (2: interfaces for dedup)
    ingest_variant = int(rng.choice([0, 0, 0, 7, 10, 11, 1, 17])) if not dedup else int(rng.choice([0, 0, 1, 10, 16]))    # 16: partition pass sorted first
    max_entries = int(rng.choice([1 << 20, 1 << 23, max(2, keys // 3), keys + 5]))
    if max_entries < keys:          # evict-on-full: bound the number of eviction round trips of a round
        n = min(n, 50_000 if max_entries < 1000 else 2_000_000)
    sketches = (nf.SKETCH_CM | nf.SKETCH_HLL) if (not dedup and rng.integers(0, 3) == 0) else 0
    n_shards = int(rng.choice([1, 1, 2, 8]))
    shard_id = int(rng.integers(0, n_shards))
    seed = int(rng.integers(1, 1 << 30))
    th = O.zipf_thresholds(keys, s) if s > 0 and keys > 1 else None
    recs = O.gen_stream(n, seed=seed, n_keys=keys, thresholds=th, hot_permille=hot, variant=variant)
    print("round", rounds, dict(n=n, keys=keys, s=s, hot=hot, dedup=dedup, variant=variant, ingest_variant=ingest_variant, max_entries=max_entries,
                                seed=seed, n_shards=n_shards, shard_id=shard_id, sketches=sketches), file=sys.stderr, flush=True)
    if rng.integers(0, 4) == 0:
        # ---- local fold across "GPUs" (round 4: both modes): R unsharded handles on this device fold ragged chunks of the ONE stream
        # with job-global sequence numbers; partials to the owners, merge, (kernel-dedup: join,) evict owned; union vs ONE table
        import torch
        n = min(n, 3_000_000); recs = recs[:n]
        R = int(rng.integers(1, 7))
        print("   local fold, R =", R, file=sys.stderr, flush=True)
        want1 = O.run_accounter(recs, 1 << 24, 1 if dedup else 0)[0][1]
        me = 1 << 22
        tabs = [nf.FlowTable(max_entries=me, mode=nf.MODE_KERNEL_DEDUP if dedup else nf.MODE_ACCOUNTER, local_fold=True,
                             ingest_variant=ingest_variant if ingest_variant in (0, 1, 10, 16) else 0) for _ in range(R)]
        keep, off = [], 0
        while off < n:
            c = min(n - off, int(rng.choice([1, 63, 5_000, 70_000, 400_000, 1_500_000])))
            dbuf = torch.from_numpy(recs[off:off + c].view(np.uint8).reshape(-1).copy()).cuda(); torch.cuda.synchronize()
            keep.append(dbuf)
            r = int(rng.integers(0, R))
            tabs[r].set_sequence(off)
            assert tabs[r].ingest_device(dbuf.data_ptr(), c) == (nf.OK, c)
            off += c
        pb = tabs[0].partial_bytes
        live = [len(t) for t in tabs]
        exp = [torch.empty(max(l, 1) * pb // 8, dtype=torch.int64, device="cuda") for l in live]
        torch.cuda.synchronize()
        cnts = []
        for r, t in enumerate(tabs):
            rc, c, tot = t.partials_export_device(R, r, exp[r].data_ptr(), live[r])
            assert rc == nf.OK
            cnts.append(c)
        for o in range(R):
            for src in range(R):
                if src != o and cnts[src][o]:
                    tabs[o].partials_merge_device(R, o, exp[src].data_ptr() + sum(cnts[src][:o]) * pb, cnts[src][o])
        parts = []
        for r, t in enumerate(tabs):
            rc, need = t.evict_owned_device(R, r, 0, 0)
            ob = torch.empty(max(need, 1) * 144 + 16, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
            if need:
                assert t.evict_owned_device(R, r, ob.data_ptr(), need) == (nf.OK, need)
            parts.append(ob[: need * 144].cpu().numpy().view(nf.FLOW_RECORD))
        for t in tabs:
            t.close()
        got1 = nf.sort_by_key(np.concatenate(parts))
        desc = dict(local_fold=True, R=R, n=n, keys=keys, s=s, hot=hot, dedup=dedup, variant=variant, ingest_variant=ingest_variant, seed=seed)
        assert got1.tobytes() == want1.tobytes(), ("local fold union differs", desc)
        rounds += 1; recs_total += n
        continue
    mine = recs if n_shards == 1 else recs[nf.distributed.shard_ids(recs.view(nf.FLOW_RECORD), n_shards) == shard_id]
    want = O.run_accounter(mine, max_entries, 1 if dedup else 0)
    cuts = np.sort(rng.integers(0, n, int(rng.integers(0, 6))))
    bounds = [0, *cuts.tolist(), n]
    got = []
    with nf.FlowTable(max_entries=max_entries, mode=nf.MODE_KERNEL_DEDUP if dedup else nf.MODE_ACCOUNTER, sketches=sketches,
                      cm_log2_width=14, hll_p=10, ingest_variant=ingest_variant, n_shards=n_shards, shard_id=shard_id,
                      staging_records=int(rng.choice([0, 1 << 16, 1 << 22, 1 << 23]))) as tab:
        view = recs.view(nf.FLOW_RECORD)
        for a, b in zip(bounds[:-1], bounds[1:]):
            off = a
            while off < b:
                rc, c = tab.ingest(view[off:b])
                off += c
                if rc == nf.FULL:
                    got.append(("full", nf.sort_by_key(tab.evict(nf.REASON_FULL))))
        got.append(("closing", nf.sort_by_key(tab.evict(nf.REASON_CLOSING))))
        if sketches:
            cm_s, cm_d, hs, hd = O.sketches(mine, 4, 14, 10)
            assert np.array_equal(tab.sketch_snapshot(nf.CM_SRC), cm_s) and np.array_equal(tab.sketch_snapshot(nf.HLL_DST), hd)
    desc = dict(n=n, keys=keys, s=s, hot=hot, dedup=dedup, variant=variant, ingest_variant=ingest_variant, max_entries=max_entries, seed=seed, cuts=cuts.tolist(), n_shards=n_shards, shard_id=shard_id)
    assert [r for r, _ in got] == [r for r, _ in want], ("eviction sequence", desc)
    for k, ((_, g), (_, w)) in enumerate(zip(got, want)):
        assert g.tobytes() == w.tobytes(), ("eviction %d differs" % k, desc)
    rounds += 1; recs_total += n
print(f"soak ok: {rounds} rounds, {recs_total} records, every eviction bit-exact vs the oracle")
