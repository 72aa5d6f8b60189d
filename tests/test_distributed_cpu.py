"""The N>1 path on CPU: world_size 2, gloo. Records shard by flow-key hash; each
rank folds only its shard (oracle stands in for the GPU here); the sketch arrays
are merged with the same collective code the GPU ranks use (all_reduce SUM / MAX);
the union of the shards' evictions and the merged sketches must equal the
unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import netobserv_ebpf_agent_amd as nf
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    th = O.zipf_thresholds(3000, 1.1)
    recs = O.gen_stream(40000, seed=12, n_keys=3000, thresholds=th, variant=1)     # every rank sees the same stream
    mine = nf.distributed.partition(recs.view(nf.FLOW_RECORD), world)[rank]
    assert all(nf.shard_of(r["id"].tobytes(), world) == rank for r in mine[:100])
    ev = O.run_accounter(mine, 1 << 20)[0][1]
    cm_s, cm_d, hs, hd = O.sketches(mine, 4, 14, 12)
    cm = [torch.from_numpy(cm_s.view(np.int64)), torch.from_numpy(cm_d.view(np.int64))]
    hll = [torch.from_numpy(hs.astype(np.uint8)), torch.from_numpy(hd.astype(np.uint8))]     # one byte per register, as on the device
    nf.distributed.merge_sketches(cm, hll)
    # gather the shards' evictions on rank 0
    gathered = [None] * world
    dist.all_gather_object(gathered, ev.tobytes())
    if rank == 0:
        whole = O.run_accounter(recs, 1 << 20)[0][1]
        allev = np.concatenate([np.frombuffer(b, dtype=O.FLOW_RECORD) for b in gathered])
        allev = nf.sort_by_key(allev.view(nf.FLOW_RECORD))
        wcm_s, wcm_d, whs, whd = O.sketches(recs, 4, 14, 12)
        ok = (allev.tobytes() == whole.tobytes()
              and np.array_equal(cm[0].numpy().view(np.uint64), wcm_s) and np.array_equal(cm[1].numpy().view(np.uint64), wcm_d)
              and np.array_equal(hll[0].numpy().astype(np.uint8), whs) and np.array_equal(hll[1].numpy().astype(np.uint8), whd))
        est = nf.hll_estimate_from_histogram(np.bincount(hll[0].numpy(), minlength=65).astype(np.uint32), 12)
        ok = ok and abs(est - O.hll_estimate(whs, 12)) <= np.spacing(est)
        # heavy hitters: every rank's top-k over ITS evicted flows with the MERGED sketch, merged on rank 0 = the unsharded answer
        tops = [O.cm_topk(cm[0].numpy().view(np.uint64), 4, 14, np.frombuffer(b, dtype=O.FLOW_RECORD), 0, 25) for b in gathered]
        want_top = O.cm_topk(wcm_s, 4, 14, whole, 0, 25)
        ok = ok and nf.distributed.merge_topk(tops, 25).tobytes() == want_top.tobytes()
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_fold_and_sketch_merge():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_partition_is_stable_and_disjoint():
    sys.path.insert(0, ROOT)
    import netobserv_ebpf_agent_amd as nf
    from oracle import oracle as O
    recs = O.gen_stream(5000, seed=3, n_keys=400, variant=1).view(nf.FLOW_RECORD)
    parts = nf.distributed.partition(recs, 8)
    assert sum(len(p) for p in parts) == len(recs)
    ids = nf.distributed.shard_ids(recs, 8)
    for s, p in enumerate(parts):
        assert p.tobytes() == recs[ids == s].tobytes()                      # arrival order kept inside a shard
        assert all(O.lib().orc_shard_of(r["id"].tobytes(), 8) == s for r in p[:50])
