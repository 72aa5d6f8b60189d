"""Multi-GPU plumbing: one process per GPU, records sharded by flow-key hash.

Flow state needs no collective (shards are disjoint). The only exchange is the
per-tick merge of the sketch arrays: Count-Min counters add, HyperLogLog
registers take the maximum. torch.distributed is used as plumbing (backend
"nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests); the tensors wrap
the device buffers the library updates in place (nfagg_config.ext_sketch).
"""
import ctypes as C

import numpy as np

from . import _lib as L


def shard_ids(records: np.ndarray, n_shards: int) -> np.ndarray:
    """nfagg_shard_ids: the GPU each record belongs to (host-side router)."""
    r = np.ascontiguousarray(records)
    n = r.nbytes // 144
    out = np.zeros(n, dtype=np.uint32)
    L.lib.nfagg_shard_ids(r.ctypes.data_as(C.c_void_p), n, n_shards, out.ctypes.data_as(C.c_void_p))
    return out


def partition(records: np.ndarray, n_shards: int):
    """Stable partition of a host batch into per-shard batches (arrival order kept inside a shard)."""
    ids = shard_ids(records, n_shards)
    return [records[ids == s] for s in range(n_shards)]


def merge_sketches(cm_tensors, hll_tensors, group=None):
    """Per-tick sketch merge across ranks, in place: CM sum, HLL max."""
    import torch.distributed as dist
    for t in cm_tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    for t in hll_tensors:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)


def merge_topk(per_rank_lists, k: int) -> np.ndarray:
    """Node-wide heavy hitters from the ranks' nfagg_cm_topk lists, taken AFTER merge_sketches (every rank's Count-Min
    then holds the node-wide counters, so an address carries the same estimate wherever it shows up): union, one row per
    address, estimate descending then address bytes ascending, first k. The shards partition FLOWS, not addresses — an
    address can appear in several ranks' lists."""
    rows = np.concatenate([np.asarray(l) for l in per_rank_lists]) if len(per_rank_lists) else np.zeros(0, dtype=[("ip", "u1", 16), ("estimate", "<u8")])
    best = {}
    for r in rows:
        best[r["ip"].tobytes()] = int(r["estimate"])
    order = sorted(best.items(), key=lambda kv: (-kv[1], kv[0]))[:k]
    out = np.zeros(len(order), dtype=rows.dtype)
    for i, (ip, est) in enumerate(order):
        out[i]["ip"] = np.frombuffer(ip, dtype=np.uint8)
        out[i]["estimate"] = est
    return out
