"""CPU only. The evict-on-full path of nfagg_account with its epochs found first (csrc/nfagg_epoch_par.hip),
restated step by step in numpy and checked against the oracle's Accounter — the design pinned where no GPU is needed:

  sort keys    (top 40 bits of the key hash) << 24 | index, sorted as 64-bit numbers        k_par_hash + the radix sort
  links        prev(i) = nearest position to the left with the same hash bits AND the same key   k_par_links
  cut walk     prev[] streamed in aligned blocks, the epochs walked over the resident block       k_par_cuts
  ranks        position of every new flow among its epoch's new flows, in arrival order           k_par_rank_count / _scan / _write
  segments     [p, first position whose key reaches (hash bits, end of the epoch)) less the other flows with these hash bits,
               folded in arrival order by AccumulateBase (pkg/model/flow_content.go:28-61)         k_par_segfold[_long]

The hash bits can be coarsened (`hash_mask`) so that MANY flows share them: the links and the segment folds must take them apart
by their full keys. tests/test_epoch_boundaries.py pins the cut rule itself; tests/test_account_par_gpu.py runs the kernels."""
import ctypes as C

import numpy as np
import pytest

NONE = 0xFFFFFFFF
IDX_BITS = 24                                                        # a launch takes at most 2^24 records
IDX = (1 << IDX_BITS) - 1
HI = 0xFFFFFFFFFFFFFFFF ^ IDX                                         # the key hash's share of a sort key: its top 40 bits


def key_ids(recs):
    ids = np.ascontiguousarray(recs["id"]).view(np.uint8).reshape(len(recs), 40)[:, :39]    # byte 39: Go's blank field
    _, inv = np.unique(ids, axis=0, return_inverse=True)
    return inv.reshape(-1).astype(np.int64)


def sort_keys(O, recs, hash_mask):
    raw = np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 144)
    ks = np.zeros(len(recs), dtype=np.uint64)
    key = np.zeros(40, dtype=np.uint8)
    for i in range(len(recs)):
        key[:39] = raw[i, :39]
        h = O.lib().orc_key_hash(key.ctypes.data_as(C.c_void_p))
        ks[i] = (h & HI & hash_mask) | i
    return np.sort(ks)                                               # sorted on the hash bits, indices in order below them


def links(ks, kid, search=1 << 30):
    n = len(ks)
    prev = np.full(n, -1, dtype=np.int64)
    overflow = False
    for p in range(n):
        i, hb = int(ks[p]) & IDX, int(ks[p]) >> IDX_BITS
        q, looked = p, 0
        while q > 0:
            q -= 1
            if int(ks[q]) >> IDX_BITS != hb:
                break
            looked += 1
            if looked > search:
                overflow = True
                break
            j = int(ks[q]) & IDX
            if kid[j] == kid[i]:
                prev[i] = j
                break
    return prev, overflow


def cut_walk(prev, max_entries, live0, max_cuts, lanes, per=16):
    """k_par_cuts: blocks of lanes * per records; a step is a block counted through or an epoch that ends in it."""
    n = len(prev)
    span = lanes * per
    s, k, before, first = 0, 0, 0, True
    budget = 0 if live0 >= max_entries else max_entries - live0
    cuts = []
    b, n_blocks = 0, (n + span - 1) // span
    while b < n_blocks and k < max_cuts:
        idx = np.arange(b * span, min(n, (b + 1) * span))
        cur = prev[idx]
        is_new = (idx >= s) & ((cur == -1) if first else (cur < s))
        counts = np.add.reduceat(is_new.astype(np.int64), np.arange(0, len(idx), per))         # per lane
        total = int(counts.sum())
        if before + total > budget:
            excl = before + np.concatenate([[0], np.cumsum(counts)[:-1]])
            need = budget - excl
            lane = int(np.nonzero((need >= 0) & (need < counts))[0][0])
            bits = np.nonzero(is_new[lane * per:(lane + 1) * per])[0]
            f = int(idx[lane * per + bits[int(need[lane])]])
            cuts.append(f)
            k += 1; s = f; before = 0; first = False; budget = max_entries
        else:
            before += total
            b += 1
    return cuts, before


def ranks(prev, cuts, max_entries, tile=64):
    """k_par_rank_count / _scan / _write: ONE running count of the heads over tiles of the records [cuts[0], cuts[-1]) — every
    complete epoch holds exactly max_entries heads, so t * max_entries + (heads of epoch t before i) = heads in [cuts[0], i).
    The check that makes that a fact: the first record of every epoch is a head at position t * max_entries, and the total."""
    pos = np.full(len(prev), NONE, dtype=np.int64)
    i_lo, i_hi = cuts[0], cuts[-1]
    cuts_a = np.asarray(cuts, dtype=np.int64)
    idx = np.arange(i_lo, i_hi)
    t_of = np.searchsorted(cuts_a, idx, side="right") - 1          # par_epoch_of: the largest t with cuts[t] <= i
    head = prev[idx] < cuts_a[t_of]
    n_tiles = (len(idx) + tile - 1) // tile
    tile_cnt = np.array([int(head[k * tile:(k + 1) * tile].sum()) for k in range(n_tiles)], dtype=np.int64)
    tile_off = np.concatenate([[0], np.cumsum(tile_cnt)[:-1]]) if n_tiles else tile_cnt
    assert int(tile_cnt.sum()) == (len(cuts) - 1) * max_entries      # k_par_rank_scan: *bad = 2
    for k in range(n_tiles):
        h = head[k * tile:(k + 1) * tile]
        p = tile_off[k] + np.cumsum(h) - h
        sl = idx[k * tile:(k + 1) * tile]
        pos[sl[h]] = p[h]
        first = sl == cuts_a[t_of[k * tile:(k + 1) * tile]]          # k_par_rank_write: *bad = 1
        assert h[first].all() and (p[first] == t_of[k * tile:(k + 1) * tile][first] * max_entries).all()
    return pos


def canonical(raw):
    r = raw.copy()
    r[39] = 0; r[40 + 66:40 + 68] = 0; r[40 + 100:40 + 104] = 0
    return r


def fold_segments(O, recs, ks, kid, pos, cuts, max_entries):
    raw = np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 144)
    n_mid = len(cuts) - 1
    out = np.zeros((n_mid * max_entries, 144), dtype=np.uint8)
    written = np.zeros(n_mid * max_entries, dtype=bool)
    longest = 0
    for p in range(len(ks)):
        i = int(ks[p]) & IDX
        if pos[i] == NONE:
            continue
        limit = (int(ks[p]) & HI) | cuts[int(pos[i]) // max_entries + 1]
        end = p + 1 + int(np.searchsorted(ks[p + 1:], np.uint64(limit), side="left"))
        acc = canonical(raw[i])
        members = 1
        for q in range(p + 1, end):
            j = int(ks[q]) & IDX
            if kid[j] != kid[i]:
                continue                                             # another flow with these hash bits
            assert pos[j] == NONE
            other = np.ascontiguousarray(raw[j, 40:])
            O.lib().orc_accumulate_base(acc[40:].ctypes.data_as(C.c_void_p), other.ctypes.data_as(C.c_void_p))
            members += 1
        longest = max(longest, members)
        assert not written[pos[i]]
        out[pos[i]] = acc; written[pos[i]] = True
    assert written.all()
    return out, longest


def run_model(O, recs, max_entries, hash_mask=0xFFFFFFFFFFFFFFFF, lanes=8, live=None):
    """Returns the evictions of the middle epochs (list of key-sorted arrays), the cuts, the flows of the epoch in progress."""
    kid = key_ids(recs)
    ks = sort_keys(O, recs, hash_mask)
    prev, _ = links(ks, kid)
    live0 = 0
    if live is not None:                                             # k_par_live: first occurrences whose flow the table holds
        live_keys, live0 = live                                      # (a set of 39-byte keys, their number)
        raw = np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 144)
        held = np.array([raw[i, :39].tobytes() in live_keys for i in range(len(recs))])
        prev[(prev == -1) & held] = -2
    cuts, tail_flows = cut_walk(prev, max_entries, live0, 65535, lanes)
    if len(cuts) < 2:
        return [], cuts, tail_flows
    pos = ranks(prev, cuts, max_entries)
    out, longest = fold_segments(O, recs, ks, kid, pos, cuts, max_entries)
    ev = out.reshape(-1).view(O.FLOW_RECORD).reshape(len(cuts) - 1, max_entries)
    return [sort_records(e) for e in ev], cuts, tail_flows


def sort_records(a):
    b = np.ascontiguousarray(a).view(np.uint8).reshape(len(a), 144)
    order = np.lexsort(b[:, :40].T[::-1])
    return np.ascontiguousarray(a)[order]


@pytest.mark.parametrize("n,keys,max_entries,hot,lanes", [(6_000, 300, 50, 0, 8), (6_000, 1_500, 200, 0, 64), (8_000, 600, 7, 700, 4),
                                                           (3_000, 50, 2, 0, 2), (3_000, 10, 1, 0, 1), (6_000, 400, 400, 0, 8)])
@pytest.mark.parametrize("hash_mask", [0xFFFFFFFFFFFFFFFF, 0x0000000F00000000])      # all 40 hash bits / four of them: sixteen buckets
def test_segment_folds_of_the_sorted_call_equal_the_reference_loop(O, n, keys, max_entries, hot, lanes, hash_mask):
    recs = O.gen_stream(n, seed=n + keys + max_entries, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=hot, variant=1)
    want = O.run_accounter(recs, max_entries)
    got, cuts, tail_flows = run_model(O, recs, max_entries, hash_mask, lanes)
    assert len(cuts) == len(want) - 1                                # every eviction but the closing one is an eviction on "full"
    assert tail_flows == len(want[-1][1])                            # what the table holds once the last epoch's records are folded
    # eviction 0 is the table's (the first epoch of the call); the middle epochs come from the segment folds
    for t, g in enumerate(got):
        assert want[t + 1][0] == "full"
        assert g.tobytes() == sort_records(want[t + 1][1]).tobytes(), "middle epoch %d" % t


def test_an_epoch_that_spans_calls_and_the_link_search_bound(O):
    max_entries = 60
    recs = O.gen_stream(9_000, seed=5, n_keys=500, thresholds=O.zipf_thresholds(500, 1.1), variant=1)
    kid = key_ids(recs)
    want = O.run_accounter(recs, max_entries)
    whole_cuts = run_model(O, recs, max_entries)[1]
    for split in (1, 37, 5_000, whole_cuts[0], whole_cuts[1] - 1):
        first = [c for c in whole_cuts if c < split]
        start = first[-1] if first else 0
        raw = np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 144)
        live_keys = {raw[i, :39].tobytes() for i in range(start, split)}
        got, cuts, _ = run_model(O, recs[split:], max_entries, live=(live_keys, len(live_keys)))
        assert first + [split + c for c in cuts] == whole_cuts
        k0 = len(first)
        for t, g in enumerate(got):
            assert g.tobytes() == sort_records(want[k0 + t + 1][1]).tobytes()
    # sixteen hash buckets for 500 flows: some previous occurrences lie further to the left than a bounded search looks
    ks = sort_keys(O, recs, 0x0000000F00000000)
    exact, over = links(ks, kid)
    bounded, over64 = links(ks, kid, search=64)
    assert not over and over64 and (bounded != exact).any()          # the kernel raises the flag and the call takes the chain
