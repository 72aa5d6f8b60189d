"""CPU only. The claim behind DESIGN.md §4.11b (first stated in §10.4 and HISTORY.md §13.1 item 1): WHERE the evict-on-full loop of Accounter.Account
(pkg/flow/account.go:81-96) cuts a record stream into epochs can be computed without the flow table, from previous-occurrence
links — prev(i) = the index of the previous record of record i's flow in the call, -1 when there is none:

    a record i >= s starts a new flow in the epoch that began at record s   <=>   prev(i) < s
                                                                                 (first epoch of a call: and its flow is not live in the table)
    the epoch ends at the record where the count of such records reaches max_entries + 1 (that record opens the next epoch)

— a prefix count per epoch instead of a chain of dependent table operations. Checked here against the oracle's Accounter driven
the way the reference's TestEvict_MaxEntries drives it (pkg/flow/account_test.go:47-128): same cuts, same flows per eviction.
Nothing on the GPU uses this yet: it pins the design before the kernels are written."""
import numpy as np
import pytest


def prev_links(keys):
    """prev[i] = largest j < i with keys[j] == keys[i], else -1 (what a sort of (key hash, index) pairs + a neighbour check gives)."""
    order = np.lexsort((np.arange(len(keys)), keys))
    prev = np.full(len(keys), -1, dtype=np.int64)
    same = keys[order[1:]] == keys[order[:-1]]
    prev[order[1:][same]] = order[:-1][same]
    return prev


def epoch_cuts(prev, max_entries, live_at_start=None):
    """The records at which an eviction on "full" happens, by the prefix-count rule. live_at_start: bool per record — its flow is
    live in the table when the call starts (matters for the first epoch only), with the number of live flows as second item."""
    n = len(prev)
    cuts, s = [], 0
    live_mask, live = (live_at_start if live_at_start is not None else (np.zeros(n, dtype=bool), 0))
    while True:
        first = not cuts                                             # the epoch the table's live flows belong to (it may end at record 0)
        new = prev[s:] < s
        if first:
            new &= ~live_mask[s:]                                    # a flow the table already holds is no new entry
        budget = max_entries - (live if first else 0)                # entries the epoch may still create
        c = np.cumsum(new)
        over = np.nonzero(c == budget + 1)[0]
        if len(over) == 0:
            return cuts
        s = s + int(over[0])                                          # this record found the map full: evict, then it is stored
        cuts.append(s)


def key_ids(recs):
    ids = np.ascontiguousarray(recs["id"]).view(np.uint8).reshape(len(recs), 40)[:, :39]    # byte 39: Go's blank field
    _, inv = np.unique(ids, axis=0, return_inverse=True)
    return inv.reshape(-1).astype(np.int64)


@pytest.mark.parametrize("n,keys,max_entries,hot", [(20_000, 300, 50, 0), (20_000, 5_000, 500, 0), (30_000, 2_000, 7, 700), (5_000, 50, 2, 0),
                                                     (5_000, 10, 1, 0), (20_000, 400, 400, 0), (20_000, 400, 100_000, 900)])
def test_cuts_from_previous_occurrence_links_equal_the_reference_loop(O, n, keys, max_entries, hot):
    recs = O.gen_stream(n, seed=n + keys + max_entries, n_keys=keys, thresholds=O.zipf_thresholds(keys, 1.1), hot_permille=hot, variant=1)
    want = O.run_accounter(recs, max_entries)
    k = key_ids(recs)
    cuts = epoch_cuts(prev_links(k), max_entries)
    assert len(cuts) == len(want) - 1                                # every eviction but the closing one is an eviction on "full"
    bounds = [0, *cuts, n]
    for e, (reason, ev) in enumerate(want):
        assert reason == ("full" if e < len(cuts) else "closing")
        seg = recs[bounds[e]:bounds[e + 1]]
        assert len(ev) == len(np.unique(k[bounds[e]:bounds[e + 1]])) == (max_entries if e < len(cuts) else len(ev))
        # the flows of the epoch, folded on their own, are the eviction: an epoch is an independent fold once its bounds are known
        alone = O.run_accounter(seg, 1 << 20)
        assert len(alone) == 1 and alone[0][1].tobytes() == ev.tobytes()


def test_an_epoch_that_spans_calls(O):
    """The table holds the tail epoch of the previous call: its flows are not new in the first epoch of this call."""
    max_entries = 60
    recs = O.gen_stream(12_000, seed=5, n_keys=500, thresholds=O.zipf_thresholds(500, 1.1), variant=1)
    k = key_ids(recs)
    whole = epoch_cuts(prev_links(k), max_entries)
    full_at = [c - 1 for c in whole[:3]]                             # calls that end with the map exactly full: the next record evicts
    for split in (1, 37, 5_000, 11_999, *whole[:2], *full_at):
        first = [c for c in whole if c < split]                      # (a cut AT the split is the second call's: its first record finds the map full)
        start = first[-1] if first else 0                            # the epoch in progress when the second call starts
        live_keys = np.unique(k[start:split])
        k2 = k[split:]
        live_mask = np.isin(k2, live_keys)
        second = epoch_cuts(prev_links(k2), max_entries, (live_mask, len(live_keys)))
        assert first + [split + c for c in second] == whole
