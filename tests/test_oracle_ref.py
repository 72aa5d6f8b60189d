"""Pins the oracle's kernel-dedup restatement (nfagg_oracle.c: update_existing_flow / add_observed_intf, mode 1) to the
REFERENCE ITSELF: oracle/_ref/libref_flows.so is bpf/flows.c:76-143 + bpf/types.h compiled with host gcc from
/root/reference where they lie (oracle/Makefile target `ref`, oracle/ref_flows_shim.c = the kernel-environment shim).
The .so is built in the build container and travels to the GPU box; a checkout with neither it nor /root/reference
skips (and then a15's parity is unpinned again — the skip reason says so)."""
import numpy as np
import pytest

from conftest import assert_records_equal, dedup_stream


@pytest.fixture(scope="module")
def ref(O):
    if not O.ref_available():
        pytest.skip("oracle/_ref/libref_flows.so absent and /root/reference not present: dedup parity UNPINNED in this checkout")
    return O.ref_lib()


def same(O, recs, max_entries):
    want = O.run_ref_dedup(recs, max_entries)
    got = O.run_accounter(recs, max_entries, mode=1)
    assert [r for r, _ in got] == [r for r, _ in want]
    for k, ((_, g), (_, w)) in enumerate(zip(got, want)):
        assert_records_equal(g, w, f"oracle vs bpf/flows.c, eviction #{k}")
    return want


@pytest.mark.parametrize("style", [0, 1, 2, 3])
@pytest.mark.parametrize("max_entries", [1 << 20, 50, 1])
def test_oracle_dedup_equals_reference_flows_c(O, ref, style, max_entries):
    th = O.zipf_thresholds(300, 1.1)
    recs = dedup_stream(O, 40000, seed=200 + style, n_keys=300, thresholds=th, style=style)
    want = same(O, recs, max_entries)
    if max_entries > 300 and style in (0, 2):   # the streams do reach the capacity cut-off and the BOTH merge
        m = want[0][1]["metrics"]
        assert (m["nb_observed_intf"] == 6).any() and (m["observed_direction"] == 3).any()


def test_oracle_dedup_equals_reference_hot_flow(O, ref):
    """configs[4] shape: 90 % of the records are one flow alternating over interfaces."""
    th = O.zipf_thresholds(2000, 1.1)
    for seed, style in ((6, 1), (7, 2)):
        same(O, dedup_stream(O, 200000, seed=seed, n_keys=2000, thresholds=th, hot_permille=900, style=style), 1 << 16)


def test_single_packet_records_need_no_packet_fixup(O, ref):
    """With packets == 1 per record the shim's `packets - 1` top-up is zero: the reference's arithmetic alone."""
    recs = dedup_stream(O, 30000, seed=31, n_keys=200, style=2)
    recs["metrics"]["packets"] = 1
    same(O, recs, 1 << 16)


def test_reference_function_directly(O, ref):
    """update_existing_flow on hand-made aggregates, one call at a time, against orc (through a two-record stream)."""
    rng = np.random.default_rng(5)
    recs = dedup_stream(O, 4000, seed=9, n_keys=1, style=2)    # one flow: record 0 is the aggregate, the rest are merged
    agg = recs[0]["metrics"].copy().reshape(1)
    agg.view(np.uint8).reshape(-1)[66:68] = 0
    agg.view(np.uint8).reshape(-1)[100:104] = 0
    for i in range(1, len(recs)):
        ref.ref_update_existing_flow(agg.ctypes.data, recs[i:i + 1].ctypes.data)
    want = O.run_accounter(recs, 10, mode=1)[0][1]
    assert agg.tobytes() == want["metrics"].tobytes()
    assert rng is not None


def test_observed_intf_missed_counter(O, ref):
    """flows.c:133-142: the capacity cut-off is reported (counter raised on non-zero proto)."""
    ref.ref_counters_reset()
    recs = dedup_stream(O, 20000, seed=202, n_keys=50, style=2)
    O.run_ref_dedup(recs, 1 << 16)
    assert ref.ref_counter_observed_intf_missed() > 0
