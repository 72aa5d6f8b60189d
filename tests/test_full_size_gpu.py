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
