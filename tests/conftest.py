import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import __graft_entry__
    __graft_entry__.ensure_built()          # a fresh checkout has no .so yet (git-ignored): compile the product, never replace it


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def nf():
    """The product package (loads lib/libnfagg.so; raises if it was not built)."""
    import netobserv_ebpf_agent_amd as m
    return m


def as_bytes(a):
    return np.ascontiguousarray(a).view(np.uint8).reshape(len(a), -1)


def assert_records_equal(got, want, what=""):
    """Bit-exact comparison of two key-sorted 144-byte record arrays with a readable diff."""
    g, w = as_bytes(got), as_bytes(want)
    assert g.shape == w.shape, f"{what}: {g.shape[0]} records, expected {w.shape[0]}"
    if not np.array_equal(g, w):
        bad = np.nonzero((g != w).any(axis=1))[0]
        i = int(bad[0])
        cols = np.nonzero(g[i] != w[i])[0]
        raise AssertionError(f"{what}: {len(bad)} of {len(g)} records differ; first at #{i}, byte offsets {cols.tolist()} "
                             f"got {g[i, cols].tolist()} want {w[i, cols].tolist()}")


def dedup_stream(O, n, seed, n_keys, thresholds=None, hot_permille=0, style=0):
    """Scrambled records (variant 1: if_index 1..8, direction 0/1, first-record observed
    lists of 0..6 entries, ssl/tls fields) re-shaped per `style` to reach every branch
    of flows.c:98-143."""
    r = O.gen_stream(n, seed=seed, n_keys=n_keys, thresholds=thresholds, hot_permille=hot_permille, variant=1)
    m = r["metrics"]
    rng = np.random.default_rng(seed)
    if style == 1:      # few interfaces, clean first records: the common shape (two interfaces, both directions)
        m["if_index_first_seen"] = 2 + rng.integers(0, 2, n)
        m["nb_observed_intf"] = 0
        m["observed_intf"] = 0
        m["observed_direction"] = 0
    elif style == 2:    # many interfaces (capacity cut-off), zero if_index, odd direction bytes, BOTH pre-set
        m["if_index_first_seen"] = rng.integers(0, 14, n)
        m["direction_first_seen"] = rng.choice(np.array([0, 1, 1, 0, 2, 3, 7], dtype=np.uint8), n)
        m["nb_observed_intf"] = rng.choice(np.array([0, 0, 1, 2, 5, 6, 7, 255], dtype=np.uint8), n)
        m["observed_intf"] = rng.integers(0, 14, (n, 6))
        m["observed_direction"] = rng.integers(0, 4, (n, 6))
        m["ssl_version"] = rng.choice(np.array([0, 0, 0, 0x0303, 0x0304], dtype=np.uint16), n)
        m["tls_types"] = rng.choice(np.array([0, 1, 2, 2, 4], dtype=np.uint8), n)
        m["tls_cipher_suite"] = rng.choice(np.array([0, 0x1301, 0x1302, 0xc02f], dtype=np.uint16), n)
        m["tls_key_share"] = rng.choice(np.array([0, 0x001d, 0x0017], dtype=np.uint16), n)
    elif style == 3:    # 64-bit end values (both tagged halves matter), end going backwards
        m["end"] = rng.integers(0, 1 << 63, n, dtype=np.uint64) * rng.integers(0, 2, n, dtype=np.uint64)
        m["if_index_first_seen"] = rng.integers(0, 4, n)
    return r
