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
