"""libnfagg driven from plain C (tools/c/nfagg_cdriver.c) — no Python, no torch in the process: what a cgo shim sees.
CPU leg: the driver compiles as C11 against include/nfagg.h and links against lib/libnfagg.so alone.
GPU leg: its evictions, protobuf frames and HLL estimate equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "netobserv-ebpf-agent_amd", "lib")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cdriver") / "nfagg_cdriver")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "c", "nfagg_cdriver.c"), "-o", exe, "-L", LIBDIR, "-lnfagg", "-Wl,-rpath," + LIBDIR])
    return exe


def test_c_driver_builds_and_needs_only_libnfagg_and_hip(nf, driver):
    needed = subprocess.check_output(["readelf", "-d", driver], text=True)
    libs = [l.split("[")[1].split("]")[0] for l in needed.splitlines() if "NEEDED" in l]
    assert "libnfagg.so" in libs and not any("torch" in l or "python" in l for l in libs)
    lib_needed = subprocess.check_output(["readelf", "-d", os.path.join(LIBDIR, "libnfagg.so")], text=True)
    assert "libamdhip64" in lib_needed and "torch" not in lib_needed and "python" not in lib_needed


@pytest.mark.gpu
@pytest.mark.parametrize("max_entries,batch,how", [(1 << 16, 70_000, ""), (700, 5_000, ""), (700, 40_000, "account"), (1 << 16, 70_000, "account"),
                                                   (1 << 16, 30_000, "ring"), (700, 5_000, "ring")])
def test_c_driver_matches_oracle(nf, O, driver, tmp_path, max_entries, batch, how):
    """how = "account": nfagg_account from / into nfagg_host_alloc buffers (the cgo shim's flush); "ring": the records come out of a
    BPF-style ring buffer (discarded and wrong-length samples in between) through nfagg_staging_acquire -> nfagg_ringbuf_drain ->
    nfagg_staging_commit; else nfagg_ingest + nfagg_evict."""
    th = O.zipf_thresholds(4000, 1.1)
    recs = O.gen_stream(150_000, seed=17, n_keys=4000, thresholds=th, variant=1)
    recs["metrics"]["if_index_first_seen"] = 2 + (np.arange(len(recs)) % 3)           # eth0 / eth1+udn / unknown
    src = tmp_path / "records.bin"
    recs.tofile(src)
    out = subprocess.check_output([driver, str(src), str(tmp_path / "out"), str(max_entries), str(batch), "1"] + ([how] if how else []), text=True).split("\n")
    want = O.run_accounter(recs, max_entries)
    lines = [l.split() for l in out if l and not l.startswith("hll_src")]
    assert [(r, len(b)) for r, b in want] == [(l[0], int(l[1])) for l in lines]
    got = np.fromfile(tmp_path / "out.records", dtype=O.FLOW_RECORD)
    pos = 0
    for reason, batch_want in want:                               # batches in order, flows inside a batch in any order
        g = got[pos:pos + len(batch_want)]
        pos += len(batch_want)
        assert nf.sort_by_key(g.view(nf.FLOW_RECORD)).tobytes() == batch_want.tobytes(), reason
    assert pos == len(got)
    last = got[len(got) - len(want[-1][1]):]
    names = O.intf_table([(2, None, "eth0", ""), (3, None, "eth1", "default")])
    opts = O.pb_options(1_700_000_000_000_000_000, 3_000_000, bytes(10) + b"\xff\xff" + bytes([10, 0, 0, 1]), names)
    want_pb = b"".join(b"\x0a" + _varint(len(b)) + b for b in O.pb_encode(last, opts))
    assert (tmp_path / "out.pb").read_bytes() == want_pb
    est = float([l for l in out if l.startswith("hll_src")][0].split()[1])
    _, _, hs, _ = O.sketches(recs, 4, 20, 14)
    assert abs(est - O.hll_estimate(hs, 14)) <= np.spacing(est)


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)
