"""The C-ABI boundary: libnfagg.so loads, exports every symbol include/nfagg.h
declares, and the struct layouts agree between the header (as compiled by gcc),
the product's numpy views and the oracle's own restatement. No GPU needed; no
compute call is made."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nfagg.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nfagg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(nf):
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(nf._lib.lib, n), f"libnfagg.so does not export {n}"
    # and the binding knows a signature for each (so no call goes out untyped)
    assert set(names) == set(nf._lib.SIGNATURES)


def test_abi_version(nf):
    assert nf._lib.lib.nfagg_abi_version() == 2


PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "nfagg.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t,f) printf(#t "." #f " %zu\n", offsetof(t,f))
int main(void){
 S(nfagg_flow_id); S(nfagg_flow_metrics); S(nfagg_flow_record); S(nfagg_additional_metrics); S(nfagg_dns_metrics);
 S(nfagg_pkt_drop_metrics); S(nfagg_network_events_metrics); S(nfagg_xlat_metrics); S(nfagg_quic_metrics);
 S(nfagg_config); S(nfagg_stats);
 O(nfagg_flow_id,src_port); O(nfagg_flow_id,transport_protocol); O(nfagg_flow_id,pad_);
 O(nfagg_flow_metrics,packets); O(nfagg_flow_metrics,eth_protocol); O(nfagg_flow_metrics,flags); O(nfagg_flow_metrics,src_mac);
 O(nfagg_flow_metrics,dst_mac); O(nfagg_flow_metrics,if_index_first_seen); O(nfagg_flow_metrics,lock); O(nfagg_flow_metrics,sampling);
 O(nfagg_flow_metrics,direction_first_seen); O(nfagg_flow_metrics,errno_); O(nfagg_flow_metrics,dscp); O(nfagg_flow_metrics,nb_observed_intf);
 O(nfagg_flow_metrics,observed_direction); O(nfagg_flow_metrics,observed_intf); O(nfagg_flow_metrics,ssl_version);
 O(nfagg_flow_metrics,tls_cipher_suite); O(nfagg_flow_metrics,tls_key_share); O(nfagg_flow_metrics,tls_types); O(nfagg_flow_metrics,misc_flags);
 O(nfagg_flow_record,metrics);
 O(nfagg_dns_metrics,errno_); O(nfagg_dns_metrics,name); O(nfagg_additional_metrics,ipsec_encrypted);
 O(nfagg_network_events_metrics,bytes); O(nfagg_network_events_metrics,network_events_idx); O(nfagg_xlat_metrics,zone_id);
 O(nfagg_quic_metrics,seen_short_hdr); O(nfagg_pkt_drop_metrics,latest_state);
 return 0; }
"""


@pytest.fixture(scope="module")
def c_layout():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write(PROBE)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    return {k: int(v) for k, v in (line.split() for line in out.strip().splitlines())}


def test_struct_sizes_match_bpf_types_h(c_layout):
    # sizes verified against bpf/types.h compiled with host gcc
    want = dict(nfagg_flow_id=40, nfagg_flow_metrics=104, nfagg_flow_record=144, nfagg_additional_metrics=32,
                nfagg_dns_metrics=64, nfagg_pkt_drop_metrics=32, nfagg_network_events_metrics=72,
                nfagg_xlat_metrics=56, nfagg_quic_metrics=24)
    for k, v in want.items():
        assert c_layout[k] == v, k


def test_numpy_views_match_header(nf, c_layout):
    def off(dt, f):
        return dt.fields[f][1]
    m = nf.FLOW_METRICS
    for f in ["packets", "eth_protocol", "flags", "src_mac", "dst_mac", "if_index_first_seen", "lock", "sampling",
              "direction_first_seen", "errno_", "dscp", "nb_observed_intf", "observed_direction", "observed_intf",
              "ssl_version", "tls_cipher_suite", "tls_key_share", "tls_types", "misc_flags"]:
        assert off(m, f) == c_layout[f"nfagg_flow_metrics.{f}"], f
    assert off(nf.FLOW_ID, "src_port") == c_layout["nfagg_flow_id.src_port"]
    assert off(nf.FLOW_ID, "transport_protocol") == c_layout["nfagg_flow_id.transport_protocol"]
    assert off(nf.FLOW_RECORD, "metrics") == c_layout["nfagg_flow_record.metrics"] == 40
    assert off(nf.DNS, "name") == c_layout["nfagg_dns_metrics.name"] == 31
    assert off(nf.DNS, "errno_") == c_layout["nfagg_dns_metrics.errno_"]
    assert off(nf.ADDITIONAL, "ipsec_encrypted") == c_layout["nfagg_additional_metrics.ipsec_encrypted"]
    assert off(nf.NETWORK_EVENTS, "bytes") == c_layout["nfagg_network_events_metrics.bytes"]
    assert off(nf.NETWORK_EVENTS, "network_events_idx") == c_layout["nfagg_network_events_metrics.network_events_idx"]
    assert off(nf.XLAT, "zone_id") == c_layout["nfagg_xlat_metrics.zone_id"]
    assert off(nf.QUIC, "seen_short_hdr") == c_layout["nfagg_quic_metrics.seen_short_hdr"]
    assert off(nf.PKT_DROP, "latest_state") == c_layout["nfagg_pkt_drop_metrics.latest_state"]
    assert C.sizeof(nf._lib.Config) == c_layout["nfagg_config"]
    assert C.sizeof(nf._lib.Stats) == c_layout["nfagg_stats"]


def test_oracle_and_product_layouts_agree(nf, O):
    pairs = [(nf.FLOW_ID, O.FLOW_ID), (nf.FLOW_METRICS, O.FLOW_METRICS), (nf.FLOW_RECORD, O.FLOW_RECORD),
             (nf.ADDITIONAL, O.ADDITIONAL), (nf.DNS, O.DNS), (nf.PKT_DROP, O.DROPS), (nf.NETWORK_EVENTS, O.NETEV),
             (nf.XLAT, O.XLAT), (nf.QUIC, O.QUIC)]
    for a, b in pairs:
        assert a.itemsize == b.itemsize
        assert sorted(v[1] for v in a.fields.values()) == sorted(v[1] for v in b.fields.values())


def test_create_without_gpu_fails_loudly(nf):
    """The product has no CPU path: on a box without a GPU nfagg_create must refuse."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(nf.NfaggError) as e:
        nf.FlowTable(16)
    assert e.value.code == nf._lib.ENODEV
    assert "no CPU path" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """Product sources and tools/ must not include, import or link anything under oracle/: only tests/ (incl. tests/tools/),
    __graft_entry__.smoke() and the cpu_baseline leg of bench.py may."""
    for top in ("netobserv-ebpf-agent_amd", "netobserv_ebpf_agent_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".c", ".cpp", ".sh", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "nfagg_oracle" not in text and "from oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    first = bench.index("from oracle import")
    assert bench.count("from oracle import") == 1 and first > bench.index("# ---- CPU baseline"), "bench.py may touch the oracle only in its cpu_baseline leg"
