"""Pins the CPU oracle to the reference's own known-answer tests.

Each test names the reference test it encodes. These are CPU tests: they are
the evidence that oracle/ restates the reference before it is trusted as the
checker of the HIP path.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import as_bytes

# ---- pkg/model/record_test.go:19-102 TestRecordBinaryEncoding ----
RECORD_VECTOR = bytes([
    0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0xff, 0xff, 0x06, 0x07, 0x08, 0x09,
    0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0xff, 0xff, 0x0a, 0x0b, 0x0c, 0x0d,
    0x0e, 0x0f, 0x10, 0x11, 0x12, 0x00, 0x00, 0x00,
    0x13, 0x14, 0x15, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x13, 0x14, 0x15, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x13, 0x14, 0x15, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x06, 0x07, 0x08, 0x09, 0x01, 0x02, 0x13, 0x14,
    0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b, 0x0c, 0x0d, 0x0e, 0x0f,
    0x13, 0x14, 0x15, 0x16, 0x00, 0x00, 0x00, 0x00, 0x02, 0x00, 0x00, 0x00,
    0x03, 0x33, 0x60, 0x02, 0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
    0x07, 0x00, 0x00, 0x00, 0x08, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
    0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
    0x03, 0x03, 0x00, 0x00, 0x00, 0x00, 0x21, 0x00, 0x00, 0x00, 0x00, 0x00,
])


def _check_record_vector(rec_dtype, names):
    assert len(RECORD_VECTOR) == 144
    r = np.frombuffer(RECORD_VECTOR, dtype=rec_dtype)[0]
    i, m = r["id"], r["metrics"]
    n = names
    assert bytes(i["src_ip"]) == bytes([0] * 10 + [0xff, 0xff, 6, 7, 8, 9])
    assert bytes(i["dst_ip"]) == bytes([0] * 10 + [0xff, 0xff, 0x0a, 0x0b, 0x0c, 0x0d])
    assert i["src_port"] == 0x0f0e and i["dst_port"] == 0x1110
    assert i[n["proto"]] == 0x12 and i["icmp_type"] == 0 and i["icmp_code"] == 0
    assert m[n["dir"]] == 3 and m["if_index_first_seen"] == 0x16151413 and m["eth_protocol"] == 0x0201
    assert bytes(m["src_mac"]) == bytes([4, 5, 6, 7, 8, 9]) and bytes(m["dst_mac"]) == bytes([0xa, 0xb, 0xc, 0xd, 0xe, 0xf])
    assert m["packets"] == 0x09080706 and m["bytes"] == 0x1a19181716151413
    assert m[n["start"]] == 0x1a19181716151413 and m[n["end"]] == 0x1a19181716151413
    assert m["flags"] == 0x1413 and m[n["errno"]] == 0x33 and m["dscp"] == 0x60 and m["sampling"] == 2
    assert m["nb_observed_intf"] == 2
    assert m["observed_intf"].tolist() == [7, 8, 0, 0, 0, 0] and m["observed_direction"].tolist() == [1, 0, 0, 0, 0, 0]
    assert m["ssl_version"] == 0x0303 and m["tls_types"] == 0x21 and m["tls_cipher_suite"] == 0 and m["misc_flags"] == 0


def test_record_binary_encoding_oracle_layout(O):
    _check_record_vector(O.FLOW_RECORD, dict(proto="proto", dir="direction_first_seen", start="start", end="end", errno="err_no"))


def test_record_binary_encoding_product_layout(nf):
    _check_record_vector(nf.FLOW_RECORD, dict(proto="transport_protocol", dir="direction_first_seen",
                                              start="start_mono_time_ts", end="end_mono_time_ts", errno="errno_"))


# ---- pkg/model/record_test.go:193-347 per-struct binary layouts ----
def test_dns_metrics_binary_encoding(O, nf):
    b = bytes([0x10, 0, 0, 0, 0, 0, 0, 0, 0xFF, 0, 0, 0, 0, 0, 0, 0, 0x11, 0x12, 0x13, 0x14, 0x15, 0x16, 0x17, 0x18,
               1, 0, 0x80, 0, 3, 0, 0]) + b"test.example.com" + bytes(16) + bytes(1)
    assert len(b) == 64
    for dt, names in ((O.DNS, ("start", "end", "err_no")), (nf.DNS, ("start_mono_time_ts", "end_mono_time_ts", "errno_"))):
        m = np.frombuffer(b, dtype=dt)[0]
        assert m[names[0]] == 0x10 and m[names[1]] == 0xFF and m["eth_protocol"] == 3 and m["id"] == 1
        assert m["flags"] == 0x80 and m["latency"] == 0x1817161514131211 and m[names[2]] == 0
        assert bytes(m["name"])[:16] == b"test.example.com" and bytes(m["name"])[16:] == bytes(16)


def test_pkt_drops_binary_encoding(O, nf):
    b = bytes([0x10, 0, 0, 0, 0, 0, 0, 0, 0xFF, 0, 0, 0, 0, 0, 0, 0, 0x14, 0x15, 0x12, 0x13, 0x11, 0, 0, 0, 0x1c, 0x1d, 3, 0, 0x1e, 0, 0, 0])
    for dt in (O.DROPS, nf.PKT_DROP):
        m = np.frombuffer(b, dtype=dt)[0]
        assert m["eth_protocol"] == 3 and m["packets"] == 0x1312 and m["bytes"] == 0x1514
        assert m["latest_flags"] == 0x1d1c and m["latest_state"] == 0x1e and m["latest_drop_cause"] == 0x11


def test_network_events_binary_encoding(O, nf):
    b = bytes([0x10] + [0] * 7 + [0xFF] + [0] * 7) + bytes(32) + bytes(8) + bytes(8) + bytes([3, 0, 1]) + bytes(5)
    assert len(b) == 72
    for dt in (O.NETEV, nf.NETWORK_EVENTS):
        m = np.frombuffer(b, dtype=dt)[0]
        assert m["eth_protocol"] == 3 and m["network_events_idx"] == 1 and not m["network_events"].any()


def test_xlat_binary_encoding(O, nf):
    b = bytes([0x10] + [0] * 7 + [0xFF] + [0] * 7) + bytes(32) + bytes([0, 0, 0, 0, 2, 0, 3, 0])
    assert len(b) == 56
    for dt in (O.XLAT, nf.XLAT):
        m = np.frombuffer(b, dtype=dt)[0]
        assert m["eth_protocol"] == 3 and m["zone_id"] == 2 and m["sport"] == 0 and m["dport"] == 0


def test_additional_metrics_binary_encoding(O, nf):
    b = bytes([0x10] + [0] * 7 + [0xFF] + [0] * 7 + [0xad, 0xde, 0xef, 0xbe, 0xef, 0xbe, 0xad, 0xde, 1, 0, 0, 0, 3, 0, 1, 0])
    m = np.frombuffer(b, dtype=O.ADDITIONAL)[0]
    assert m["flow_rtt"] == 0xdeadbeefbeefdead and m["ipsec_encrypted"] == 1 and m["ipsec_ret"] == 1 and m["eth_protocol"] == 3
    m = np.frombuffer(b, dtype=nf.ADDITIONAL)[0]
    assert m["flow_rtt"] == 0xdeadbeefbeefdead and m["ipsec_encrypted"] == 1 and m["ipsec_encrypted_ret"] == 1


# ---- pkg/flow/account_test.go ----
def _ip4(a, b, c, d):
    return [0] * 10 + [0xff, 0xff, a, b, c, d]


SRC1, SRC2 = _ip4(0x12, 0x34, 0x56, 0x78), _ip4(0xaa, 0xbb, 0xcc, 0xdd)
DST1, DST2 = _ip4(0x43, 0x21, 0x00, 0xff), _ip4(0x11, 0x22, 0x33, 0x44)
K1 = dict(src_port=333, dst_port=8080, src_ip=SRC1, dst_ip=DST1)
K2 = dict(src_port=12, dst_port=8080, src_ip=SRC2, dst_ip=DST1)
K3 = dict(src_port=333, dst_port=443, src_ip=SRC1, dst_ip=DST2)


def mk(dtype, key, names, **metrics):
    r = np.zeros(1, dtype=dtype)
    for k, v in key.items():
        r["id"][k] = v
    for k, v in metrics.items():
        r["metrics"][names.get(k, k)] = v
    return r


ON = dict(start="start", end="end")


def test_evict_max_entries_oracle(O):
    """account_test.go:47-128 TestEvict_MaxEntries against the oracle."""
    acc = O.Accounter(2)
    recs = np.concatenate([
        mk(O.FLOW_RECORD, K1, ON, bytes=123, packets=1, start=123, end=123, flags=1),
        mk(O.FLOW_RECORD, K2, ON, bytes=456, packets=1, start=456, end=456, flags=1),
        mk(O.FLOW_RECORD, K1, ON, bytes=321, packets=1, start=789, end=789, flags=1),
        mk(O.FLOW_RECORD, K3, ON, bytes=111, packets=1, start=888, end=888, flags=1),
    ])
    consumed = acc.ingest(recs)
    assert consumed == 3 and len(acc) == 2          # the third key surpasses maxEntries: eviction of exactly 2
    ev = acc.evict()
    by_port = {int(r["id"]["src_port"]): r["metrics"] for r in ev}
    assert len(ev) == 2
    m1, m2 = by_port[333], by_port[12]
    assert (m1["bytes"], m1["packets"], m1["start"], m1["end"], m1["flags"]) == (444, 2, 123, 789, 1)
    assert (m2["bytes"], m2["packets"], m2["start"], m2["end"], m2["flags"]) == (456, 1, 456, 456, 1)
    # TimeFlowStart = now - (1000-123) ns, TimeFlowEnd = now - (1000-789) ns
    now = 1661272402 * 10**9
    s, e = C.c_int64(0), C.c_int64(0)
    one = np.ascontiguousarray(np.array([m1]))
    O.lib().orc_record_times(now, 1000, one.ctypes.data_as(C.c_void_p), C.byref(s), C.byref(e))
    assert s.value == now - (1000 - 123) and e.value == now - (1000 - 789)
    assert acc.ingest(recs[3:]) == 1 and len(acc) == 1   # k3 enters the fresh map; no further eviction


def test_evict_period_oracle(O):
    """account_test.go:130-217 TestEvict_Period: 3 records, tick, 2 records."""
    acc = O.Accounter(200)
    first = np.concatenate([mk(O.FLOW_RECORD, K1, ON, bytes=10, packets=1, start=t, end=t, flags=1) for t in (123, 456, 789)])
    second = np.concatenate([mk(O.FLOW_RECORD, K1, ON, bytes=10, packets=1, start=t, end=t, flags=1) for t in (1123, 1456)])
    assert acc.ingest(first) == 3
    ev = acc.evict()
    assert len(ev) == 1
    m = ev[0]["metrics"]
    assert (m["bytes"], m["packets"], m["start"], m["end"], m["flags"]) == (30, 3, 123, 789, 1)
    assert acc.ingest(second) == 2
    ev = acc.evict()
    m = ev[0]["metrics"]
    assert (m["bytes"], m["packets"], m["start"], m["end"], m["flags"]) == (20, 2, 1123, 1456, 1)
    assert len(acc) == 0


# ---- pkg/model/flow_content_test.go ----
def content(O, **base):
    c = np.zeros(1, dtype=O.CONTENT)
    for k, v in base.items():
        c["base"][k] = v
    return c


def part(dt, **kw):
    p = np.zeros(1, dtype=dt)
    for k, v in kw.items():
        p[k] = v
    return p


def call(O, fn, c, p):
    getattr(O.lib(), fn)(c.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p))


def base_tuple(c):
    b = c["base"][0]
    return int(b["start"]), int(b["end"]), int(b["packets"]), int(b["eth_protocol"])


def test_accumulate_dns(O):
    """flow_content_test.go:11-53"""
    c = content(O, start=10, end=20, packets=3)
    call(O, "orc_accumulate_dns", c, part(O.DNS, start=25, end=25, latency=1000, id=1, flags=0b11))
    assert base_tuple(c)[:3] == (10, 25, 3) and c["has_dns"][0] == 1
    d = c["dns"][0]
    assert (d["start"], d["end"], d["latency"], d["id"], d["flags"]) == (25, 25, 1000, 1, 0b11)
    call(O, "orc_accumulate_dns", c, part(O.DNS, start=30, end=30, latency=2000, id=1, flags=0b1001))
    d = c["dns"][0]
    assert base_tuple(c)[:3] == (10, 30, 3)
    assert (d["start"], d["end"], d["latency"], d["id"], d["flags"]) == (25, 25, 2000, 1, 0b1011)


def test_accumulate_pkt_drops(O):
    """flow_content_test.go:55-104"""
    c = content(O, start=10, end=20, packets=3)
    call(O, "orc_accumulate_drops", c, part(O.DROPS, start=25, end=25, bytes=5, packets=1, latest_drop_cause=100, latest_flags=0b11, latest_state=200))
    d = c["drops"][0]
    assert base_tuple(c)[:3] == (10, 25, 3)
    assert (d["bytes"], d["packets"], d["latest_drop_cause"], d["latest_flags"], d["latest_state"]) == (5, 1, 100, 0b11, 200)
    call(O, "orc_accumulate_drops", c, part(O.DROPS, start=30, end=30, bytes=10, packets=2, latest_drop_cause=101, latest_flags=0b1001, latest_state=201))
    d = c["drops"][0]
    assert base_tuple(c)[:3] == (10, 30, 3)
    assert (d["start"], d["end"], d["bytes"], d["packets"], d["latest_drop_cause"], d["latest_flags"], d["latest_state"]) == (25, 25, 15, 3, 101, 0b1011, 201)


def test_accumulate_net_events(O):
    """flow_content_test.go:106-151 (ring index, metadata de-duplication)"""
    c = content(O, start=10, end=20, packets=3)
    p = part(O.NETEV, start=25, end=25, network_events_idx=2)
    p["network_events"][0, 0, :2] = [1, 1]; p["network_events"][0, 1, :2] = [1, 2]
    p["bytes"][0, :2] = [20, 25]; p["packets"][0, :2] = [1, 2]
    call(O, "orc_accumulate_netev", c, p)
    assert base_tuple(c)[:3] == (10, 25, 3) and c["netev"][0]["network_events_idx"] == 2
    q = part(O.NETEV, start=30, end=30, network_events_idx=2)
    q["network_events"][0, 0, :2] = [1, 2]; q["network_events"][0, 1, :2] = [1, 3]
    q["bytes"][0, :2] = [11, 12]; q["packets"][0, :2] = [1, 1]
    call(O, "orc_accumulate_netev", c, q)
    e = c["netev"][0]
    assert base_tuple(c)[:3] == (10, 30, 3)
    assert e["start"] == 25 and e["end"] == 25 and e["network_events_idx"] == 3
    assert e["network_events"][:, :2].tolist() == [[1, 1], [1, 2], [1, 3], [0, 0]]
    assert e["bytes"].tolist() == [20, 25, 12, 0] and e["packets"].tolist() == [1, 2, 1, 0]


def test_accumulate_xlat(O):
    """flow_content_test.go:153-182"""
    c = content(O, start=10, end=20, packets=3)
    call(O, "orc_accumulate_xlat", c, part(O.XLAT, start=25, end=25))
    assert base_tuple(c)[:3] == (10, 25, 3) and c["xlat"][0]["start"] == 25
    call(O, "orc_accumulate_xlat", c, part(O.XLAT, start=30, end=30))
    assert base_tuple(c)[:3] == (10, 30, 3) and c["xlat"][0]["start"] == 25 and c["xlat"][0]["end"] == 25


def test_accumulate_additional(O):
    """flow_content_test.go:184-246 (RTT max, IPsec precedence)"""
    c = content(O, start=10, end=20, packets=3)
    call(O, "orc_accumulate_additional", c, part(O.ADDITIONAL, start=25, end=25, flow_rtt=200, ipsec_encrypted=1))
    a = c["additional"][0]
    assert base_tuple(c)[:3] == (10, 25, 3) and (a["flow_rtt"], a["ipsec_encrypted"], a["ipsec_ret"]) == (200, 1, 0)
    call(O, "orc_accumulate_additional", c, part(O.ADDITIONAL, start=30, end=30, flow_rtt=1000))
    a = c["additional"][0]
    assert base_tuple(c)[:3] == (10, 30, 3) and (a["start"], a["end"], a["flow_rtt"], a["ipsec_encrypted"]) == (25, 25, 1000, 1)
    call(O, "orc_accumulate_additional", c, part(O.ADDITIONAL, start=30, end=30, flow_rtt=800, ipsec_ret=5))
    a = c["additional"][0]
    assert (a["flow_rtt"], a["ipsec_ret"], a["ipsec_encrypted"]) == (1000, 5, 0)
    call(O, "orc_accumulate_additional", c, part(O.ADDITIONAL, start=30, end=30, flow_rtt=800))
    a = c["additional"][0]
    assert (a["flow_rtt"], a["ipsec_ret"], a["ipsec_encrypted"]) == (1000, 5, 0)


def test_accumulate_quic(O):
    """flow_content_test.go:248-336"""
    c = content(O, start=10, end=20, packets=3)
    call(O, "orc_accumulate_quic", c, part(O.QUIC, start=25, end=25, eth_protocol=3, version=1, seen_long_hdr=1))
    assert base_tuple(c) == (10, 25, 3, 3)
    call(O, "orc_accumulate_quic", c, part(O.QUIC, start=30, end=30, eth_protocol=3, version=2, seen_short_hdr=1))
    q = c["quic"][0]
    assert base_tuple(c) == (10, 30, 3, 3)
    assert (q["start"], q["end"], q["version"], q["seen_long_hdr"], q["seen_short_hdr"]) == (25, 25, 2, 1, 1)
    # DoesNotDecrease (:312-336)
    c = content(O, start=10, end=20, packets=3, eth_protocol=2048)
    call(O, "orc_accumulate_quic", c, part(O.QUIC, start=25, end=25, eth_protocol=2048, version=2, seen_long_hdr=1, seen_short_hdr=1))
    call(O, "orc_accumulate_quic", c, part(O.QUIC, start=30, end=30, eth_protocol=2048, version=1))
    q = c["quic"][0]
    assert (q["version"], q["seen_long_hdr"], q["seen_short_hdr"]) == (2, 1, 1)


@pytest.mark.parametrize("fn,dt", [("orc_accumulate_dns", "DNS"), ("orc_accumulate_drops", "DROPS"), ("orc_accumulate_netev", "NETEV"),
                                   ("orc_accumulate_xlat", "XLAT"), ("orc_accumulate_additional", "ADDITIONAL")])
def test_accumulate_now_base(O, fn, dt):
    """flow_content_test.go:338-380: base times come from the first partial when unset"""
    c = content(O)
    call(O, fn, c, part(getattr(O, dt), start=25, end=25))
    assert base_tuple(c) == (25, 25, 0, 0)


def test_accumulate_now_base_quic(O):
    c = content(O)
    call(O, "orc_accumulate_quic", c, part(O.QUIC, start=25, end=25, eth_protocol=3))
    assert base_tuple(c) == (25, 25, 0, 3)


def test_add_uint16_saturates(O):
    """flow_content.go:209-215"""
    assert O.lib().orc_add_uint16(65000, 1000) == 0xFFFF and O.lib().orc_add_uint16(5, 6) == 11


# ---- AccumulateBase order-dependent fields: PARITY UNPINNED (no reference test); follows
# the source text of flow_content.go:45-59 ----
def test_accumulate_base_order_dependent_fields(O):
    p = np.zeros(1, dtype=O.FLOW_METRICS); o = np.zeros(1, dtype=O.FLOW_METRICS)
    p["eth_protocol"], p["dscp"], p["sampling"] = 0x0800, 5, 7
    o["eth_protocol"], o["dscp"], o["sampling"] = 0, 0, 0
    o["src_mac"] = [1, 2, 3, 4, 5, 6]
    O.lib().orc_accumulate_base(p.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
    assert (p["eth_protocol"][0], p["dscp"][0], p["sampling"][0]) == (0x0800, 5, 7)      # zero never overwrites
    assert p["src_mac"][0].tolist() == [1, 2, 3, 4, 5, 6] and not p["dst_mac"][0].any()   # first non-zero MAC sticks
    o["eth_protocol"], o["dscp"], o["sampling"] = 0x86DD, 9, 11
    o["src_mac"] = [9, 9, 9, 9, 9, 9]
    O.lib().orc_accumulate_base(p.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
    assert (p["eth_protocol"][0], p["dscp"][0], p["sampling"][0]) == (0x86DD, 9, 11)      # last non-zero wins
    assert p["src_mac"][0].tolist() == [1, 2, 3, 4, 5, 6]
