"""The export encode pinned to the reference's OWN vectors: every case of
pkg/exporter/converters_test.go:20-545 (TestConversions), pkg/decode/decode_protobuf_test.go:177-287
(TestRecordToMap_OptionalMetrics, TestPBFlowRoundTrip_OptionalFields) and kafka_proto_test.go's record is restated as the
raw flow (144-byte record + folded feature structs + interface namer) whose model.NewRecord is the test's model.Record,
encoded by the oracle (orc_pb_encode_content) and — on the GPU — by nfagg_encode_pb_content, parsed with the protobuf
runtime over a mirror of proto/flow.proto, turned into the FLP map by tests/ref_decode.py (the reference's decode
direction restated) and compared with the map the reference's test expects. ref_decode.py itself is pinned by
TestPBFlowToMap (decode_protobuf_test.go:21-175): a pbflow.Record built field by field as in that test must decode to the
map that test expects."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)

NOW_NS = 1_700_000_000_123_000_000          # someTime
MONO = 5_000_000_000
MS = NOW_NS // 10**6
V4 = lambda a, b, c, d: bytes(10) + b"\xff\xff" + bytes([a, b, c, d])
SRC4, DST4, AGENT = V4(6, 7, 8, 9), V4(10, 11, 12, 13), V4(10, 11, 12, 13)
SMAC, DMAC = bytes([4, 5, 6, 7, 8, 9]), bytes([10, 11, 12, 13, 14, 15])
DNS_NAME = bytes([3]) + b"www" + bytes([7]) + b"example" + bytes([3]) + b"com" + bytes([0])
BASE_EXPECT = {"IfDirections": [1], "DstMac": "0a:0b:0c:0d:0e:0f", "SrcMac": "04:05:06:07:08:09", "TimeFlowStartMs": MS, "TimeFlowEndMs": MS,
               "Interfaces": ["eth0"], "Udns": [""], "AgentIP": "10.11.12.13"}
TCP_ID = dict(src_ip=SRC4, dst_ip=DST4, src_port=23000, dst_port=443, proto=6)
V4_ADDRS = {"SrcAddr": "6.7.8.9", "DstAddr": "10.11.12.13"}

# (name, id fields, metrics fields, features {kind: {field: value}}, interfaces [(if_index, name, direction)], expected map)
CASES = [
    ("TCP record with TLS", TCP_ID,
     dict(eth_protocol=2048, src_mac=SMAC, dst_mac=DMAC, bytes=456, packets=123, flags=0x100, dscp=64, sampling=1, ssl_version=0x0304,
          tls_types=0x22, tls_cipher_suite=0x1302, tls_key_share=0x1d),
     {"dns": dict(errno=0), "additional": dict(ipsec_encrypted=1)}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, **V4_ADDRS, Bytes=456, Dscp=64, Etype=2048, Packets=123, Proto=6, SrcPort=23000, DstPort=443, Flags=0x100, Sampling=1,
          IPSecRetCode=0, IPSecStatus="success", TLSVersion="TLS 1.3", TLSCipherSuite="TLS_AES_256_GCM_SHA384", TLSGroup="X25519",
          TLSTypes=["ServerHello", "AppData"])),
    ("UDP record", dict(TCP_ID, proto=17),
     dict(eth_protocol=2048, src_mac=SMAC, dst_mac=DMAC, bytes=456, packets=123, dscp=64, sampling=2),
     {"quic": dict(version=1, seen_long_hdr=1, seen_short_hdr=1)}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, **V4_ADDRS, Bytes=456, Dscp=64, Etype=2048, Packets=123, Proto=17, Sampling=2, SrcPort=23000, DstPort=443,
          QuicVersion="QUIC v2", QuicSeenLongHdr=1, QuicSeenShortHdr=1)),
    ("ICMPv4 record", dict(src_ip=SRC4, dst_ip=DST4, proto=1, icmp_type=8, icmp_code=0),
     dict(eth_protocol=2048, src_mac=SMAC, dst_mac=DMAC, bytes=456, packets=123, dscp=64), {}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, **V4_ADDRS, Bytes=456, Dscp=64, Etype=2048, Packets=123, Proto=1, IcmpType=8, IcmpCode=0)),
    ("ICMPv6 record", dict(src_ip=bytes(range(1, 17)), dst_ip=bytes(range(11, 27)), proto=58, icmp_type=8, icmp_code=0),
     dict(eth_protocol=0x86dd, src_mac=SMAC, dst_mac=DMAC, bytes=456, packets=123, dscp=64), {}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, Bytes=456, SrcAddr="102:304:506:708:90a:b0c:d0e:f10", DstAddr="b0c:d0e:f10:1112:1314:1516:1718:191a", Dscp=64,
          Etype=0x86dd, Packets=123, Proto=58, IcmpType=8, IcmpCode=0)),
    ("ARP layer2", dict(src_ip=bytes(16), dst_ip=bytes(16), proto=0, icmp_type=8, icmp_code=0),
     dict(eth_protocol=2054, src_mac=SMAC, dst_mac=DMAC, bytes=500, packets=128), {}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, Bytes=500, Etype=2054, Packets=128)),
    ("L2 drops", dict(src_ip=bytes(16), dst_ip=bytes(16), proto=0),
     dict(eth_protocol=2054, src_mac=SMAC, dst_mac=DMAC, bytes=500, packets=128),
     {"drops": dict(packets=10, bytes=100, latest_flags=0x200, latest_state=0, latest_drop_cause=2)}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, Bytes=500, Etype=2054, Packets=128, PktDropBytes=100, PktDropPackets=10, PktDropLatestFlags=0x200,
          PktDropLatestState="TCP_INVALID_STATE", PktDropLatestDropCause="SKB_DROP_REASON_NOT_SPECIFIED")),
    ("TCP + drop + DNS + RTT record", TCP_ID,
     dict(eth_protocol=2048, src_mac=SMAC, dst_mac=DMAC, bytes=456, packets=123, flags=0x100, dscp=64, ssl_version=0x0200),
     {"dns": dict(latency=10_000_000, id=1, name=DNS_NAME, flags=0x8001, errno=0),
      "drops": dict(packets=10, bytes=100, latest_flags=0x200, latest_state=6, latest_drop_cause=5),
      # the test's Record carries TimeFlowRtt = 10 ms; NewRecord takes it from AdditionalMetrics.FlowRtt (record.go:121-125)
      "additional": dict(ipsec_encrypted=1, flow_rtt=10_000_000)}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, **V4_ADDRS, Bytes=456, Dscp=64, Etype=2048, Packets=123, Proto=6, SrcPort=23000, DstPort=443, Flags=0x100,
          PktDropBytes=100, PktDropPackets=10, PktDropLatestFlags=0x200, PktDropLatestState="TCP_CLOSE",
          PktDropLatestDropCause="SKB_DROP_REASON_TCP_CSUM", DnsLatencyMs=10, DnsId=1, DnsName="www.example.com", DnsFlags=0x8001,
          DnsFlagsResponseCode="FormErr", TimeFlowRttNs=10_000_000, IPSecRetCode=0, IPSecStatus="success", TLSVersion="0x0200")),
    ("Multiple interfaces record", TCP_ID,
     dict(eth_protocol=2048, src_mac=SMAC, dst_mac=DMAC, bytes=64, packets=1, flags=0x100, dscp=64, ssl_version=0x0303),
     {"dns": dict(errno=0), "additional": dict(ipsec_encrypted=1)}, [(7, "5e6e92caa1d51cf", 0), (2, "eth0", 1)],
     dict(BASE_EXPECT, **V4_ADDRS, IfDirections=[0, 1], Bytes=64, Dscp=64, Etype=2048, Packets=1, Proto=6, SrcPort=23000, DstPort=443,
          Flags=0x100, Interfaces=["5e6e92caa1d51cf", "eth0"], Udns=["", ""], IPSecRetCode=0, IPSecStatus="success", TLSVersion="TLS 1.2")),
    ("SSL Mismatch", TCP_ID,
     dict(eth_protocol=2048, src_mac=SMAC, dst_mac=DMAC, bytes=64, packets=1, flags=0x100, dscp=64, ssl_version=0x0303, misc_flags=1),
     {}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, **V4_ADDRS, Bytes=64, Dscp=64, Etype=2048, Packets=1, Proto=6, SrcPort=23000, DstPort=443, Flags=0x100,
          TLSVersion="~ TLS 1.2")),
    # decode_protobuf_test.go:177-238 TestRecordToMap_OptionalMetrics (someTime = 1700000000 s; MACs zero)
    ("optional metrics: without", dict(TCP_ID, proto=17), dict(eth_protocol=2048, bytes=456, packets=123), {}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, **V4_ADDRS, SrcMac="00:00:00:00:00:00", DstMac="00:00:00:00:00:00", Bytes=456, Dscp=0, Etype=2048, Packets=123,
          Proto=17, SrcPort=23000, DstPort=443)),
    ("optional metrics: with", dict(TCP_ID, proto=17), dict(eth_protocol=2048, bytes=456, packets=123),
     {"quic": dict(version=1, seen_long_hdr=1, seen_short_hdr=1)}, [(2, "eth0", 1)],
     dict(BASE_EXPECT, **V4_ADDRS, SrcMac="00:00:00:00:00:00", DstMac="00:00:00:00:00:00", Bytes=456, Dscp=0, Etype=2048, Packets=123,
          Proto=17, SrcPort=23000, DstPort=443, QuicVersion="QUIC v2", QuicSeenLongHdr=1, QuicSeenShortHdr=1)),
]
KIND_BITS = {"additional": 1, "dns": 2, "drops": 4, "network_events": 8, "xlat": 16, "quic": 32}
M_ALIAS = {"errno": "errno_"}


def build(nf, O, case):
    _, idf, mf, feats, intfs, _ = case
    rec = np.zeros(1, dtype=O.FLOW_RECORD)
    rid, m = rec["id"], rec["metrics"]
    rid["src_ip"][0] = np.frombuffer(idf["src_ip"], np.uint8); rid["dst_ip"][0] = np.frombuffer(idf["dst_ip"], np.uint8)
    rid["src_port"], rid["dst_port"], rid["proto"] = idf.get("src_port", 0), idf.get("dst_port", 0), idf["proto"]
    rid["icmp_type"], rid["icmp_code"] = idf.get("icmp_type", 0), idf.get("icmp_code", 0)
    for k, v in mf.items():
        if k in ("src_mac", "dst_mac"):
            m[k][0] = np.frombuffer(v, np.uint8)
        else:
            m[k] = v
    m["start"], m["end"] = MONO, MONO                                     # TimeFlowStart = TimeFlowEnd = someTime (record.go:90-97)
    m["if_index_first_seen"], m["direction_first_seen"] = intfs[0][0], intfs[0][2]        # record.go:100-106
    m["nb_observed_intf"] = len(intfs) - 1                               # :108-114
    for k, (ifx, _, direction) in enumerate(intfs[1:]):
        m["observed_intf"][0][k], m["observed_direction"][0][k] = ifx, direction
    names = [(ifx, None, name, "") for ifx, name, _ in intfs]
    present = 0
    parts = {k: np.zeros(1, dtype=nf.ROLLUP_KINDS[k]) for k in ("additional", "dns", "drops", "xlat", "quic")}
    for kind, fields in feats.items():
        present |= KIND_BITS[kind]
        for k, v in fields.items():
            k = M_ALIAS.get(k, k)
            k = {"ipsec_ret": "ipsec_encrypted_ret"}.get(k, k)
            if k == "name":
                parts[kind][k][0][: len(v)] = np.frombuffer(v, np.uint8)
            else:
                parts[kind][k] = v
    return rec, np.array([present], dtype=np.uint8), parts, names


def oracle_content(O, present, parts):
    c = np.zeros(1, dtype=O.CONTENT)
    for kind, flag, field in (("dns", "has_dns", "dns"), ("drops", "has_drops", "drops"), ("xlat", "has_xlat", "xlat"),
                              ("additional", "has_additional", "additional"), ("quic", "has_quic", "quic")):
        if present[0] & KIND_BITS[kind]:
            c[flag] = 1
            c[field] = np.frombuffer(parts[kind].tobytes(), dtype=c.dtype[field])[0]      # same bytes, the oracle's own dtype
    return c


@pytest.fixture(scope="module")
def Record():
    import gen_pb_golden as G
    return G.build_classes()[0]


def check(case, body, Record):
    import ref_decode
    pb = Record.FromString(body)
    got = ref_decode.pb_to_map(pb)
    assert got == case[5], f"{case[0]}:\n got {got}\nwant {case[5]}"
    quic = any(k == "quic" for k in case[3])
    assert pb.HasField("quic") == quic                                   # TestPBFlowRoundTrip_OptionalFields: pb.Quic nil / non-nil


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_encode_decodes_to_the_reference_maps(nf, O, Record, case):
    rec, present, parts, names = build(nf, O, case)
    content = oracle_content(O, present, parts)
    content["base"] = rec["metrics"]
    opts = O.pb_options(NOW_NS, MONO, AGENT, O.intf_table(names))
    body = O.pb_encode_contents(rec["id"], content, opts)[0]
    check(case, body, Record)


@pytest.mark.gpu
def test_gpu_encode_decodes_to_the_reference_maps(nf, O, Record):
    from test_pb_gpu import frames
    with nf.FlowTable(max_entries=64) as tab:
        for case in CASES:
            rec, present, parts, names = build(nf, O, case)
            buf, off, blen = tab.encode_pb(rec.view(nf.FLOW_RECORD), NOW_NS, MONO, AGENT, nf.intf_table(names), present=present, parts=parts)
            check(case, frames(buf, off, blen)[0], Record)
            if not case[3]:          # no feature parts: the Accounter-branch encoder must produce the same frame
                buf2, off2, blen2 = tab.encode_pb(rec.view(nf.FLOW_RECORD), NOW_NS, MONO, AGENT, nf.intf_table(names))
                assert frames(buf2, off2, blen2)[0] == frames(buf, off, blen)[0]


def test_ref_decode_is_pinned_by_TestPBFlowToMap(Record):
    """decode_protobuf_test.go:21-175: the pbflow.Record of that test, built field by field, through tests/ref_decode.py ==
    the map that test expects (NetworkEvents aside: field 27 is not part of the mirrored descriptor)."""
    import ref_decode
    pb = Record()
    for name, direction in (("5e6e92caa1d51cf", 0), ("eth0", 1)):
        e = pb.dup_list.add(); e.interface, e.direction = name, direction
    pb.eth_protocol, pb.bytes, pb.packets = 2048, 456, 123
    pb.time_flow_start.seconds, pb.time_flow_start.nanos = NOW_NS // 10**9, NOW_NS % 10**9
    pb.time_flow_end.CopyFrom(pb.time_flow_start)
    pb.network.src_addr.ipv4, pb.network.dst_addr.ipv4, pb.network.dscp = 0x01020304, 0x05060708, 64
    pb.data_link.dst_mac, pb.data_link.src_mac = 0x112233445566, 0x010203040506
    pb.transport.protocol, pb.transport.src_port, pb.transport.dst_port = 6, 23000, 443
    pb.agent_ip.ipv4 = 0x0a090807
    pb.flags, pb.pkt_drop_bytes, pb.pkt_drop_packets, pb.pkt_drop_latest_flags = 0x100, 200, 20, 0x100
    pb.pkt_drop_latest_state, pb.pkt_drop_latest_drop_cause = 1, 4
    pb.dns_latency.nanos = 10_000_000
    pb.dns_id, pb.dns_name, pb.dns_flags, pb.dns_errno = 1, "www.example.com", 0x80, 0
    pb.time_flow_rtt.nanos = 10_000_000
    pb.xlat.src_addr.ipv4, pb.xlat.dst_addr.ipv4, pb.xlat.src_port, pb.xlat.dst_port, pb.xlat.zone_id = 0x01020304, 0x05060708, 1, 2, 100
    pb.ipsec_encrypted, pb.ipsec_encrypted_ret, pb.ssl_version = 1, 0, 0x0303
    pb.quic.version, pb.quic.seen_long_hdr, pb.quic.seen_short_hdr = 1, 1, 1
    want = {"IfDirections": [0, 1], "Bytes": 456, "SrcAddr": "1.2.3.4", "DstAddr": "5.6.7.8", "Dscp": 64, "DstMac": "11:22:33:44:55:66",
            "SrcMac": "01:02:03:04:05:06", "SrcPort": 23000, "DstPort": 443, "Etype": 2048, "Packets": 123, "Proto": 6, "TimeFlowStartMs": MS,
            "TimeFlowEndMs": MS, "Interfaces": ["5e6e92caa1d51cf", "eth0"], "Udns": ["", ""], "AgentIP": "10.9.8.7", "Flags": 0x100,
            "PktDropBytes": 200, "PktDropPackets": 20, "PktDropLatestFlags": 0x100, "PktDropLatestState": "TCP_ESTABLISHED",
            "PktDropLatestDropCause": "SKB_DROP_REASON_PKT_TOO_SMALL", "DnsLatencyMs": 10, "DnsId": 1, "DnsName": "www.example.com",
            "DnsFlags": 0x80, "DnsFlagsResponseCode": "NoError", "TimeFlowRttNs": 10_000_000, "XlatSrcAddr": "1.2.3.4", "XlatDstAddr": "5.6.7.8",
            "XlatSrcPort": 1, "XlatDstPort": 2, "ZoneId": 100, "IPSecRetCode": 0, "IPSecStatus": "success", "TLSVersion": "TLS 1.2",
            "QuicVersion": "QUIC v2", "QuicSeenLongHdr": 1, "QuicSeenShortHdr": 1}
    assert ref_decode.pb_to_map(Record.FromString(pb.SerializeToString())) == want
