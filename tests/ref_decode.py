"""Test infrastructure: the reference's DECODE direction restated, so that the bytes our encoders emit can be judged by
the maps the reference's own tests expect (pkg/exporter/converters_test.go TestConversions,
pkg/decode/decode_protobuf_test.go). pbflow.Record -> model.Record -> GenericMap:
  pb_to_map = decode.PBFlowToMap = RecordToMap(PBToFlow(pb))   (pkg/pbflow/proto.go:150-245, pkg/decode/decode_protobuf.go:56-205)
with the helpers they call (TCPStateToStr :209-236, PktDropCauseToStr :240-420, DNSRcodeToStr :426-464,
model.SSLVersionToString / TLSTypesToStrings / QuicVersionToString pkg/model/record.go:240-270, tls_types.go:8-25, Go's
crypto/tls VersionName / CipherSuiteName / CurveID.String). The restatement itself is pinned to the reference by
TestPBFlowToMap (decode_protobuf_test.go:21-175), see tests/test_export_reference_vectors.py."""
import ipaddress

TCP_STATES = {1: "TCP_ESTABLISHED", 2: "TCP_SYN_SENT", 3: "TCP_SYN_RECV", 4: "TCP_FIN_WAIT1", 5: "TCP_FIN_WAIT2", 6: "TCP_CLOSE",
              7: "TCP_CLOSE_WAIT", 8: "TCP_LAST_ACK", 9: "TCP_LISTEN", 10: "TCP_CLOSING", 11: "TCP_NEW_SYN_RECV"}
# decode_protobuf.go:232-262 (core subsystem, the first entries; everything the reference's vectors use)
DROP_CAUSES = {2: "SKB_DROP_REASON_NOT_SPECIFIED", 3: "SKB_DROP_REASON_NO_SOCKET", 4: "SKB_DROP_REASON_PKT_TOO_SMALL",
               5: "SKB_DROP_REASON_TCP_CSUM", 6: "SKB_DROP_REASON_SOCKET_FILTER", 7: "SKB_DROP_REASON_UDP_CSUM",
               8: "SKB_DROP_REASON_NETFILTER_DROP", 9: "SKB_DROP_REASON_OTHERHOST", 10: "SKB_DROP_REASON_IP_CSUM",
               11: "SKB_DROP_REASON_IP_INHDR", 12: "SKB_DROP_REASON_IP_RPFILTER", 13: "SKB_DROP_REASON_UNICAST_IN_L2_MULTICAST",
               14: "SKB_DROP_REASON_XFRM_POLICY", 15: "SKB_DROP_REASON_IP_NOPROTO"}
DNS_RCODES = {0: "NoError", 1: "FormErr", 2: "ServFail", 3: "NXDomain", 4: "NotImp", 5: "Refused", 6: "YXDomain", 7: "YXRRSet",
              8: "NXRRSet", 9: "NotAuth", 10: "NotZone", 16: "BADVERS", 17: "BADKEY", 18: "BADTIME", 19: "BADMODE", 20: "BADNAME", 21: "BADALG"}
TLS_TYPES = [(1, "ClientHello"), (2, "ServerHello"), (4, "OtherHandshake"), (8, "ChangeCipher"), (16, "Alert"), (32, "AppData")]
TLS_VERSIONS = {0x0300: "SSLv3", 0x0301: "TLS 1.0", 0x0302: "TLS 1.1", 0x0303: "TLS 1.2", 0x0304: "TLS 1.3"}
# crypto/tls cipher_suites.go (CipherSuites + InsecureCipherSuites)
CIPHER_SUITES = {0x0005: "TLS_RSA_WITH_RC4_128_SHA", 0x000a: "TLS_RSA_WITH_3DES_EDE_CBC_SHA", 0x002f: "TLS_RSA_WITH_AES_128_CBC_SHA",
                 0x0035: "TLS_RSA_WITH_AES_256_CBC_SHA", 0x003c: "TLS_RSA_WITH_AES_128_CBC_SHA256", 0x009c: "TLS_RSA_WITH_AES_128_GCM_SHA256",
                 0x009d: "TLS_RSA_WITH_AES_256_GCM_SHA384", 0xc007: "TLS_ECDHE_ECDSA_WITH_RC4_128_SHA",
                 0xc009: "TLS_ECDHE_ECDSA_WITH_AES_128_CBC_SHA", 0xc00a: "TLS_ECDHE_ECDSA_WITH_AES_256_CBC_SHA",
                 0xc011: "TLS_ECDHE_RSA_WITH_RC4_128_SHA", 0xc012: "TLS_ECDHE_RSA_WITH_3DES_EDE_CBC_SHA",
                 0xc013: "TLS_ECDHE_RSA_WITH_AES_128_CBC_SHA", 0xc014: "TLS_ECDHE_RSA_WITH_AES_256_CBC_SHA",
                 0xc023: "TLS_ECDHE_ECDSA_WITH_AES_128_CBC_SHA256", 0xc027: "TLS_ECDHE_RSA_WITH_AES_128_CBC_SHA256",
                 0xc02f: "TLS_ECDHE_RSA_WITH_AES_128_GCM_SHA256", 0xc02b: "TLS_ECDHE_ECDSA_WITH_AES_128_GCM_SHA256",
                 0xc030: "TLS_ECDHE_RSA_WITH_AES_256_GCM_SHA384", 0xc02c: "TLS_ECDHE_ECDSA_WITH_AES_256_GCM_SHA384",
                 0xcca8: "TLS_ECDHE_RSA_WITH_CHACHA20_POLY1305_SHA256", 0xcca9: "TLS_ECDHE_ECDSA_WITH_CHACHA20_POLY1305_SHA256",
                 0x1301: "TLS_AES_128_GCM_SHA256", 0x1302: "TLS_AES_256_GCM_SHA384", 0x1303: "TLS_CHACHA20_POLY1305_SHA256"}
CURVES = {23: "CurveP256", 24: "CurveP384", 25: "CurveP521", 29: "X25519", 4588: "X25519MLKEM768"}


def mac_str(v):                                    # macToUint8 + MacAddr.String
    return ":".join("%02x" % ((v >> s) & 0xff) for s in (40, 32, 24, 16, 8, 0))


def pb_ip16(ip):                                   # pbIPToNetIP + IPAddrFromNetIP: always the 16-byte form
    if ip.WhichOneof("ip_family") == "ipv6" and len(ip.ipv6):
        return bytes(ip.ipv6)
    n = ip.ipv4
    return bytes(10) + b"\xff\xff" + bytes([(n >> 24) & 0xff, (n >> 16) & 0xff, (n >> 8) & 0xff, n & 0xff])


def ip_str(b16):                                   # net.IP.String
    b16 = bytes(b16)
    if b16[:12] == bytes(10) + b"\xff\xff":
        return ".".join(str(x) for x in b16[12:])
    return str(ipaddress.IPv6Address(b16))


def string_to_int8_array(s):                       # pbflow/proto.go:286-330
    out, pos = bytearray(32), 0
    if s:
        for label in s.split("."):
            if pos >= 31:
                break
            n = min(len(label.encode()), 63)
            if n == 0:
                continue
            if pos + n + 1 > 31:
                break
            out[pos] = n
            out[pos + 1:pos + 1 + n] = label.encode()[:n]
            pos += 1 + n
    return bytes(out)


def dns_raw_name_to_dotted(raw):                   # pkg/utils/utils.go:18-58 (the restatement gen_pb_golden.py holds, pinned by its 17 KATs)
    import gen_pb_golden as G
    return G.dns_raw_name_to_dotted(raw).decode()


def pb_to_map(pb):
    eth = pb.eth_protocol & 0xffff
    start_ns = pb.time_flow_start.seconds * 10**9 + pb.time_flow_start.nanos
    end_ns = pb.time_flow_end.seconds * 10**9 + pb.time_flow_end.nanos
    out = {"SrcMac": mac_str(pb.data_link.src_mac), "DstMac": mac_str(pb.data_link.dst_mac), "Etype": eth,
           "TimeFlowStartMs": start_ns // 10**6, "TimeFlowEndMs": end_ns // 10**6, "AgentIP": ip_str(pb_ip16(pb.agent_ip))}
    out["IfDirections"] = [int(e.direction) for e in pb.dup_list]
    out["Interfaces"] = [e.interface for e in pb.dup_list]
    if len(pb.dup_list):
        out["Udns"] = [e.udn for e in pb.dup_list]
    if pb.bytes:
        out["Bytes"] = pb.bytes
    if pb.packets & 0xffffffff:
        out["Packets"] = pb.packets & 0xffffffff
    if pb.sampling:
        out["Sampling"] = pb.sampling
    ssl = pb.ssl_version & 0xffff
    if ssl:
        v = TLS_VERSIONS.get(ssl, "0x%04X" % ssl)
        out["TLSVersion"] = ("~ " + v) if pb.ssl_mismatch else v
    types = pb.tls_types & 0xff
    if types:
        out["TLSTypes"] = [n for b, n in TLS_TYPES if types & b]
    cs = pb.tls_cipher_suite & 0xffff
    if cs:
        out["TLSCipherSuite"] = CIPHER_SUITES.get(cs, "0x%04X" % cs)
    ks = pb.tls_key_share & 0xffff
    if ks:
        out["TLSGroup"] = CURVES.get(ks, "CurveID(%d)" % ks)
    proto = pb.transport.protocol & 0xff
    if eth in (0x0800, 0x86DD):
        out["SrcAddr"], out["DstAddr"] = ip_str(pb_ip16(pb.network.src_addr)), ip_str(pb_ip16(pb.network.dst_addr))
        out["Proto"], out["Dscp"] = proto, pb.network.dscp & 0xff
        if proto in (1, 58):
            out["IcmpType"], out["IcmpCode"] = pb.icmp_type & 0xff, pb.icmp_code & 0xff
        elif proto in (6, 17, 132):
            out["SrcPort"], out["DstPort"] = pb.transport.src_port & 0xffff, pb.transport.dst_port & 0xffff
            if proto == 6:
                out["Flags"] = pb.flags & 0xffff
    dns_lat_ns = pb.dns_latency.seconds * 10**9 + pb.dns_latency.nanos
    if pb.dns_errno & 0xff:
        out["DnsErrno"] = pb.dns_errno & 0xff
    if pb.dns_id & 0xffff:
        out["DnsId"], out["DnsFlags"] = pb.dns_id & 0xffff, pb.dns_flags & 0xffff
        out["DnsFlagsResponseCode"] = DNS_RCODES.get(pb.dns_flags & 0xF, "UnDefined")
        out["DnsLatencyMs"] = int(dns_lat_ns / 10**6)
        name = dns_raw_name_to_dotted(string_to_int8_array(pb.dns_name))
        if name:
            out["DnsName"] = name
    if pb.pkt_drop_latest_drop_cause:
        out["PktDropBytes"], out["PktDropPackets"] = pb.pkt_drop_bytes & 0xffff, pb.pkt_drop_packets & 0xffff
        out["PktDropLatestFlags"] = pb.pkt_drop_latest_flags & 0xffff
        out["PktDropLatestState"] = TCP_STATES.get(pb.pkt_drop_latest_state & 0xff, "TCP_INVALID_STATE")
        out["PktDropLatestDropCause"] = DROP_CAUSES[pb.pkt_drop_latest_drop_cause]
    xs, xd = pb_ip16(pb.xlat.src_addr), pb_ip16(pb.xlat.dst_addr)
    zero = lambda b: b[12:] == bytes(4) if b[:12] == bytes(10) + b"\xff\xff" else b == bytes(16)       # model.AllZeroIP
    if pb.HasField("xlat") and not zero(xs) and not zero(xd):
        out["ZoneId"] = pb.xlat.zone_id & 0xffff
        if pb.xlat.src_port & 0xffff:
            out["XlatSrcPort"] = pb.xlat.src_port & 0xffff
        if pb.xlat.dst_port & 0xffff:
            out["XlatDstPort"] = pb.xlat.dst_port & 0xffff
        out["XlatSrcAddr"], out["XlatDstAddr"] = ip_str(xs), ip_str(xd)
    if pb.ipsec_encrypted_ret:
        out["IPSecRetCode"], out["IPSecStatus"] = pb.ipsec_encrypted_ret, "error"
    elif pb.ipsec_encrypted:
        out["IPSecRetCode"], out["IPSecStatus"] = 0, "success"
    rtt_ns = pb.time_flow_rtt.seconds * 10**9 + pb.time_flow_rtt.nanos
    if rtt_ns:
        out["TimeFlowRttNs"] = rtt_ns
    if pb.HasField("quic"):
        v = pb.quic.version
        out["QuicVersion"] = {0: "QUIC v1", 1: "QUIC v2"}.get(v, "QUIC Unknown (%d)" % v)
        out["QuicSeenLongHdr"], out["QuicSeenShortHdr"] = pb.quic.seen_long_hdr & 0xff, pb.quic.seen_short_hdr & 0xff
    return out
