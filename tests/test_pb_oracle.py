"""Record -> protobuf, CPU leg: the C oracle (oracle/nfagg_oracle_pb.c) against golden wire
bytes produced by the Python protobuf runtime over a mirror of proto/flow.proto
(tests/golden/gen_pb_golden.py), plus the reference's own conversion test
(pkg/exporter/kafka_proto_test.go:26-86) decoded with that runtime."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "pb_golden.json")))


def golden_inputs(O, g):
    recs = np.frombuffer(bytes.fromhex(g["records_hex"]), dtype=O.FLOW_RECORD)
    names = O.intf_table([(i, bytes.fromhex(m) if m is not None else None, n, u) for (i, m, n, u) in g["names"]])
    return recs, names


def test_oracle_matches_protobuf_runtime_bytes(O, golden):
    recs, names = golden_inputs(O, golden)
    for case in golden["cases"]:
        opts = O.pb_options(case["now_unix_ns"], case["mono_now_ns"], bytes.fromhex(case["agent_ip"]), names,
                            golden["unknown_name"].encode())
        got = O.pb_encode(recs, opts)
        want = [bytes.fromhex(h) for h in case["records_pb"]]
        for k, (g, w) in enumerate(zip(got, want)):
            assert g == w, f"record {k}: {g.hex()} != {w.hex()}"
        # framing of pbflow.Records{entries} (field 1, length-delimited) = concatenation of 0x0A len body
        framed = b"".join(b"\x0a" + _varint(len(b)) + b for b in got[:10])
        assert framed == bytes.fromhex(case["records10_message"])


def test_oracle_contents_match_protobuf_runtime_bytes(O, golden):
    """The MapTracer branch: full BpfFlowContent (DNS, drops, xlat, RTT/IPsec, QUIC) per flow."""
    recs, names = golden_inputs(O, golden)
    contents = np.frombuffer(bytes.fromhex(golden["contents_hex"]), dtype=O.CONTENT)
    assert len(contents) == len(recs)
    seen = set()
    for case in golden["cases"]:
        opts = O.pb_options(case["now_unix_ns"], case["mono_now_ns"], bytes.fromhex(case["agent_ip"]), names,
                            golden["unknown_name"].encode())
        got = O.pb_encode_contents(recs["id"], contents, opts)
        want = [bytes.fromhex(h) for h in case["contents_pb"]]
        for k, (g, w) in enumerate(zip(got, want)):
            assert g == w, f"content {k}: {g.hex()} != {w.hex()}"
        seen.update(len(w) for w in want)
    assert max(seen) > 300          # the fixtures reach the long frames (all features + 7 interfaces)


# pkg/decode/decode_protobuf_test.go:288-395 TestDnsRawNameToDotted, every case (input -> expected); the
# 1000-byte case is cut to the 32 bytes the kernel struct holds
DNS_KATS = [
    (b"", b""), (b"\x00", b""), (b"\x03abc\x00", b"abc"), (b"\x03abc\x03def\x00", b"abc.def"),
    (b"\x03www\x07example\x03com\x00", b"www.example.com"), (b"\x03abc\xc0\x12\x00", b"abc"), (b"\xc0\x12", b""),
    (b"\x0aabc", b""), (b"\x00\x03abc\x00", b""), (b"\x03ab\x00\x03def\x00", b""), (b"\x01a\x01b\x01c\x00", b"a.b.c"),
    (b"\x0aabcdefghij\x00", b"abcdefghij"), (b"\x03AbC\x03DeF\x00", b"AbC.DeF"), (b"\x05test1\x03abc\x00", b"test1.abc"),
    (b"\x03abc\x05de", b"abc"), (b"\x03abc\xc0\x12\xc0\x34\x00", b"abc"), (b"\x03abc\x00" + bytes(27), b"abc"),
]


@pytest.mark.parametrize("raw,want", DNS_KATS)
def test_dns_raw_name_to_dotted_kats(O, raw, want):
    import gen_pb_golden as G
    assert O.dns_name_dotted(raw) == want
    assert G.dns_raw_name_to_dotted((raw + bytes(32))[:32]) == want


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def test_reference_proto_conversion_test(O):
    """pkg/exporter/kafka_proto_test.go:26-86 TestProtoConversion: same record, same assertions on the decoded
    message, plus the Kafka key (:84-85)."""
    import gen_pb_golden as G
    Record, _ = G.build_classes()
    r = np.zeros(1, dtype=O.FLOW_RECORD)
    r["id"]["src_ip"][0] = np.frombuffer(bytes(10) + b"\xff\xff" + bytes([192, 1, 2, 3]), dtype=np.uint8)
    r["id"]["dst_ip"][0] = np.frombuffer(bytes(10) + b"\xff\xff" + bytes([127, 3, 2, 1]), dtype=np.uint8)
    r["id"]["src_port"], r["id"]["dst_port"], r["id"]["icmp_type"], r["id"]["proto"] = 4321, 1234, 8, 210
    m = r["metrics"]
    m["direction_first_seen"], m["eth_protocol"], m["bytes"], m["packets"], m["flags"] = 1, 3, 789, 987, 1
    m["src_mac"][0] = np.frombuffer(bytes.fromhex("aabbccddeeff"), dtype=np.uint8)
    m["dst_mac"][0] = np.frombuffer(bytes.fromhex("112233445566"), dtype=np.uint8)
    # Interfaces veth0/0 and abcde/1 -> first interface + one observed interface
    m["if_index_first_seen"], m["nb_observed_intf"] = 10, 1
    m["observed_intf"][0][0], m["observed_direction"][0][0] = 11, 1
    m["direction_first_seen"] = 1
    now, mono = 1_700_000_005_000_000_000, 50_000_000_000
    m["start"], m["end"] = mono - 5_000_000_000, mono            # TimeFlowStart = now - 5 s, TimeFlowEnd = now
    names = O.intf_table([(10, None, "veth0", ""), (11, None, "abcde", "")])
    opts = O.pb_options(now, mono, bytes(10) + b"\xff\xff" + bytes([10, 0, 0, 1]), names)
    msg = Record.FromString(O.pb_encode(r, opts)[0])
    assert msg.eth_protocol == 3 and msg.direction == 1
    assert len(msg.dup_list) == 2
    assert (msg.dup_list[0].interface, msg.dup_list[0].direction) == ("veth0", 1)   # the first entry carries direction_first_seen
    assert (msg.dup_list[1].interface, msg.dup_list[1].direction) == ("abcde", 1)
    assert msg.data_link.src_mac == 0xaabbccddeeff and msg.data_link.dst_mac == 0x112233445566
    assert msg.network.src_addr.ipv4 == 0xC0010203 and msg.network.dst_addr.ipv4 == 0x7F030201
    assert (msg.transport.src_port, msg.transport.dst_port, msg.transport.protocol) == (4321, 1234, 210)
    assert msg.icmp_type == 8 and msg.bytes == 789 and msg.packets == 987 and msg.flags == 1
    assert msg.time_flow_start.seconds * 1000 + msg.time_flow_start.nanos // 10**6 == (now - 5 * 10**9) // 10**6
    assert msg.time_flow_end.seconds * 1000 + msg.time_flow_end.nanos // 10**6 == now // 10**6
    key = O.kafka_keys(r)[0].tobytes()
    assert key[:16] == r["id"]["dst_ip"][0].tobytes() and key[16:] == r["id"]["src_ip"][0].tobytes()   # 127.3.2.1 sorts first


def test_kafka_key_orders_the_two_addresses(O):
    recs = O.gen_stream(500, seed=3, n_keys=200)
    keys = O.kafka_keys(recs)
    for r, k in zip(recs, keys):
        a, b = r["id"]["src_ip"].tobytes(), r["id"]["dst_ip"].tobytes()
        assert k.tobytes() == (a + b if a <= b else b + a)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_matches_protobuf_runtime_on_random_contents(O, seed):
    """Beyond the committed golden vectors: random flows and feature parts, the C oracle against the Python protobuf
    runtime filled by the independent Python restatement of NewRecord/FlowToPB (tests/golden/gen_pb_golden.py)."""
    import gen_pb_golden as G
    Record, _ = G.build_classes()
    rng = np.random.default_rng(seed)
    n = 150
    recs = O.gen_stream(n, seed=seed, n_keys=60, variant=1)
    m = recs["metrics"]
    m["if_index_first_seen"] = rng.choice(np.array([0, 1, 2, 3, 4, 7, 9, 4321], dtype=np.uint32), n)
    m["observed_intf"] = rng.choice(np.array([0, 1, 2, 3, 4, 7, 9, 4321], dtype=np.uint32), (n, 6))
    m["eth_protocol"][rng.integers(0, 3, n) == 0] = 0x86DD
    m["start"][rng.integers(0, 5, n) == 0] = 0
    contents = G.gen_contents(O, rng, recs)
    for name in ("has_dns", "has_drops", "has_netev", "has_xlat", "has_additional", "has_quic"):
        contents[name] = rng.integers(0, 2, n)
    contents["dns"]["latency"] = rng.integers(0, 1 << 63, n, dtype=np.uint64) * rng.integers(0, 3, n).astype(np.uint64)
    contents["additional"]["flow_rtt"] = rng.integers(0, 1 << 40, n, dtype=np.uint64) * rng.integers(0, 2, n).astype(np.uint64)
    contents["additional"]["ipsec_ret"] = rng.integers(-(1 << 31), 1 << 31, n)
    names_rows = G.NAMES
    namer = G.namer_from(names_rows)
    now, mono = int(rng.integers(10**9, 2 * 10**18)), int(rng.integers(0, 10**15))
    agent = bytes(10) + b"\xff\xff" + bytes(rng.integers(0, 256, 4, dtype=np.uint8)) if seed % 2 else bytes(rng.integers(0, 256, 16, dtype=np.uint8))
    opts = O.pb_options(now, mono, agent, O.intf_table(names_rows))
    got_r = O.pb_encode(recs, opts)
    got_c = O.pb_encode_contents(recs["id"], contents, opts)
    for k in range(n):
        want_r = G.flow_to_pb(Record, recs[k], now, mono, agent, namer).SerializeToString(deterministic=True)
        want_c = G.flow_to_pb(Record, recs[k], now, mono, agent, namer, content=contents[k]).SerializeToString(deterministic=True)
        assert got_r[k] == want_r, f"record {k}"
        assert got_c[k] == want_c, f"content {k}"


def test_reference_identical_keys(O):
    """pkg/exporter/kafka_proto_test.go:88-122 TestIdenticalKeys: A->B and B->A of one conversation get the same Kafka key."""
    r = np.zeros(2, dtype=O.FLOW_RECORD)
    a = np.frombuffer(bytes(10) + b"\xff\xff" + bytes([192, 1, 2, 3]), dtype=np.uint8)
    b = np.frombuffer(bytes(10) + b"\xff\xff" + bytes([127, 3, 2, 1]), dtype=np.uint8)
    r["id"]["src_ip"][0], r["id"]["dst_ip"][0] = a, b
    r["id"]["src_ip"][1], r["id"]["dst_ip"][1] = b, a
    r["id"]["src_port"], r["id"]["dst_port"], r["id"]["icmp_type"], r["id"]["proto"] = 4321, 1234, 8, 210
    keys = O.kafka_keys(r)
    assert keys[0].tobytes() == keys[1].tobytes() == b.tobytes() + a.tobytes()
