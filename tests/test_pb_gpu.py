"""Record -> protobuf on the GPU (csrc/nfagg_pb.hip) through the C ABI: bit-exact against the
golden wire bytes of the Python protobuf runtime (tests/golden/pb_golden.json) and against the
C oracle on seeded streams; framing, offsets, Kafka keys, truncation, device-resident path."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

# deliberately NOT sorted by index, with several rows per index: the answer must be that of a scan in table order
# (exact (index, MAC) row wherever it stands, else the FIRST any-MAC row of the index)
NAMES = [(8, None, "x" * 16, "u" * 63), (3, bytes.fromhex("020000000001"), "veth3a", "udn-blue"), (1, None, "lo", ""),
         (3, None, "veth3", ""), (2, None, "eth0", "default"), (3, None, "veth3-second-any", "never"),
         (3, bytes.fromhex("aabbccddeeff"), "veth3b", "udn-late"), (4, None, "ovn-k8s-mp0", "t"), (6, None, "", "nameless"),
         (2, bytes.fromhex("020000000002"), "eth0-mac", "")]
AGENT4 = bytes(10) + b"\xff\xff" + bytes([10, 1, 2, 3])


def frames(buf, off, blen):
    raw = buf.tobytes()
    out = []
    for i in range(len(blen)):
        fr = raw[int(off[i]):int(off[i + 1])]
        body = fr[len(fr) - int(blen[i]):]
        assert fr[0] == 0x0A and _varint(len(body)) == fr[1:len(fr) - len(body)]
        out.append(body)
    return out


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def test_golden_vectors(nf, O):
    g = json.load(open(os.path.join(HERE, "golden", "pb_golden.json")))
    recs = np.frombuffer(bytes.fromhex(g["records_hex"]), dtype=O.FLOW_RECORD)
    names = nf.intf_table([(i, bytes.fromhex(m) if m is not None else None, n, u) for (i, m, n, u) in g["names"]])
    with nf.FlowTable(max_entries=64) as tab:
        for case in g["cases"]:
            buf, off, blen = tab.encode_pb(recs.view(nf.FLOW_RECORD), case["now_unix_ns"], case["mono_now_ns"],
                                           bytes.fromhex(case["agent_ip"]), names, g["unknown_name"].encode())
            got = frames(buf, off, blen)
            want = [bytes.fromhex(h) for h in case["records_pb"]]
            assert len(got) == len(want)
            for k, (a, b) in enumerate(zip(got, want)):
                assert a == b, f"record {k}: {a.hex()} != {b.hex()}"
            # the first ten frames are exactly the pbflow.Records message of those ten entries
            assert buf[: int(off[10])].tobytes() == bytes.fromhex(case["records10_message"])


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 1025, 50_000])
def test_stream_parity_with_oracle(nf, O, n):
    recs = O.gen_stream(n, seed=n, n_keys=997, variant=1)     # scrambled: interfaces 1..8, observed lists, tls fields, zero times
    recs["metrics"]["eth_protocol"][::5] = 0x86DD
    names_rows = NAMES
    now, mono = 1_700_000_000_123_456_789, 2_500_000
    want = O.pb_encode(recs, O.pb_options(now, mono, AGENT4, O.intf_table(names_rows)))
    with nf.FlowTable(max_entries=64) as tab:
        buf, off, blen, keys = tab.encode_pb(recs.view(nf.FLOW_RECORD), now, mono, AGENT4, nf.intf_table(names_rows), kafka_keys=True)
    got = frames(buf, off, blen)
    assert got == want
    assert int(off[0]) == 0 and int(off[-1]) == len(buf) and (np.diff(off.astype(np.int64)) > 0).all()
    assert np.array_equal(keys, O.kafka_keys(recs))


def test_namer_table_larger_than_lds(nf, O):
    """More rows than the kernels stage in LDS (96): the lookups go to the table in HBM, same bytes."""
    rows = [(1000 + k, None, "if%d" % k, "u%d" % k) for k in range(150)] + NAMES
    recs = O.gen_stream(3000, seed=8, n_keys=500, variant=1)
    recs["metrics"]["if_index_first_seen"][::2] = 1000 + (np.arange(1500) % 150)
    recs["metrics"]["src_mac"][::3] = np.frombuffer(bytes.fromhex("aabbccddeeff"), dtype=np.uint8)
    recs["metrics"]["dst_mac"][::3] = np.frombuffer(bytes.fromhex("020000000002"), dtype=np.uint8)
    now, mono = 1_700_000_000_000_000_000, 10**9
    want = O.pb_encode(recs, O.pb_options(now, mono, AGENT4, O.intf_table(rows)))
    with nf.FlowTable(max_entries=64) as tab:
        buf, off, blen = tab.encode_pb(recs.view(nf.FLOW_RECORD), now, mono, AGENT4, nf.intf_table(rows))
    assert frames(buf, off, blen) == want


def test_empty_truncated_and_unknown_names(nf, O):
    import ctypes as C
    recs = O.gen_stream(300, seed=4, n_keys=50, variant=1)
    with nf.FlowTable(max_entries=64) as tab:
        buf, off, blen = tab.encode_pb(recs[:0].view(nf.FLOW_RECORD), 1, 1, AGENT4, nf.intf_table([]))
        assert len(buf) == 0 and list(off) == [0] and len(blen) == 0
        # no table at all: every interface is "unknown" (interfaces_listener.go:77)
        buf, off, blen = tab.encode_pb(recs.view(nf.FLOW_RECORD), 10**18, 10**9, bytes(range(16)), nf.intf_table([]), unknown=b"unknown")
        want = O.pb_encode(recs, O.pb_options(10**18, 10**9, bytes(range(16)), O.intf_table([])))
        assert frames(buf, off, blen) == want
        # too small an output buffer: nothing written, size reported
        o, keep = tab._pb_options(10**18, 10**9, bytes(range(16)), nf.intf_table([]), b"unknown")
        need = C.c_size_t(0)
        small = np.zeros(100, dtype=np.uint8)
        off2 = np.zeros(301, dtype=np.uint64); bl2 = np.zeros(300, dtype=np.uint32)
        rc = nf._lib.lib.nfagg_encode_pb(tab._h, recs.ctypes.data_as(C.c_void_p), 300, C.byref(o), small.ctypes.data_as(C.c_void_p), 100,
                                         off2.ctypes.data_as(C.c_void_p), bl2.ctypes.data_as(C.c_void_p), None, C.byref(need))
        assert rc == nf.TRUNCATED and need.value == len(buf) and not small.any()


def test_device_resident_evict_then_encode(nf, O):
    """nfagg_evict_device -> nfagg_encode_pb_device without leaving HBM: what the exporter hand-off would be."""
    import torch
    th = O.zipf_thresholds(3000, 1.1)
    recs = O.gen_stream(100_000, seed=12, n_keys=3000, thresholds=th, variant=1)
    want_flows = O.run_accounter(recs, 1 << 16)[0][1]
    now, mono = 1_700_000_000_000_000_000, 10**12
    names = NAMES
    with nf.FlowTable(max_entries=1 << 16) as tab:
        assert tab.ingest(recs.view(nf.FLOW_RECORD)) == (nf.OK, len(recs))
        d_ev = torch.empty(len(want_flows) * 144 + 16, dtype=torch.uint8, device="cuda")
        n = tab.evict_device(d_ev.data_ptr(), len(want_flows))
        assert n == len(want_flows)
        d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        d_len = torch.empty(n, dtype=torch.int32, device="cuda")
        d_keys = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        rc, need = tab.encode_pb_device(d_ev.data_ptr(), n, now, mono, AGENT4, nf.intf_table(names), 0, 0, d_off.data_ptr(), d_len.data_ptr())
        assert rc == nf.TRUNCATED and need > 0                   # size query
        d_out = torch.empty(need + 16, dtype=torch.uint8, device="cuda")
        rc, wrote = tab.encode_pb_device(d_ev.data_ptr(), n, now, mono, AGENT4, nf.intf_table(names), d_out.data_ptr(), need,
                                         d_off.data_ptr(), d_len.data_ptr(), d_keys.data_ptr())
        assert rc == nf.OK and wrote == need
        ev = d_ev[: n * 144].cpu().numpy().view(nf.FLOW_RECORD)
        got = frames(d_out[:need].cpu().numpy(), d_off.cpu().numpy().astype(np.uint64), d_len.cpu().numpy().astype(np.uint32))
    # eviction order is unspecified: encode the evicted records (as evicted) with the oracle
    assert got == O.pb_encode(ev.view(O.FLOW_RECORD), O.pb_options(now, mono, AGENT4, O.intf_table(names)))
    assert np.array_equal(d_keys.cpu().numpy().reshape(-1, 32), O.kafka_keys(ev.view(O.FLOW_RECORD)))


# ---- the MapTracer branch: full BpfFlowContent (nfagg_encode_pb_content)
def _split_contents(nf, O, contents):
    """oracle CONTENT array -> (present bits, parts dict) of the product ABI."""
    n = len(contents)
    present = np.zeros(n, dtype=np.uint8)
    for name, bit in (("has_additional", nf.FEAT_ADDITIONAL), ("has_dns", nf.FEAT_DNS), ("has_drops", nf.FEAT_DROPS),
                      ("has_netev", nf.FEAT_NETWORK_EVENTS), ("has_xlat", nf.FEAT_XLAT), ("has_quic", nf.FEAT_QUIC)):
        present |= (contents[name] != 0).astype(np.uint8) * np.uint8(bit)
    parts = {k: np.ascontiguousarray(contents[src]).view(np.uint8).reshape(n, -1).copy().view(nf.ROLLUP_KINDS[k]).reshape(n)
             for k, src in (("additional", "additional"), ("dns", "dns"), ("drops", "drops"), ("xlat", "xlat"), ("quic", "quic"))}
    return present, parts


def _records_of(nf, O, ids, contents):
    recs = np.zeros(len(ids), dtype=O.FLOW_RECORD)
    recs["id"], recs["metrics"] = ids, contents["base"]
    return recs.view(nf.FLOW_RECORD)


def test_content_golden_vectors(nf, O):
    g = json.load(open(os.path.join(HERE, "golden", "pb_golden.json")))
    recs = np.frombuffer(bytes.fromhex(g["records_hex"]), dtype=O.FLOW_RECORD)
    contents = np.frombuffer(bytes.fromhex(g["contents_hex"]), dtype=O.CONTENT)
    names = nf.intf_table([(i, bytes.fromhex(m) if m is not None else None, n, u) for (i, m, n, u) in g["names"]])
    present, parts = _split_contents(nf, O, contents)
    with nf.FlowTable(max_entries=64) as tab:
        for case in g["cases"]:
            buf, off, blen = tab.encode_pb(_records_of(nf, O, recs["id"], contents), case["now_unix_ns"], case["mono_now_ns"],
                                           bytes.fromhex(case["agent_ip"]), names, g["unknown_name"].encode(),
                                           present=present, parts=parts)
            got = frames(buf, off, blen)
            want = [bytes.fromhex(h) for h in case["contents_pb"]]
            for k, (a, b) in enumerate(zip(got, want)):
                assert a == b, f"content {k}: {a.hex()} != {b.hex()}"


def _random_contents(O, n, seed):
    rng = np.random.default_rng(seed)
    recs = O.gen_stream(n, seed=seed, n_keys=max(n // 3, 1), variant=1)
    recs["metrics"]["eth_protocol"][::3] = 0x86DD
    c = np.zeros(n, dtype=O.CONTENT)
    raw = c.view(np.uint8).reshape(n, -1)
    raw[:] = rng.integers(0, 256, raw.shape, dtype=np.uint8)            # every byte of every part random (padding too)
    c["base"] = recs["metrics"]
    for name in ("has_dns", "has_drops", "has_netev", "has_xlat", "has_additional", "has_quic"):
        c[name] = rng.integers(0, 2, n)
    # a third of the DNS names well-formed label sequences, a third NUL-terminated early, the rest raw noise
    for k in range(0, n, 3):
        labels = b"".join(bytes([l]) + bytes(rng.integers(97, 123, l, dtype=np.uint8)) for l in rng.integers(1, 9, 4))[:31]
        nm = np.zeros(32, dtype=np.uint8); nm[:len(labels)] = np.frombuffer(labels, dtype=np.uint8)
        c["dns"]["name"][k] = nm
    c["dns"]["name"][1::3, rng.integers(0, 32)] = 0
    zero = rng.integers(0, 4, n) == 0                                     # zero latencies / rtt: omitted or empty messages
    c["dns"]["latency"][zero] = 0
    c["additional"]["flow_rtt"][rng.integers(0, 4, n) == 0] = 0
    return recs, c


@pytest.mark.parametrize("n", [1, 64, 65, 1000, 1025, 40_000])
def test_content_stream_parity_with_oracle(nf, O, n):
    recs, contents = _random_contents(O, n, seed=100 + n)
    now, mono = 1_700_000_000_123_456_789, 2_500_000
    want = O.pb_encode_contents(recs["id"], contents, O.pb_options(now, mono, AGENT4, O.intf_table(NAMES)))
    present, parts = _split_contents(nf, O, contents)
    with nf.FlowTable(max_entries=64) as tab:
        buf, off, blen, keys = tab.encode_pb(_records_of(nf, O, recs["id"], contents), now, mono, AGENT4, nf.intf_table(NAMES),
                                             kafka_keys=True, present=present, parts=parts)
        got = frames(buf, off, blen)
        assert got == want
        assert np.array_equal(keys, O.kafka_keys(recs))
        assert max(len(b) for b in want) > (250 if n >= 64 else 0)
        # a part whose array is NULL is absent whatever `present` says; no parts at all = the Accounter encoding
        buf2, off2, blen2 = tab.encode_pb(_records_of(nf, O, recs["id"], contents), now, mono, AGENT4, nf.intf_table(NAMES),
                                          present=present, parts={"dns": parts["dns"]})
        only_dns = contents.copy()
        for name in ("has_drops", "has_netev", "has_xlat", "has_additional", "has_quic"):
            only_dns[name] = 0
        assert frames(buf2, off2, blen2) == O.pb_encode_contents(recs["id"], only_dns, O.pb_options(now, mono, AGENT4, O.intf_table(NAMES)))
        buf3, off3, blen3 = tab.encode_pb(_records_of(nf, O, recs["id"], contents), now, mono, AGENT4, nf.intf_table(NAMES),
                                          present=np.zeros(n, dtype=np.uint8), parts=parts)
        base = np.zeros(n, dtype=O.FLOW_RECORD); base["id"], base["metrics"] = recs["id"], contents["base"]
        assert frames(buf3, off3, blen3) == O.pb_encode(base, O.pb_options(now, mono, AGENT4, O.intf_table(NAMES)))


def test_content_rollup_then_encode_device(nf, O):
    """Per-CPU partials -> nfagg_rollup_* -> nfagg_encode_pb_content_device: the MapTracer hand-off with the folded
    parts resident in HBM."""
    import torch
    rng = np.random.default_rng(77)
    n, n_cpu = 5000, 8
    recs = O.gen_stream(n, seed=9, n_keys=n, variant=1)
    base = recs["metrics"].copy()
    parts_o, parts_p = {}, {}
    with nf.FlowTable(max_entries=64) as tab:
        b_o = base.copy(); b_p = base.copy().view(nf.FLOW_METRICS)
        for kind in ("dns", "drops", "xlat", "additional", "quic"):       # tracer.go:1057-1110 order (network events left out)
            dt = O.KIND_DTYPES[O.KIND_INDEX[kind]]
            partials = np.zeros((n, n_cpu), dtype=dt)
            partials.view(np.uint8).reshape(n, -1)[:] = rng.integers(0, 256, (n, n_cpu * dt.itemsize), dtype=np.uint8)
            b_o, parts_o[kind] = O.rollup(kind, partials, n_cpu, b_o)
            b_p, parts_p[kind] = tab.rollup(kind, partials.view(np.uint8).reshape(n, -1).copy().view(nf.ROLLUP_KINDS[kind]).reshape(n, n_cpu), n_cpu, b_p)
            assert parts_p[kind].tobytes() == parts_o[kind].tobytes()
        assert b_p.tobytes() == b_o.tobytes()
        present = rng.integers(0, 64, n).astype(np.uint8)
        contents = np.zeros(n, dtype=O.CONTENT)
        contents["base"] = b_o
        for kind, has in (("dns", "has_dns"), ("drops", "has_drops"), ("xlat", "has_xlat"), ("additional", "has_additional"), ("quic", "has_quic")):
            contents[kind] = parts_o[kind]
            contents[has] = (present >> O.KIND_INDEX[kind]) & 1
        now, mono = 1_712_345_678_000_000_001, 77_000_000_000
        want = O.pb_encode_contents(recs["id"], contents, O.pb_options(now, mono, AGENT4, O.intf_table(NAMES)))
        flows = np.zeros(n, dtype=O.FLOW_RECORD); flows["id"], flows["metrics"] = recs["id"], b_o
        d_recs = torch.from_numpy(flows.view(np.uint8).reshape(-1).copy()).cuda()
        d_present = torch.from_numpy(present).cuda()
        d_parts = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.uint8).reshape(-1).copy()).cuda() for k, v in parts_p.items()}
        d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        d_len = torch.empty(n, dtype=torch.int32, device="cuda")
        ptrs = {k: v.data_ptr() for k, v in d_parts.items()}
        rc, need = tab.encode_pb_device(d_recs.data_ptr(), n, now, mono, AGENT4, nf.intf_table(NAMES), 0, 0, d_off.data_ptr(), d_len.data_ptr(),
                                        d_present=d_present.data_ptr(), d_parts=ptrs)
        assert rc == nf.TRUNCATED and need == sum(len(b) + 1 + len(_varint(len(b))) for b in want)
        d_out = torch.empty(need + 16, dtype=torch.uint8, device="cuda")
        rc, wrote = tab.encode_pb_device(d_recs.data_ptr(), n, now, mono, AGENT4, nf.intf_table(NAMES), d_out.data_ptr(), need,
                                         d_off.data_ptr(), d_len.data_ptr(), d_present=d_present.data_ptr(), d_parts=ptrs)
        assert rc == nf.OK and wrote == need
        got = frames(d_out[:need].cpu().numpy(), d_off.cpu().numpy().astype(np.uint64), d_len.cpu().numpy().astype(np.uint32))
    assert got == want


def test_grpc_split_large_messages(nf, O):
    """pkg/exporter/grpc_proto_test.go:120-163 TestGRPCProto_SplitLargeMessages: 25 000 IPv6 flows, GRPC_MESSAGE_MAX_FLOWS =
    10 000 -> three pbflow.Records messages of 10 000, 10 000 and 5 000 entries; here the messages are byte ranges of the
    device-encoded buffer, parsed back with the protobuf runtime."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import gen_pb_golden as G
    _, Records = G.build_classes()
    recs = np.zeros(25000, dtype=nf.FLOW_RECORD)
    recs["metrics"]["eth_protocol"] = 0x86DD
    recs["id"]["src_port"] = np.arange(25000) % 60000                      # tell the entries apart
    agent = bytes.fromhex("11110000000000000000000000001111")             # 1111::1111
    names = nf.intf_table([(0, None, "12345678", "")])
    with nf.FlowTable(max_entries=64) as tab:
        buf, off, blen = tab.encode_pb(recs, 10**18, 10**9, agent, names)
    msgs = nf.FlowsToPBMessages(buf, off, 10_000)
    assert len(msgs) == 3
    seen = 0
    for m, want in zip(msgs, (10_000, 10_000, 5_000)):
        rs = Records.FromString(bytes(m))
        assert len(rs.entries) == want
        assert rs.entries[0].eth_protocol == 0x86DD and rs.entries[0].agent_ip.ipv6 == agent
        assert rs.entries[0].dup_list[0].interface == "12345678" and rs.entries[0].transport.src_port == seen % 60000
        seen += want
