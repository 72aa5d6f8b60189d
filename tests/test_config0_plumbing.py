"""BASELINE configs[0]: 10 000 synthetic flow_record_t over 1 000 5-tuples -> Accounter -> capacity limiter ->
direct-flp stdout (one JSON object per evicted flow). CPU leg = plumbing with the oracle standing in for the
flow table (no GPU); GPU leg = the same pipeline with libnfagg behind the Accounter, same JSON lines.
Host mirrors checked against the reference's own tests: pkg/flow/limiter_test.go:17-71,
pkg/exporter/direct_flp_test.go:17-62, pkg/decode/decode_protobuf_test.go:20-176 (the keys of the base metrics)."""
import io
import ipaddress
import json
import queue
import threading

import numpy as np
import pytest

N_RECORDS, N_KEYS = 10_000, 1_000
NOW, MONO = 1_700_000_000_000_000_000, 3_000_000
NAMER = lambda ifx, mac: {2: "eth0", 3: "eth1", 4: "br-ex", 5: "ovn-k8s-mp0"}.get(ifx, "unknown")


@pytest.fixture(autouse=True)
def _restore_globals(nf):
    """SetInterfaceNamer / SetGlobalIP are process globals, as in the reference (record.go:50-61)."""
    from netobserv_ebpf_agent_amd import accounter as A
    namer, ip = A._interface_namer, A._agent_ip
    yield
    nf.SetInterfaceNamer(namer); nf.SetGlobalIP(ip)


def _mods(nf):
    import importlib
    return importlib.import_module("netobserv_ebpf_agent_amd.pipeline")


def test_capacity_limiter_no_drop_and_drop(nf):
    """limiter_test.go:17-71 with limiterLen = 50."""
    P = _mods(nf)
    for sent, expect in ((49, 49), (52, 50)):
        inp, out = queue.Queue(), queue.Queue(maxsize=50)
        lim = P.CapacityLimiter(nf.NoOp())
        t = threading.Thread(target=lim.Limit, args=(inp, out))
        t.start()
        for i in range(sent):
            inp.put([nf.Record(ID=None, Metrics=None, TimeFlowStart=0, TimeFlowEnd=0, Interfaces=[nf.NewIntfDirUdn(str(i), 0, None)])])
        import time as _t
        while inp.qsize():
            _t.sleep(0.001)
        got = [out.get(timeout=5) for _ in range(expect)]
        assert [g[0].Interfaces[0].Interface for g in got] == [str(i) for i in range(expect)]
        assert out.empty() and lim.droppedFlows == sent - expect
        inp.put(nf.CLOSE); t.join(timeout=5)
        assert out.get(timeout=5) is nf.CLOSE


def test_direct_flp_stdout_zero_record(nf):
    """direct_flp_test.go:44-61: a record with zero metrics and AgentIP 10.9.8.7."""
    P = _mods(nf)
    rec = nf.Record(ID=np.zeros((), dtype=nf.FLOW_ID), Metrics=np.zeros((), dtype=nf.FLOW_METRICS), TimeFlowStart=0, TimeFlowEnd=0,
                    AgentIP=ipaddress.ip_address("10.9.8.7"))
    buf, q = io.StringIO(), queue.Queue()
    q.put([rec]); q.put(nf.CLOSE)
    P.DirectFLPStdout(buf).ExportFlows(q)
    captured = json.loads(buf.getvalue().splitlines()[0])
    assert captured["TimeReceived"] != 0 and captured["AgentIP"] == "10.9.8.7"


def test_record_to_map_base_keys(nf):
    """decode_protobuf_test.go:20-176 TestPBFlowToMap, the keys that come from BpfFlowMetrics and the id."""
    P = _mods(nf)
    k = np.zeros((), dtype=nf.FLOW_ID); m = np.zeros((), dtype=nf.FLOW_METRICS)
    k["src_ip"] = np.frombuffer(bytes(10) + b"\xff\xff\x01\x02\x03\x04", dtype=np.uint8)
    k["dst_ip"] = np.frombuffer(bytes(10) + b"\xff\xff\x05\x06\x07\x08", dtype=np.uint8)
    k["src_port"], k["dst_port"], k["transport_protocol"] = 23000, 443, 6
    m["eth_protocol"], m["bytes"], m["packets"], m["dscp"], m["flags"] = 2048, 456, 123, 64, 0x100
    m["src_mac"] = np.frombuffer(bytes.fromhex("010203040506"), dtype=np.uint8)
    m["dst_mac"] = np.frombuffer(bytes.fromhex("112233445566"), dtype=np.uint8)
    some = 1_700_000_123_456_789_012
    rec = nf.Record(ID=k, Metrics=m, TimeFlowStart=some, TimeFlowEnd=some, AgentIP=ipaddress.ip_address("10.9.8.7"), TimeFlowRtt=10_000_000,
                    Interfaces=[nf.IntfDirUdn("5e6e92caa1d51cf", 0), nf.IntfDirUdn("eth0", 1)])
    out = P.RecordToMap(rec)
    assert out.pop("TimeReceived") != 0
    assert out == {"IfDirections": [0, 1], "Bytes": 456, "SrcAddr": "1.2.3.4", "DstAddr": "5.6.7.8", "Dscp": 64,
                   "DstMac": "11:22:33:44:55:66", "SrcMac": "01:02:03:04:05:06", "SrcPort": 23000, "DstPort": 443, "Etype": 2048,
                   "Packets": 123, "Proto": 6, "TimeFlowStartMs": some // 10**6, "TimeFlowEndMs": some // 10**6,
                   "Interfaces": ["5e6e92caa1d51cf", "eth0"], "Udns": ["", ""], "AgentIP": "10.9.8.7", "Flags": 0x100,
                   "TimeFlowRttNs": 10_000_000}


def _lines_from_evicted(nf, P, evicted):
    nf.SetInterfaceNamer(NAMER); nf.SetGlobalIP(ipaddress.ip_address("10.1.2.3"))
    buf, q = io.StringIO(), queue.Queue()
    q.put([nf.NewRecord(r["id"], r["metrics"], NOW, MONO) for r in evicted]); q.put(nf.CLOSE)
    P.DirectFLPStdout(buf, time_received=1_700_000_000).ExportFlows(q)
    return sorted(buf.getvalue().splitlines())


def _config0_stream(O):
    return O.gen_stream(N_RECORDS, seed=1, n_keys=N_KEYS)                  # uniform keys, bench fixture metrics


def test_config0_plumbing_on_cpu(nf, O):
    P = _mods(nf)
    recs = _config0_stream(O)
    evicted = O.run_accounter(recs, 1 << 20)[0][1]
    lines = _lines_from_evicted(nf, P, evicted.view(nf.FLOW_RECORD))
    assert len(lines) == len(evicted) <= N_KEYS and len(lines) > 0.99 * N_KEYS
    objs = [json.loads(l) for l in lines]
    assert sum(o["Bytes"] for o in objs) == int(recs["metrics"]["bytes"].astype(object).sum())
    assert sum(o["Packets"] for o in objs) == int(recs["metrics"]["packets"].sum())
    assert set(objs[0]) == {"AgentIP", "Bytes", "DstAddr", "DstMac", "DstPort", "Dscp", "Etype", "Flags", "IfDirections", "Interfaces",
                            "Packets", "Proto", "SrcAddr", "SrcMac", "SrcPort", "TimeFlowEndMs", "TimeFlowStartMs", "TimeReceived", "Udns"}
    assert all(o["Interfaces"][0] in ("eth0", "eth1", "br-ex", "ovn-k8s-mp0") for o in objs)


@pytest.mark.gpu
def test_config0_pipeline_on_gpu_matches_cpu_plumbing(nf, O):
    """ring records -> Accounter.Account (libnfagg) -> CapacityLimiter.Limit -> DirectFLP stdout."""
    P = _mods(nf)
    recs = _config0_stream(O)
    want = _lines_from_evicted(nf, P, O.run_accounter(recs, 1 << 20)[0][1].view(nf.FLOW_RECORD))
    nf.SetInterfaceNamer(NAMER); nf.SetGlobalIP(ipaddress.ip_address("10.1.2.3"))
    acc = nf.NewAccounter(1 << 16, 3600.0, lambda: NOW, lambda: MONO)
    q_in, q_mid, q_out = queue.Queue(), queue.Queue(), queue.Queue(maxsize=50)
    buf = io.StringIO()
    threads = [threading.Thread(target=acc.Account, args=(q_in, q_mid)),
               threading.Thread(target=_forward_close, args=(q_mid, q_out, P.CapacityLimiter(nf.NoOp()))),
               threading.Thread(target=P.DirectFLPStdout(buf, time_received=1_700_000_000).ExportFlows, args=(q_out,))]
    for t in threads:
        t.start()
    for off in range(0, N_RECORDS, 1000):                                   # ten ring batches
        q_in.put(recs[off:off + 1000].view(nf.FLOW_RECORD))
    q_in.put(nf.CLOSE)
    for t in threads:
        t.join(timeout=60)
        assert not t.is_alive()
    acc.close()
    assert sorted(buf.getvalue().splitlines()) == want


def _forward_close(q_mid, q_out, limiter):
    """Account() returns after the closing eviction without closing its output (Go closes it via the graph):
    forward that one batch through the limiter, then close."""
    inner = queue.Queue()
    t = threading.Thread(target=limiter.Limit, args=(inner, q_out))
    t.start()
    inner.put(q_mid.get(timeout=60))
    from netobserv_ebpf_agent_amd import CLOSE
    inner.put(CLOSE)
    t.join(timeout=30)


# ------------------------------------------------------------------ the kernel-map branch
def _decoration_fixture(nf, now):
    """pkg/agent/agent_test.go:140-166: key1/key2 and their metrics."""
    ids = np.zeros(2, dtype=nf.FLOW_ID); m = np.zeros(2, dtype=nf.FLOW_METRICS)
    ids["src_port"], ids["dst_port"] = [123, 333], [456, 532]
    m["packets"], m["bytes"] = [3, 7], [44, 33]
    m["start_mono_time_ts"], m["end_mono_time_ts"] = [now + 1000, now], [now + 1_000_000_000, now + 2_000_000_000]
    m["if_index_first_seen"], m["direction_first_seen"], m["nb_observed_intf"] = [1, 4], [1, 0], [1, 2]
    m["observed_intf"][0][:1] = [3]; m["observed_direction"][0][:1] = [0]
    m["observed_intf"][1][:2] = [1, 99]; m["observed_direction"][1][:2] = [1, 1]
    return ids, m


AGENT_NAMER = lambda ifx, mac: {1: "eth0", 3: "foo", 4: "bar"}.get(ifx, "unknown")   # agent_test.go:200-204 + interfaces_listener.go:77


def _check_decoration(nf, exported):
    """agent_test.go:168-189."""
    assert len(exported) == 2
    for f in exported:
        assert str(f.AgentIP) == "192.168.1.13"
        names = [i.Interface for i in f.Interfaces]
        if int(f.ID["src_port"]) == 123:
            assert names == ["eth0", "foo"]
        elif int(f.ID["src_port"]) == 333:
            assert names == ["bar", "eth0", "unknown"]
        else:
            raise AssertionError("unexpected key")


def test_flows_agent_decoration_new_record(nf):
    """TestFlowsAgent_Decoration through the NewRecord mirror alone (no GPU)."""
    nf.SetInterfaceNamer(AGENT_NAMER); nf.SetGlobalIP(ipaddress.ip_address("192.168.1.13"))
    ids, m = _decoration_fixture(nf, 10**12)
    _check_decoration(nf, [nf.NewRecord(ids[i], m[i], NOW, 10**12) for i in range(2)])


@pytest.mark.gpu
def test_map_tracer_evict_flows_decoration(nf, O):
    """The same reference test through MapTracer.evictFlows with LookupAndDeleteMap on the GPU; key2 also carries per-CPU
    DNS and RTT partials, so DNSLatency / TimeFlowRtt (record.go:116-125) are checked against the oracle's fold."""
    P = _mods(nf)
    nf.SetInterfaceNamer(AGENT_NAMER); nf.SetGlobalIP(ipaddress.ip_address("192.168.1.13"))
    now = 10**12
    ids, m = _decoration_fixture(nf, now)
    n_cpu = 4
    dns = np.zeros((1, n_cpu), dtype=nf.DNS); add = np.zeros((1, n_cpu), dtype=nf.ADDITIONAL)
    dns["latency"][0] = [0, 5_000, 70_000, 0]; dns["id"][0] = [0, 77, 0, 0]
    add["flow_rtt"][0] = [10, 0, 999, 12]
    feats = {"dns": (ids[1:2], dns), "additional": (ids[1:2], add)}
    with nf.FlowTable(max_entries=64) as tab:
        tracer = P.NewMapTracer(P.GPUMapFetcher(tab, lambda: (ids, m, feats, n_cpu)), 5.0, 5.0, nf.NoOp(), clock=lambda: NOW, mono_clock=lambda: now)
        out = queue.Queue()
        assert tracer.evictFlows(out) == 2
    exported = out.get_nowait()
    _check_decoration(nf, exported)
    k2 = [f for f in exported if int(f.ID["src_port"]) == 333][0]
    assert (k2.DNSLatency, k2.TimeFlowRtt) == (70_000, 999) and int(k2.DNSMetrics["id"]) == 77
    k1 = [f for f in exported if int(f.ID["src_port"]) == 123][0]
    assert (k1.DNSLatency, k1.TimeFlowRtt, k1.DNSMetrics) == (0, 0, None)
    assert tracer.metrics.evicted_flows_total == {("hashmap", ""): 2}


def test_limit_batches_abi_equals_the_limiter_on_a_burst(nf):
    """nfagg_limit_batches (include/nfagg.h) = CapacityLimiter.Limit (pkg/flow/limiter.go:28-38) applied to the burst of evictions
    one nfagg_account call delivers, nothing draining meanwhile — the setting of limiter_test.go:17-71 (limiterLen = 50: 49 batches
    all pass, of 52 the first 50 pass) — checked against the host mirror of the limiter itself."""
    import queue
    P = nf.pipeline
    rng = np.random.default_rng(3)
    for n_batches, qlen, qcap in ((49, 0, 50), (52, 0, 50), (10, 45, 50), (7, 50, 50), (5, 0, 0), (0, 3, 50), (30, 0, 1)):
        sizes = rng.integers(0, 5000, n_batches)
        ends = np.cumsum(sizes).tolist()
        keep, dropped = P.limit_batches(ends, qlen, qcap)
        # the mirror of the reference limiter on a queue that already holds qlen batches
        inp, out = queue.Queue(), queue.Queue(maxsize=qcap)
        for _ in range(qlen):
            out.put([None])
        m = nf.Metrics() if hasattr(nf, "Metrics") else None
        lim = P.CapacityLimiter(m)
        for s in sizes:
            inp.put([None] * int(s))
        inp.put(P.CLOSE)
        if qcap == 0:                      # an unbuffered channel never drops (the mirror would block on put: take the rule itself)
            assert keep == [True] * n_batches and dropped == 0
            continue
        want_keep = []
        for s in sizes:                    # limiter.go:30 step by step (Limit() itself needs a consumer for CLOSE)
            if out.qsize() < out.maxsize:
                out.put([None] * int(s)); want_keep.append(True)
            else:
                lim.droppedFlows += int(s); want_keep.append(False)
        assert keep == want_keep, (n_batches, qlen, qcap)
        assert dropped == lim.droppedFlows == int(sizes[[not k for k in want_keep]].sum()) if n_batches else dropped == 0
    keep, dropped = P.limit_batches(list(range(1, 53)), 0, 50)       # limiter_test.go:44-71: 52 one-record batches, 50 pass
    assert keep == [True] * 50 + [False] * 2 and dropped == 2
