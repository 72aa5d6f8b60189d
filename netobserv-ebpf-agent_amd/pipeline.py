"""Host-side mirrors of the two pipeline nodes that sit right after the Accounter, for
BASELINE configs[0] ("10k flow_record_t, 1k 5-tuples, Accounter via direct-flp stdout"):

  CapacityLimiter.Limit(in, out)      pkg/flow/limiter.go:28-38 (drop when the destination buffer is full)
  RecordToMap(record)                 pkg/decode/decode_protobuf.go:63-127, the keys a BpfFlowMetrics-only
                                      record (what the Accounter evicts) produces
  DirectFLPStdout.ExportFlows(in)     pkg/exporter/direct_flp.go + flowlogs-pipeline write_stdout.go:37-51
                                      with `format: json` (one JSON object per flow, keys sorted)

Plumbing only — no flow state is touched here; the records come from libnfagg (accounter.py). The
string tables of the feature branch (TCP states, drop causes, DNS rcodes, TLS names) stay with the Go
decoder: RecordToMap refuses records that would need them instead of guessing."""
import ipaddress
import json
import queue
import sys
import time

from .accounter import CLOSE, Record


class CapacityLimiter:                                    # limiter.go:19-26
    def __init__(self, metrics=None):
        self.droppedFlows = 0
        self.metrics = metrics

    def Limit(self, inp: "queue.Queue", out: "queue.Queue"):
        """limiter.go:28-38. `out.maxsize` is cap(out); 0 = unbuffered, which never drops."""
        while True:
            batch = inp.get()
            if batch is CLOSE:
                out.put(CLOSE)
                return
            if out.maxsize == 0 or out.qsize() < out.maxsize:
                out.put(batch)
            else:
                if self.metrics is not None:
                    k = ("limiter", "full")
                    self.metrics.dropped_flows_total = getattr(self.metrics, "dropped_flows_total", {})
                    self.metrics.dropped_flows_total[k] = self.metrics.dropped_flows_total.get(k, 0) + len(batch)
                self.droppedFlows += len(batch)


def _mac(b) -> str:                                       # net.HardwareAddr.String()
    return ":".join("%02x" % x for x in bytes(b))


def _ip(b) -> str:                                        # model.IP(...).String(): net.IP of 16 bytes
    a = ipaddress.IPv6Address(bytes(b))
    return str(a.ipv4_mapped) if a.ipv4_mapped is not None else str(a)


def _unix_milli(ns: int) -> int:                          # time.Time.UnixMilli(): floor division
    return ns // 1_000_000


def RecordToMap(fr: Record, time_received: int = None) -> dict:
    """decode_protobuf.go:63-127 for a record that carries only BpfFlowMetrics."""
    m, k = fr.Metrics, fr.ID
    if int(m["ssl_version"]) or int(m["tls_types"]) or int(m["tls_cipher_suite"]) or int(m["tls_key_share"]):
        raise NotImplementedError("TLS name tables (crypto/tls) stay with the Go decoder")
    if fr.DNSMetrics is not None or fr.AdditionalMetrics is not None:
        raise NotImplementedError("feature keys stay with the Go decoder")
    out = {
        "SrcMac": _mac(m["src_mac"]), "DstMac": _mac(m["dst_mac"]), "Etype": int(m["eth_protocol"]),
        "TimeFlowStartMs": _unix_milli(fr.TimeFlowStart), "TimeFlowEndMs": _unix_milli(fr.TimeFlowEnd),
        "TimeReceived": int(time.time()) if time_received is None else time_received,
        "AgentIP": str(fr.AgentIP) if fr.AgentIP is not None else "<nil>",
        "IfDirections": [i.Direction for i in fr.Interfaces], "Interfaces": [i.Interface for i in fr.Interfaces],
    }
    if fr.Interfaces:
        out["Udns"] = [i.Udn for i in fr.Interfaces]
    if int(m["bytes"]):
        out["Bytes"] = int(m["bytes"])
    if int(m["packets"]):
        out["Packets"] = int(m["packets"])
    if int(m["sampling"]):
        out["Sampling"] = int(m["sampling"])
    if int(m["eth_protocol"]) in (0x0800, 0x86DD):
        out["SrcAddr"], out["DstAddr"] = _ip(k["src_ip"]), _ip(k["dst_ip"])
        proto = int(k["transport_protocol"])
        out["Proto"], out["Dscp"] = proto, int(m["dscp"])
        if proto in (1, 58):                              # IPPROTO_ICMP, IPPROTO_ICMPV6
            out["IcmpType"], out["IcmpCode"] = int(k["icmp_type"]), int(k["icmp_code"])
        elif proto in (6, 17, 132):                       # TCP, UDP, SCTP
            out["SrcPort"], out["DstPort"] = int(k["src_port"]), int(k["dst_port"])
            if proto == 6:
                out["Flags"] = int(m["flags"])
    if fr.TimeFlowRtt:
        out["TimeFlowRttNs"] = fr.TimeFlowRtt
    return out


class DirectFLPStdout:
    """StartDirectFLP with a lone `write: stdout, format: json` stage (direct_flp_test.go:17-33)."""

    def __init__(self, stream=None, time_received=None):
        self.stream = stream or sys.stdout
        self.time_received = time_received

    def ExportFlows(self, inp: "queue.Queue"):
        while True:
            batch = inp.get()
            if batch is CLOSE:
                return
            for rec in batch:
                self.stream.write(json.dumps(RecordToMap(rec, self.time_received), sort_keys=True, separators=(",", ":")) + "\n")
