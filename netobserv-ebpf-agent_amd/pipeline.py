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


def limit_batches(epoch_end, queue_len: int, queue_cap: int):
    """nfagg_limit_batches: CapacityLimiter.Limit's decision (limiter.go:28-38) for the evictions one nfagg_account call
    delivered, taken before any Record is built. Returns (keep flags, dropped flows)."""
    import ctypes as C
    import numpy as np
    from . import _lib as L
    ends = (C.c_uint64 * max(len(epoch_end), 1))(*[int(x) for x in epoch_end])
    keep = np.zeros(max(len(epoch_end), 1), dtype=np.uint8)
    dropped = C.c_uint64(0)
    L.lib.nfagg_limit_batches(ends, len(epoch_end), queue_len, queue_cap, keep.ctypes.data_as(C.c_void_p), C.byref(dropped))
    return [bool(k) for k in keep[: len(epoch_end)]], dropped.value


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


# ---------------------------------------------------------------------------------------------
# The kernel-map branch: MapTracer (pkg/flow/tracer_map.go:22-146) over a fetcher whose
# LookupAndDeleteMap runs on the GPU (nfagg_map_merge). The eBPF syscalls that drain the maps stay
# with the caller (`drain`); the join, the per-CPU folds and buildBaseFromAdditional are libnfagg's.
from dataclasses import dataclass  # noqa: E402
from typing import Callable, Optional  # noqa: E402

import numpy as np  # noqa: E402

from . import _lib as _L  # noqa: E402
from .accounter import NewRecord  # noqa: E402


@dataclass
class BpfFlowContent:                                     # pkg/model/flow_content.go:9-17 (nil = None)
    BpfFlowMetrics: np.void
    DNSMetrics: Optional[np.void] = None
    PktDropMetrics: Optional[np.void] = None
    NetworkEventsMetrics: Optional[np.void] = None
    XlatMetrics: Optional[np.void] = None
    AdditionalMetrics: Optional[np.void] = None
    QuicMetrics: Optional[np.void] = None


class GPUMapFetcher:
    """mapFetcher (tracer_map.go:37-40) whose LookupAndDeleteMap (pkg/tracer/tracer.go:1022-1116) is one
    nfagg_map_merge call. drain() -> (main_ids, main_vals, {kind: (ids, partials[n, n_cpu])}, n_cpu)."""

    _PARTS = (("dns", "DNSMetrics", _L.FEAT_DNS), ("drops", "PktDropMetrics", _L.FEAT_DROPS),
              ("network_events", "NetworkEventsMetrics", _L.FEAT_NETWORK_EVENTS), ("xlat", "XlatMetrics", _L.FEAT_XLAT),
              ("additional", "AdditionalMetrics", _L.FEAT_ADDITIONAL), ("quic", "QuicMetrics", _L.FEAT_QUIC))

    def __init__(self, table, drain: Callable):
        self.table, self.drain = table, drain

    def LookupAndDeleteMap(self, metrics=None):
        main_ids, main_vals, feats, n_cpu = self.drain()
        recs, present, parts, _dups = self.table.map_merge(main_ids, main_vals, feats, n_cpu)
        flows = []
        for i in range(len(recs)):
            c = BpfFlowContent(BpfFlowMetrics=recs[i]["metrics"])
            for kind, attr, bit in self._PARTS:
                if present[i] & bit:
                    setattr(c, attr, parts[kind][i])
            flows.append((recs[i]["id"], c))
        if metrics is not None:
            metrics.buffer_size["merged-maps"] = len(flows)            # tracer.go:1112
        return flows

    def DeleteMapsStaleEntries(self, timeout):                          # kernel-side housekeeping: stays in Go
        pass


class MapTracer:
    """tracer_map.go:22-60. TraceLoop's ticker / condition variable (:62-101) is goroutine plumbing: callers
    invoke evictFlows directly (what Flush() ends up doing)."""

    def __init__(self, fetcher, eviction_timeout, stale_entries_evict_timeout, metrics=None, s=None, udn_enabled=False,
                 clock=None, mono_clock=None):
        self.mapFetcher, self.evictionTimeout, self.staleEntriesEvictTimeout = fetcher, eviction_timeout, stale_entries_evict_timeout
        self.metrics, self.s, self.udnEnabled = metrics, s, udn_enabled
        self.clock = clock or (lambda: time.time_ns())
        self.monoClock = mono_clock or (lambda: time.monotonic_ns())

    def evictFlows(self, forwardFlows: "queue.Queue"):                  # :103-146
        monotonic_now, current = self.monoClock(), self.clock()
        flows = self.mapFetcher.LookupAndDeleteMap(self.metrics)
        udn_cache = dict(self.s.GetInterfaceUDNs()) if (self.s is not None and self.udnEnabled) else {}
        forwarding = [NewRecord(k, c.BpfFlowMetrics, current, monotonic_now, udn_cache,
                                dns_metrics=c.DNSMetrics, additional_metrics=c.AdditionalMetrics) for k, c in flows]
        self.mapFetcher.DeleteMapsStaleEntries(self.staleEntriesEvictTimeout)
        forwardFlows.put(forwarding)
        if self.metrics is not None:
            self.metrics.eviction("hashmap", "", len(forwarding))       # EvictionCounter / EvictedFlowsCounter WithSource("hashmap")
        return len(forwarding)


def NewMapTracer(fetcher, evictionTimeout, staleEntriesEvictTimeout, m=None, s=None, udnEnabled=False, **kw) -> MapTracer:
    return MapTracer(fetcher, evictionTimeout, staleEntriesEvictTimeout, m, s, udnEnabled, **kw)


def FlowsToPBMessages(buf, frame_offsets, max_len: int):
    """pbflow.FlowsToPB(records, maxLen) (pkg/pbflow/proto.go:18-36) over the output of nfagg_encode_pb: the serialized
    pbflow.Records messages GRPCProto.ExportFlows sends (pkg/exporter/grpc_proto.go:120), at most max_len entries each —
    byte ranges of `buf`, no copy of the frames' contents, no per-record allocation."""
    n = len(frame_offsets) - 1
    raw = memoryview(np.ascontiguousarray(buf))
    return [raw[int(frame_offsets[a]):int(frame_offsets[min(a + max_len, n)])] for a in range(0, n, max_len)]
