"""Host-side mirror of the reference's Accounter for callers above the C ABI.

The reference's toolchain (Go) is absent from the build image, so the thin
host logic a Go `GPUAccounter` would carry (INTEGRATION.md shows the cgo form)
is mirrored here in Python with the same names, argument meaning and
behaviour, so that the parity tests read like pkg/flow/account_test.go:

  NewAccounter(maxEntries, evictTimeout, clock, monoClock, metrics)   account.go:34-53
  Accounter.Account(in, out)                                          account.go:58-100
  Accounter.evict(..., reason)                                        account.go:102-124
  NewRecord(key, metrics, currentTime, monotonicCurrentTime)          pkg/model/record.go:82-125

Channels are queue.Queue objects; closing the input channel is putting CLOSE.
Every flow-table operation goes through libnfagg (table.FlowTable): nothing
is accumulated in Python.
"""
import ipaddress
import queue
import threading
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from . import _lib as L
from .records import FLOW_RECORD
from .table import FlowTable

CLOSE = object()   # close(in): account.go:73-80


def _default_namer(if_index: int, _mac) -> str:          # record.go:52
    return f"[namer unset] {if_index}"


_interface_namer: Callable = _default_namer
_agent_ip = None


def SetInterfaceNamer(namer: Callable):                   # record.go:59-61
    global _interface_namer
    _interface_namer = namer


def SetGlobalIP(ip):                                      # record.go:55-57
    global _agent_ip
    _agent_ip = ip


@dataclass
class IntfDirUdn:                                         # record.go:161-165
    Interface: str
    Direction: int
    Udn: str = ""


def NewIntfDirUdn(intf: str, direction: int, cache: Optional[dict]) -> IntfDirUdn:   # record.go:167-187
    if not cache:
        return IntfDirUdn(intf, direction, "")
    udn = ""
    if intf in cache:
        udn = cache[intf] if cache[intf] != "" else "default"
    return IntfDirUdn(intf, direction, udn)


@dataclass
class Record:                                             # record.go:66-80
    ID: np.void
    Metrics: np.void                                      # BpfFlowContent.BpfFlowMetrics (base part)
    TimeFlowStart: int                                    # unix ns (time.Time in Go)
    TimeFlowEnd: int
    DNSLatency: int = 0
    Interfaces: List[IntfDirUdn] = field(default_factory=list)
    AgentIP: object = None
    TimeFlowRtt: int = 0
    DNSMetrics: object = None
    AdditionalMetrics: object = None

    def key(self) -> bytes:
        return bytes(np.asarray(self.ID).tobytes()[:39])


def _i64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def NewRecord(key, metrics, current_time_ns: int, monotonic_current_time: int, udns_cache=None,
              dns_metrics=None, additional_metrics=None) -> Record:
    """pkg/model/record.go:82-125 (network-event decoration, :126-157, needs the
    OVN sample decoder and stays with the Go host)."""
    start_delta = _i64(monotonic_current_time - int(metrics["start_mono_time_ts"]))     # :90
    end_delta = _i64(monotonic_current_time - int(metrics["end_mono_time_ts"]))         # :91
    rec = Record(ID=key, Metrics=metrics,
                 TimeFlowStart=_i64(current_time_ns - start_delta),                      # :96
                 TimeFlowEnd=_i64(current_time_ns - end_delta),                          # :97
                 AgentIP=_agent_ip)
    lmac = bytes(metrics["src_mac"])                                                     # :100-103
    if int(metrics["direction_first_seen"]) == 0:
        lmac = bytes(metrics["dst_mac"])
    rec.Interfaces = [NewIntfDirUdn(_interface_namer(int(metrics["if_index_first_seen"]), lmac),
                                    int(metrics["direction_first_seen"]), udns_cache)]  # :104-106
    for i in range(int(metrics["nb_observed_intf"])):                                    # :108-114
        rec.Interfaces.append(NewIntfDirUdn(_interface_namer(int(metrics["observed_intf"][i]), lmac),
                                            int(metrics["observed_direction"][i]), udns_cache))
    if dns_metrics is not None and int(dns_metrics["latency"]) != 0:                    # :116-120
        rec.DNSLatency = int(dns_metrics["latency"])
        rec.DNSMetrics = dns_metrics
    if additional_metrics is not None and int(additional_metrics["flow_rtt"]) != 0:     # :121-125
        rec.TimeFlowRtt = int(additional_metrics["flow_rtt"])
        rec.AdditionalMetrics = additional_metrics
    return rec


class Metrics:
    """The Prometheus series the Accounter touches (pkg/metrics/metrics.go:66-162),
    names and labels kept: evictions_total / evicted_flows_total
    {source="accounter",reason}, buffer_size{name="accounter-entries"}."""

    def __init__(self):
        self.evictions_total = {}
        self.evicted_flows_total = {}
        self.buffer_size = {}
        self.errors_total = {}

    def eviction(self, source, reason, flows):
        k = (source, reason)
        self.evictions_total[k] = self.evictions_total.get(k, 0) + 1
        self.evicted_flows_total[k] = self.evicted_flows_total.get(k, 0) + flows


def NoOp() -> Metrics:                                    # metrics.NoOp()
    return Metrics()


# The shim's batching policy (measured: profiles/r06_account_small_calls.txt). The reference hands Account ONE
# record per channel operation through a channel of BUFFERS_LENGTH = 50 (pkg/agent/agent.go:408, pkg/config/config.go:134,
# pkg/flow/tracer_ringbuf.go:112-134). An nfagg_account call costs ~80 us + ~0.011 us per record from a page-locked buffer; the
# reference's own loop ~0.09 us per record on one core: a call of fewer than ~1000 records costs more than the loop it replaces.
# So records are gathered until BATCH_RECORDS of them wait or the oldest has waited BATCH_TIMEOUT, whichever comes first. The
# timeout bounds what batching adds to a flow's way to the exporter: 1 ms against CACHE_ACTIVE_TIMEOUT = 5 s (config.go:142), and
# at most one 80 us call per millisecond when the node is nearly idle.
BATCH_RECORDS = 65536
BATCH_TIMEOUT = 0.001        # seconds


class Accounter:
    """pkg/flow/account.go:19-28. `entries` lives in HBM behind libnfagg."""

    def __init__(self, max_entries: int, evict_timeout: float, clock: Callable[[], int],
                 mono_clock: Callable[[], int], metrics: Optional[Metrics] = None,
                 batch_records: int = BATCH_RECORDS, batch_timeout: float = BATCH_TIMEOUT, **table_kw):
        self.maxEntries = max_entries
        self.evictTimeout = evict_timeout                 # seconds
        self.clock = clock                                # () -> unix ns
        self.monoClock = mono_clock                       # () -> monotonic ns
        self.metrics = metrics or NoOp()
        self.batchRecords = max(1, int(batch_records))
        self.batchTimeout = max(0.0, float(batch_timeout))   # 0: every item is its own call (round 5's behaviour)
        self.calls = 0                                    # nfagg_account calls made (the tests of the policy look at it)
        self.table = FlowTable(max_entries=max_entries, **table_kw)

    # -- account.go:58-100
    def Account(self, inp: "queue.Queue", out: "queue.Queue"):
        next_tick = time.monotonic() + self.evictTimeout
        pending, pending_n, first_at = [], 0, 0.0         # records received and not yet handed to the device

        def flush():
            nonlocal pending, pending_n, next_tick
            if not pending:
                return
            records = pending[0] if len(pending) == 1 else np.concatenate(pending)
            pending, pending_n = [], 0
            if self.account_batch(records, out):           # a "full" eviction resets the ticker (:93)
                next_tick = time.monotonic() + self.evictTimeout
            self.metrics.buffer_size["accounter-entries"] = len(self.table)   # :98 (per batch, not per record)

        while True:
            # Go's select serves evictTick.C as soon as it is ready, whether or not `in` has records waiting (:61-71);
            # Queue.get(timeout=0) would keep returning queued items, so a due tick is served before the next item.
            deadline = next_tick if not pending else min(next_tick, first_at + self.batchTimeout)
            timeout = deadline - time.monotonic()
            item = None
            if timeout > 0:
                try:
                    item = inp.get(timeout=timeout)
                except queue.Empty:
                    pass
            if item is None:
                now = time.monotonic()
                # every record received BEFORE the tick is in the map when the tick evicts it, as in the reference (its record
                # arm ran when the record arrived): the pending batch goes first
                if pending and (now >= first_at + self.batchTimeout or now >= next_tick):
                    flush()
                if now >= next_tick:                       # case <-evictTick.C (:63-71)
                    next_tick = time.monotonic() + self.evictTimeout
                    if len(self.table) == 0:
                        continue
                    self.evict(out, "timeout")
                continue
            if item is CLOSE:                              # :73-80 (what was received is accounted, then evicted)
                flush()
                self.evict(out, "closing")
                return
            records = np.atleast_1d(np.asarray(item, dtype=FLOW_RECORD))
            if not pending:
                first_at = time.monotonic()
            pending.append(records)
            pending_n += len(records)
            if pending_n >= self.batchRecords or self.batchTimeout == 0.0:
                flush()

    def account_batch(self, records: np.ndarray, out) -> bool:
        """The record arm (:81-96) for a batch, in arrival order, WITH its evictions on full: nfagg_account evicts inline
        whenever a record's new key finds len(c.entries) >= c.maxEntries (:85-94) and goes on — one call per batch whatever
        CACHE_MAX_FLOWS is; every eviction is one put on `out`, as the reference's. Returns True if one happened."""
        evicted_full = False
        off = 0
        while off < len(records):
            self.calls += 1
            rc, consumed, epochs = self.table.account(records[off:])
            off += consumed
            for raw in epochs:
                self._send(out, raw, "full")
                evicted_full = True
        return evicted_full

    # -- account.go:102-124
    def evict(self, out, reason: str):
        now = self.clock()
        monotonic_now = self.monoClock() & ((1 << 64) - 1)
        code = {"timeout": L.REASON_TIMEOUT, "full": L.REASON_FULL, "closing": L.REASON_CLOSING}[reason]
        self._send(out, self.table.evict(code), reason, now, monotonic_now)

    def _send(self, out, raw, reason: str, now=None, monotonic_now=None):
        if now is None:
            now = self.clock()
            monotonic_now = self.monoClock() & ((1 << 64) - 1)
        records = [NewRecord(r["id"], r["metrics"], now, monotonic_now) for r in raw]   # :116-119
        self.metrics.eviction("accounter", reason, len(records))                         # :120-121
        out.put(records)

    def close(self):
        self.table.close()


def NewAccounter(max_entries, evict_timeout, clock, mono_clock, metrics=None, **kw) -> Accounter:
    return Accounter(max_entries, evict_timeout, clock, mono_clock, metrics, **kw)
