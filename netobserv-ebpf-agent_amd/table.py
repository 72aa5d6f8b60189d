"""FlowTable — object wrapper over one libnfagg handle (include/nfagg.h).

All computation happens in the HIP library; this file only marshals numpy
buffers and raw device pointers across the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .records import FLOW_RECORD, ROLLUP_KINDS, FLOW_METRICS


class NfaggError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nfagg error {code}: {msg}")
        self.code = code


class PinnedRecords:
    """Room for n 144-byte records in page-locked host memory (nfagg_host_alloc): `.records` is a numpy view. Buffers like this
    are sent to / filled by the GPU by DMA directly (include/nfagg.h nfagg_host_alloc); pageable arrays take one more host copy."""

    def __init__(self, n: int):
        p = C.c_void_p()
        rc = L.lib.nfagg_host_alloc(max(int(n), 1) * 144, C.byref(p))
        if rc != L.OK or not p:
            raise NfaggError(rc, "nfagg_host_alloc failed")
        self._p = p
        self.records = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(int(n), 1) * 144,)).view(FLOW_RECORD)[:n]

    def close(self):
        if self._p:
            self.records = None
            L.lib.nfagg_host_free(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _HostPoolInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("workers", C.c_uint32), ("parts", C.c_uint32), ("bound", C.c_uint32),
                ("numa_node", C.c_int32), ("pad_", C.c_uint32), ("calibrated_gbs", C.c_double)]


def host_threads(threads: int = 0, numa_node: int = -1) -> int:
    """nfagg_host_threads: (re)shape the process's copy workers; returns the workers running."""
    rc = L.lib.nfagg_host_threads(threads, numa_node)
    if rc < 0:
        raise NfaggError(rc, "nfagg_host_threads")
    return rc


def host_info() -> dict:
    """nfagg_host_info: workers, the calibrated number of parts per copy, NUMA binding, the calibration's best rate."""
    info = _HostPoolInfo(struct_size=C.sizeof(_HostPoolInfo))
    rc = L.lib.nfagg_host_info(C.byref(info))
    if rc != L.OK:
        raise NfaggError(rc, "nfagg_host_info")
    return {"workers": info.workers, "parts": info.parts, "bound": bool(info.bound), "numa_node": info.numa_node,
            "calibrated_GBs": round(info.calibrated_gbs, 1)}


def device_numa_node(device: int = 0) -> int:
    return int(L.lib.nfagg_device_numa_node(device))


_ROLLUP_FN = {
    "additional": L.lib.nfagg_rollup_additional, "dns": L.lib.nfagg_rollup_dns, "drops": L.lib.nfagg_rollup_drops,
    "network_events": L.lib.nfagg_rollup_network_events, "xlat": L.lib.nfagg_rollup_xlat, "quic": L.lib.nfagg_rollup_quic,
}


class FlowTable:
    """The GPU-resident replacement of Accounter.entries (pkg/flow/account.go:22)."""

    def __init__(self, max_entries=5000, device=0, mode=L.MODE_ACCOUNTER, sketches=0, cm_depth=0, cm_log2_width=0,
                 hll_p=0, table_log2_slots=0, staging_records=0, n_shards=1, shard_id=0, profile=False,
                 ingest_variant=0, ext_sketch=None, copy_threads=0, local_fold=False):
        cfg = L.Config()
        cfg.struct_size = C.sizeof(L.Config)
        cfg.device = device
        cfg.max_entries = max_entries
        cfg.table_log2_slots = table_log2_slots
        cfg.mode = mode
        cfg.sketch_flags = sketches
        cfg.cm_depth, cfg.cm_log2_width, cfg.hll_p = cm_depth, cm_log2_width, hll_p
        cfg.staging_records = staging_records
        cfg.n_shards, cfg.shard_id = n_shards, shard_id
        cfg.profile = 1 if profile else 0
        cfg.ingest_variant = ingest_variant
        cfg.copy_threads = copy_threads
        cfg.local_fold = 1 if local_fold else 0       # a rank of a local-fold job (nfagg_partials_*); kernel-dedup mode: sub-flow table
        if ext_sketch:
            for k, p in enumerate(ext_sketch):
                cfg.ext_sketch[k] = p
        self._h = C.c_void_p()
        rc = L.lib.nfagg_create(C.byref(cfg), C.byref(self._h))
        if rc != L.OK:
            msg = L.lib.nfagg_last_error(None)
            self._h = None
            raise NfaggError(rc, msg.decode() if msg else "nfagg_create failed")
        self.max_entries = max_entries if max_entries else 5000

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            L.lib.nfagg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, ok=(L.OK,)):
        if rc not in ok:
            msg = L.lib.nfagg_last_error(self._h)
            raise NfaggError(rc, msg.decode() if msg else "")
        return rc

    # -- ingest
    def ingest(self, records: np.ndarray):
        """nfagg_ingest: fold host records in order. Returns (status, consumed)."""
        records = np.ascontiguousarray(records)
        assert records.dtype.itemsize == 144 or records.dtype == np.uint8
        n = records.nbytes // 144
        consumed = C.c_size_t(0)
        rc = L.lib.nfagg_ingest(self._h, records.ctypes.data_as(C.c_void_p), n, C.byref(consumed))
        self._check(rc, (L.OK, L.FULL))
        return rc, consumed.value

    def ingest_device(self, d_ptr: int, n: int):
        consumed = C.c_size_t(0)
        rc = L.lib.nfagg_ingest_device(self._h, C.c_void_p(d_ptr), n, C.byref(consumed))
        self._check(rc, (L.OK, L.FULL))
        return rc, consumed.value

    def staging_acquire(self):
        buf, cap = C.c_void_p(), C.c_size_t(0)
        self._check(L.lib.nfagg_staging_acquire(self._h, C.byref(buf), C.byref(cap)))
        arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(cap.value * 144,)).view(FLOW_RECORD)
        return arr

    def staging_commit(self, n):
        consumed = C.c_size_t(0)
        rc = L.lib.nfagg_staging_commit(self._h, n, C.byref(consumed))
        self._check(rc, (L.OK, L.FULL))
        return rc, consumed.value

    def __len__(self):
        v = C.c_uint64(0)
        self._check(L.lib.nfagg_len(self._h, C.byref(v)))
        return v.value

    # -- evict
    def evict(self, reason=L.REASON_TIMEOUT, cap=None, out=None) -> np.ndarray:
        """nfagg_evict: every live flow as one flow_record_t; table cleared. out: a FLOW_RECORD array to deliver into (a caller
        that evicts tick after tick reuses one, as the cgo shim does)."""
        if out is not None:
            cap = len(out)
        elif cap is None:
            cap = max(len(self), 1)
        if out is None:
            out = np.empty(cap, dtype=FLOW_RECORD)
        n = C.c_size_t(0)
        rc = L.lib.nfagg_evict(self._h, reason, out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        self._check(rc, (L.OK, L.TRUNCATED))
        if rc == L.TRUNCATED:
            return self.evict(reason, cap=n.value)
        return out[: n.value]

    # -- account: the record arm with its evictions on "full" (account.go:81-96) in one call
    def account(self, records: np.ndarray, out_cap=None, max_epochs=None, out=None):
        """nfagg_account. Returns (status, consumed, epochs): epochs = list of arrays (views of `out`), one per eviction on
        "full", in order. out: a FLOW_RECORD array to deliver into (a caller that accounts batch after batch reuses one)."""
        records = np.ascontiguousarray(records)
        n = records.nbytes // 144
        if out is not None:
            out_cap = len(out)
        if out_cap is None:
            out_cap = max(n + self.max_entries, self.max_entries)        # every record may start a flow; plus what is live already
        if max_epochs is None:
            max_epochs = n // max(self.max_entries, 1) + 2
        if out is None:
            out = np.empty(max(out_cap, 1), dtype=FLOW_RECORD)
        ends = (C.c_uint64 * max(max_epochs, 1))()
        n_ep, consumed = C.c_size_t(0), C.c_size_t(0)
        rc = L.lib.nfagg_account(self._h, records.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), out_cap, ends, max_epochs,
                                 C.byref(n_ep), C.byref(consumed))
        self._check(rc, (L.OK, L.TRUNCATED))
        epochs, lo = [], 0
        for e in range(n_ep.value):
            epochs.append(out[lo:int(ends[e])])
            lo = int(ends[e])
        return rc, consumed.value, epochs

    def account_device(self, d_ptr: int, n: int, d_out: int, out_cap: int, max_epochs: int):
        """nfagg_account_device. Returns (status, consumed, epoch_end list)."""
        ends = (C.c_uint64 * max(max_epochs, 1))()
        n_ep, consumed = C.c_size_t(0), C.c_size_t(0)
        rc = L.lib.nfagg_account_device(self._h, C.c_void_p(d_ptr), n, C.c_void_p(d_out or None), out_cap, ends, max_epochs, C.byref(n_ep), C.byref(consumed))
        self._check(rc, (L.OK, L.TRUNCATED))
        return rc, consumed.value, [int(ends[e]) for e in range(n_ep.value)]

    def evict_device(self, d_ptr: int, cap: int, reason=L.REASON_TIMEOUT) -> int:
        n = C.c_size_t(0)
        self._check(L.lib.nfagg_evict_device(self._h, reason, C.c_void_p(d_ptr), cap, C.byref(n)))
        return n.value

    # -- rollups (pkg/tracer/tracer.go:1057-1146)
    def rollup(self, kind: str, partials: np.ndarray, n_cpu: int, base: np.ndarray):
        dt = ROLLUP_KINDS[kind]
        partials = np.ascontiguousarray(partials, dtype=dt)
        n_flows = partials.size // n_cpu
        base = np.ascontiguousarray(base, dtype=FLOW_METRICS).copy()
        folded = np.zeros(n_flows, dtype=dt)
        self._check(_ROLLUP_FN[kind](self._h, partials.ctypes.data_as(C.c_void_p), n_flows, n_cpu,
                                     base.ctypes.data_as(C.c_void_p), folded.ctypes.data_as(C.c_void_p)))
        return base, folded

    # -- map merge (LookupAndDeleteMap's join), nfagg_map_merge
    _KIND_ORDER = ("additional", "dns", "drops", "network_events", "xlat", "quic")      # NFAGG_ROLLUP_*

    def map_merge(self, main_ids, main_vals, feats: dict, n_cpu: int, cap=None):
        """feats: {kind: (ids[n], partials[n, n_cpu])}. Returns (records, present, parts dict, n_duplicate_keys):
        one entry per merged flow, in order of first appearance."""
        mi = np.ascontiguousarray(main_ids, dtype=FLOW_RECORD["id"])
        mv = np.ascontiguousarray(main_vals, dtype=FLOW_METRICS)
        assert len(mi) == len(mv)
        keep = [mi, mv]
        main = L.MapView(mi.ctypes.data if len(mi) else None, mv.ctypes.data if len(mv) else None, len(mi))
        views = (L.MapView * 6)()
        total = len(mi)
        for k, name in enumerate(self._KIND_ORDER):
            if name not in feats:
                continue
            fi = np.ascontiguousarray(feats[name][0], dtype=FLOW_RECORD["id"])
            fv = np.ascontiguousarray(feats[name][1], dtype=ROLLUP_KINDS[name]).reshape(-1)
            assert fv.size == len(fi) * n_cpu, name
            keep += [fi, fv]
            views[k] = L.MapView(fi.ctypes.data if len(fi) else None, fv.ctypes.data if len(fi) else None, len(fi))
            total += len(fi)
        cap = total if cap is None else cap
        recs = np.zeros(max(cap, 1), dtype=FLOW_RECORD)
        present = np.zeros(max(cap, 1), dtype=np.uint8)
        parts = {name: np.zeros(max(cap, 1), dtype=ROLLUP_KINDS[name]) for name in self._KIND_ORDER}
        out = L.MergedFlows(recs.ctypes.data, present.ctypes.data, *[parts[name].ctypes.data for name in self._KIND_ORDER])
        n_out, n_dup = C.c_size_t(0), C.c_size_t(0)
        rc = L.lib.nfagg_map_merge(self._h, C.byref(main), views, n_cpu, C.byref(out), cap, C.byref(n_out), C.byref(n_dup))
        if rc == L.TRUNCATED:                      # caller's cap too small: retry with the size the library reported (as evict does)
            return self.map_merge(main_ids, main_vals, feats, n_cpu, cap=n_out.value)
        self._check(rc)
        n = n_out.value
        return recs[:n], present[:n], {k: v[:n] for k, v in parts.items()}, n_dup.value

    def map_merge_device(self, d_main, d_feats: dict, n_cpu: int, d_out: dict, cap: int):
        """Raw device pointers: d_main = (d_ids, d_vals, n); d_feats = {kind: (d_ids, d_vals, n)};
        d_out = {"records", "present", kind...: pointer}. Returns (rc, n_out, n_duplicate_keys)."""
        main = L.MapView(d_main[0] or None, d_main[1] or None, d_main[2])
        views = (L.MapView * 6)()
        for k, name in enumerate(self._KIND_ORDER):
            if name in d_feats:
                views[k] = L.MapView(d_feats[name][0] or None, d_feats[name][1] or None, d_feats[name][2])
        out = L.MergedFlows(d_out.get("records") or None, d_out.get("present") or None,
                            *[d_out.get(name) or None for name in self._KIND_ORDER])
        n_out, n_dup = C.c_size_t(0), C.c_size_t(0)
        rc = L.lib.nfagg_map_merge_device(self._h, C.byref(main), views, n_cpu, C.byref(out), cap, C.byref(n_out), C.byref(n_dup))
        self._check(rc, ok=(L.OK, L.TRUNCATED))
        return rc, n_out.value, n_dup.value

    # -- sketches
    def sketch_snapshot(self, which):
        p, b = C.c_void_p(), C.c_size_t(0)
        self._check(L.lib.nfagg_sketch_device_ptr(self._h, which, C.byref(p), C.byref(b)))
        if which in (L.CM_SRC, L.CM_DST):
            out = np.zeros(b.value // 8, dtype=np.uint64)
        else:
            out = np.zeros(b.value, dtype=np.uint8)
        self._check(L.lib.nfagg_sketch_snapshot(self._h, which, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def sketch_device_ptr(self, which):
        p, b = C.c_void_p(), C.c_size_t(0)
        self._check(L.lib.nfagg_sketch_device_ptr(self._h, which, C.byref(p), C.byref(b)))
        return p.value, b.value

    def sketch_reset(self):
        self._check(L.lib.nfagg_sketch_reset(self._h))

    def hll_estimate(self, which):
        v = C.c_double(0)
        self._check(L.lib.nfagg_hll_estimate(self._h, which, C.byref(v)))
        return v.value

    def cm_query(self, which, ip16: bytes):
        v = C.c_uint64(0)
        buf = (C.c_uint8 * 16).from_buffer_copy(bytes(ip16))
        self._check(L.lib.nfagg_cm_query(self._h, which, buf, C.byref(v)))
        return v.value

    HEAVY_HITTER = np.dtype([("ip", "u1", 16), ("estimate", "<u8")])

    def cm_topk(self, which, records, k: int, device_ptr: int = 0, n: int = 0) -> np.ndarray:
        """nfagg_cm_topk: the k heaviest endpoints (Count-Min estimate) among the addresses of `records` (host array), or of
        the n records at device_ptr."""
        out = np.zeros(max(k, 1), dtype=self.HEAVY_HITTER)
        n_out = C.c_size_t(0)
        if device_ptr:
            rc = L.lib.nfagg_cm_topk_device(self._h, which, C.c_void_p(device_ptr), n, k, out.ctypes.data_as(C.c_void_p), C.byref(n_out))
        else:
            r = np.ascontiguousarray(records)
            rc = L.lib.nfagg_cm_topk(self._h, which, r.ctypes.data_as(C.c_void_p), r.nbytes // 144, k, out.ctypes.data_as(C.c_void_p), C.byref(n_out))
        self._check(rc)
        return out[: n_out.value]

    # -- misc
    # -- export encode (record -> protobuf), nfagg_encode_pb
    def _pb_options(self, now_unix_ns, mono_now_ns, agent_ip16, names, unknown):
        o = L.PbOptions()
        o.struct_size = C.sizeof(L.PbOptions)
        o.now_unix_ns, o.mono_now_ns = now_unix_ns, mono_now_ns
        o.agent_ip[:] = list(bytes(agent_ip16))
        names = np.ascontiguousarray(names)
        o.names, o.n_names = (names.ctypes.data if len(names) else None), len(names)
        o.unknown_name, o.unknown_len = unknown, len(unknown)
        return o, names

    @staticmethod
    def _pb_features(n, present, parts, device=False):
        """nfagg_pb_features from `present` (n bytes of FEAT_* bits) and parts = {"additional"|"dns"|"drops"|"xlat"|"quic":
        array of n structs}; with device=True the values are raw device pointers."""
        f = L.PbFeatures()
        f.struct_size = C.sizeof(L.PbFeatures)
        keep = []
        if device:
            f.present = present or None
            for k, v in parts.items():
                setattr(f, k, v or None)
            return f, keep
        p = np.ascontiguousarray(present, dtype=np.uint8)
        assert p.size == n
        keep.append(p)
        f.present = p.ctypes.data
        for k, v in parts.items():
            a = np.ascontiguousarray(v)
            assert a.dtype == ROLLUP_KINDS[k] and len(a) == n, k
            keep.append(a)
            setattr(f, k, a.ctypes.data)
        return f, keep

    def encode_pb(self, records: np.ndarray, now_unix_ns: int, mono_now_ns: int, agent_ip16: bytes, names: np.ndarray,
                  unknown: bytes = b"unknown", kafka_keys=False, present=None, parts=None):
        """FlowsToPB + proto.Marshal of evicted records on the GPU. Returns (buf, frame_offsets, body_len[, keys]):
        buf[frame_offsets[a]:frame_offsets[b]] is a serialized pbflow.Records of entries a..b-1; the last body_len[i]
        bytes of frame i are the serialized pbflow.Record. With present/parts (see _pb_features) the flows are full
        BpfFlowContents of the MapTracer branch (nfagg_encode_pb_content)."""
        r = np.ascontiguousarray(records)
        n = r.nbytes // 144
        o, keep = self._pb_options(now_unix_ns, mono_now_ns, agent_ip16, names, unknown)
        feat = None
        if present is not None:
            feat, keep_f = self._pb_features(n, present, parts or {})
        off = np.zeros(n + 1, dtype=np.uint64)
        blen = np.zeros(max(n, 1), dtype=np.uint32)
        keys = np.zeros((max(n, 1), 32), dtype=np.uint8) if kafka_keys else None
        need = C.c_size_t(0)
        cap = max(256 * n, 64)
        while True:
            buf = np.zeros(cap, dtype=np.uint8)
            tail = (C.byref(o), buf.ctypes.data_as(C.c_void_p), cap, off.ctypes.data_as(C.c_void_p), blen.ctypes.data_as(C.c_void_p),
                    keys.ctypes.data_as(C.c_void_p) if kafka_keys else None, C.byref(need))
            if feat is None:
                rc = L.lib.nfagg_encode_pb(self._h, r.ctypes.data_as(C.c_void_p), n, *tail)
            else:
                rc = L.lib.nfagg_encode_pb_content(self._h, r.ctypes.data_as(C.c_void_p), n, C.byref(feat), *tail)
            if rc == L.TRUNCATED:
                cap = need.value
                continue
            self._check(rc)
            break
        out = (buf[: need.value], off, blen[:n])
        return out + (keys[:n],) if kafka_keys else out

    def encode_pb_device(self, d_records: int, n: int, now_unix_ns: int, mono_now_ns: int, agent_ip16: bytes, names: np.ndarray,
                         d_out: int, out_cap: int, d_frame_offsets: int, d_body_len: int, d_kafka_keys: int = 0,
                         unknown: bytes = b"unknown", d_present: int = 0, d_parts=None):
        """Device-resident variant (raw device pointers). Returns (rc, bytes needed/written)."""
        o, keep = self._pb_options(now_unix_ns, mono_now_ns, agent_ip16, names, unknown)
        need = C.c_size_t(0)
        tail = (C.byref(o), C.c_void_p(d_out or None), out_cap, C.c_void_p(d_frame_offsets), C.c_void_p(d_body_len),
                C.c_void_p(d_kafka_keys or None), C.byref(need))
        if d_present:
            feat, _ = self._pb_features(n, d_present, d_parts or {}, device=True)
            rc = L.lib.nfagg_encode_pb_content_device(self._h, C.c_void_p(d_records), n, C.byref(feat), *tail)
        else:
            rc = L.lib.nfagg_encode_pb_device(self._h, C.c_void_p(d_records), n, *tail)
        self._check(rc, ok=(L.OK, L.TRUNCATED))
        return rc, need.value

    def stats(self) -> L.Stats:
        s = L.Stats()
        self._check(L.lib.nfagg_stats_get(self._h, C.byref(s)))
        return s

    def reset_profile(self):
        self._check(L.lib.nfagg_stats_reset_profile(self._h))

    def sync(self):
        self._check(L.lib.nfagg_sync(self._h))

    # -- local fold across GPUs, one process per GPU (nfagg_partials_*)
    def set_sequence(self, next_seq: int):
        self._check(L.lib.nfagg_set_sequence(self._h, next_seq))

    @property
    def partial_bytes(self) -> int:
        """nfagg_partial_bytes: 192, or 256 for the sub-flow partials of a kernel-dedup handle."""
        return int(L.lib.nfagg_partial_bytes(self._h))

    def partials_export_device(self, n_shards: int, self_shard: int, d_out: int, cap: int):
        """The live flows as partials of partial_bytes grouped by owner shard, into device memory at d_out (room for cap partials).
        Returns (rc, counts[n_shards], n): rc TRUNCATED = nothing written, n partials needed."""
        counts = (C.c_uint64 * n_shards)()
        n = C.c_size_t(0)
        rc = L.lib.nfagg_partials_export_device(self._h, n_shards, self_shard, C.c_void_p(d_out or None), cap, counts, C.byref(n))
        self._check(rc, (L.OK, L.TRUNCATED))
        return rc, [int(x) for x in counts], n.value

    def partials_merge_device(self, n_shards: int, shard_id: int, d_partials: int, n: int):
        self._check(L.lib.nfagg_partials_merge_device(self._h, n_shards, shard_id, C.c_void_p(d_partials or None), n))

    def window_restart_device(self, n_shards: int, shard_id: int, d_partials: int, n: int, next_seq: int):
        self._check(L.lib.nfagg_window_restart_device(self._h, n_shards, shard_id, C.c_void_p(d_partials or None), n, next_seq))

    def evict_owned_device(self, n_shards: int, shard_id: int, d_out: int, cap: int, reason=L.REASON_TIMEOUT):
        """Returns (rc, n): rc TRUNCATED = nothing evicted, n records needed."""
        n = C.c_size_t(0)
        rc = L.lib.nfagg_evict_owned_device(self._h, reason, n_shards, shard_id, C.c_void_p(d_out or None), cap, C.byref(n))
        self._check(rc, (L.OK, L.TRUNCATED))
        return rc, n.value

    def debug_skip_sequence(self, records: int):
        self._check(L.lib.nfagg_debug_skip_sequence(self._h, records))

    @property
    def stream(self) -> int:
        return L.lib.nfagg_stream(self._h) or 0


class _Member(FlowTable):
    """A group member's handle, borrowed: queries only (sketches, stats, export); the group owns and destroys it."""

    def __init__(self, handle, max_entries):
        self._h = C.c_void_p(handle)
        self.max_entries = max_entries

    def close(self):
        self._h = None


class FlowGroup:
    """nfagg_group (include/nfagg.h): N GPUs behind one process; flows shard by key hash, member i owns shard i.
    local_fold: NFAGG_GROUP_LOCAL_FOLD — chunks are folded where they arrive, the members' slots are merged into their
    owners at the eviction."""

    def __init__(self, devices, max_entries=5000, mode=L.MODE_ACCOUNTER, sketches=0, cm_depth=0, cm_log2_width=0, hll_p=0,
                 staging_records=0, profile=False, ingest_variant=0, local_fold=False):
        cfg = L.Config()
        cfg.struct_size = C.sizeof(L.Config)
        cfg.max_entries = max_entries
        cfg.mode = mode
        cfg.sketch_flags = sketches
        cfg.cm_depth, cfg.cm_log2_width, cfg.hll_p = cm_depth, cm_log2_width, hll_p
        cfg.staging_records = staging_records
        cfg.profile = 1 if profile else 0
        cfg.ingest_variant = ingest_variant
        cfg.group_flags = L.GROUP_LOCAL_FOLD if local_fold else 0
        self.local_fold = bool(local_fold)
        devs = (C.c_int32 * len(devices))(*devices)
        self._g = C.c_void_p()
        rc = L.lib.nfagg_group_create(C.byref(cfg), devs, len(devices), C.byref(self._g))
        if rc != L.OK:
            msg = L.lib.nfagg_group_last_error(None)
            self._g = None
            raise NfaggError(rc, msg.decode() if msg else "nfagg_group_create failed")
        self.n = len(devices)
        self.max_entries = max_entries
        share = max_entries if local_fold else (max_entries + self.n - 1) // self.n
        self.members = [_Member(L.lib.nfagg_group_member(self._g, i), share) for i in range(self.n)]

    def close(self):
        if getattr(self, "_g", None):
            for m in self.members:
                m.close()
            L.lib.nfagg_group_destroy(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, ok=(L.OK,)):
        if rc not in ok:
            msg = L.lib.nfagg_group_last_error(self._g)
            raise NfaggError(rc, msg.decode() if msg else "")
        return rc

    def ingest(self, records: np.ndarray):
        records = np.ascontiguousarray(records)
        n = records.nbytes // 144
        consumed = C.c_size_t(0)
        rc = L.lib.nfagg_group_ingest(self._g, records.ctypes.data_as(C.c_void_p), n, C.byref(consumed))
        self._check(rc, (L.OK, L.FULL))
        return rc, consumed.value

    def ingest_device(self, src_member: int, d_ptr: int, n: int):
        consumed = C.c_size_t(0)
        rc = L.lib.nfagg_group_ingest_device(self._g, src_member, C.c_void_p(d_ptr), n, C.byref(consumed))
        self._check(rc, (L.OK, L.FULL))
        return rc, consumed.value

    def __len__(self):
        v = C.c_uint64(0)
        self._check(L.lib.nfagg_group_len(self._g, C.byref(v)))
        return v.value

    def merge_sketches(self):
        self._check(L.lib.nfagg_group_merge_sketches(self._g))

    def debug_skip_sequence(self, records: int):
        self._check(L.lib.nfagg_group_debug_skip_sequence(self._g, records))

    def evict(self, reason=L.REASON_TIMEOUT) -> np.ndarray:
        cap = max(len(self), 1)
        out = np.zeros(cap, dtype=FLOW_RECORD)
        n = C.c_size_t(0)
        self._check(L.lib.nfagg_group_evict(self._g, reason, out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[: n.value]

    def evict_device(self, d_ptrs, caps, reason=L.REASON_TIMEOUT):
        """Shard i's flows into device buffer d_ptrs[i] (on member i's device). Returns the per-member counts."""
        p = (C.c_void_p * self.n)(*d_ptrs)
        c = (C.c_size_t * self.n)(*caps)
        n = (C.c_size_t * self.n)()
        rc = L.lib.nfagg_group_evict_device(self._g, reason, p, c, n)
        if rc == L.TRUNCATED:        # nothing was evicted; n holds what every member needs
            raise NfaggError(rc, "device buffers too small, needed %s" % [int(x) for x in n])
        self._check(rc)
        return [int(x) for x in n]


def key_hash(flow_id_bytes: bytes) -> int:
    buf = (C.c_uint8 * 40).from_buffer_copy(bytes(flow_id_bytes)[:40])
    return L.lib.nfagg_key_hash(buf)


def shard_of(flow_id_bytes: bytes, n_shards: int) -> int:
    buf = (C.c_uint8 * 40).from_buffer_copy(bytes(flow_id_bytes)[:40])
    return L.lib.nfagg_shard_of(buf, n_shards)


def ip_hash(ip16: bytes, seed_index: int) -> int:
    buf = (C.c_uint8 * 16).from_buffer_copy(bytes(ip16))
    return L.lib.nfagg_ip_hash(buf, seed_index)


def hll_estimate_from_histogram(hist, p: int) -> float:
    h = np.ascontiguousarray(hist, dtype=np.uint32)
    assert h.size == 65
    return L.lib.nfagg_hll_estimate_from_histogram(h.ctypes.data_as(C.c_void_p), p)


def record_times(now_unix_ns: int, mono_now_ns: int, metrics: np.void):
    """pkg/model/record.go:90-97 via the library helper."""
    m = np.ascontiguousarray(np.array([metrics], dtype=FLOW_METRICS))
    a, b = C.c_int64(0), C.c_int64(0)
    L.lib.nfagg_record_times(now_unix_ns, mono_now_ns, m.ctypes.data_as(C.c_void_p), C.byref(a), C.byref(b))
    return a.value, b.value
