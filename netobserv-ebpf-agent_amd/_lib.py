"""ctypes binding of libnfagg.so (include/nfagg.h). There is no fallback: if the
HIP library has not been built, importing this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NFAGG_LIB: tools/phase_timing.py points this at lib/libnfagg_diag.so (the -DNFAGG_DIAG build with the phase-timing kernels)
LIB_PATH = os.environ.get("NFAGG_LIB") or os.path.join(_HERE, "lib", "libnfagg.so")
SYNTH_PATH = os.path.join(_HERE, "lib", "libnfagg_synth.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
        "(or __graft_entry__.build()). libnfagg has no CPU/Python fallback.")

# PyTorch-ROCm bundles its own libamdhip64.so (same soname as /opt/rocm's). A
# process must use ONE HIP runtime: when torch is installed, let it load first so
# libnfagg.so binds to the runtime torch's tensors, streams and RCCL live in.
# (A Go/C host without torch simply gets /opt/rocm's runtime.)
try:
    import torch  # noqa: F401
except Exception:   # torch is optional plumbing, not a dependency of the library
    torch = None

lib = C.CDLL(LIB_PATH)

OK, FULL, TRUNCATED = 0, 1, 2
EINVAL, ENODEV, ENOMEM, EDEVICE, ESTATE, ERANGE = -1, -2, -3, -4, -5, -6
REASON_TIMEOUT, REASON_FULL, REASON_CLOSING = 0, 1, 2
REASON_NAMES = {0: "timeout", 1: "full", 2: "closing"}
MODE_ACCOUNTER, MODE_KERNEL_DEDUP = 0, 1
GROUP_LOCAL_FOLD = 1
PARTIAL_BYTES = 192
PARTIAL_BYTES_DEDUP = 256
SHARD_NONE = 0xFFFFFFFF
SKETCH_CM, SKETCH_HLL = 1, 2
CM_SRC, CM_DST, HLL_SRC, HLL_DST = 0, 1, 2, 3


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("max_entries", C.c_uint64),
        ("table_log2_slots", C.c_uint32), ("mode", C.c_uint32), ("sketch_flags", C.c_uint32),
        ("cm_depth", C.c_uint32), ("cm_log2_width", C.c_uint32), ("hll_p", C.c_uint32),
        ("staging_records", C.c_uint64), ("n_shards", C.c_uint32), ("shard_id", C.c_uint32),
        ("profile", C.c_uint32), ("ingest_variant", C.c_uint32), ("ext_sketch", C.c_void_p * 4),
        ("copy_threads", C.c_uint32), ("group_flags", C.c_uint32), ("local_fold", C.c_uint32),
    ]


class RingBuf(C.Structure):
    """nfagg_ringbuf (include/nfagg.h)."""
    _fields_ = [("data", C.c_void_p), ("mask", C.c_uint64), ("producer_pos", C.c_void_p), ("consumer_pos", C.c_void_p)]


class PbOptions(C.Structure):
    """nfagg_pb_options (include/nfagg.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_names", C.c_uint32), ("now_unix_ns", C.c_int64), ("mono_now_ns", C.c_uint64),
        ("agent_ip", C.c_uint8 * 16), ("names", C.c_void_p), ("unknown_name", C.c_char * 16), ("unknown_len", C.c_uint8),
        ("pad_", C.c_uint8 * 7),
    ]


class MapView(C.Structure):
    """nfagg_map_view (include/nfagg.h)."""
    _fields_ = [("ids", C.c_void_p), ("values", C.c_void_p), ("n", C.c_size_t)]


class MergedFlows(C.Structure):
    """nfagg_merged_flows (include/nfagg.h)."""
    _fields_ = [("records", C.c_void_p), ("present", C.c_void_p), ("additional", C.c_void_p), ("dns", C.c_void_p),
                ("drops", C.c_void_p), ("network_events", C.c_void_p), ("xlat", C.c_void_p), ("quic", C.c_void_p)]


class PbFeatures(C.Structure):
    """nfagg_pb_features (include/nfagg.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("reserved_", C.c_uint32), ("present", C.c_void_p), ("additional", C.c_void_p),
                ("dns", C.c_void_p), ("drops", C.c_void_p), ("xlat", C.c_void_p), ("quic", C.c_void_p)]


FEAT_ADDITIONAL, FEAT_DNS, FEAT_DROPS, FEAT_NETWORK_EVENTS, FEAT_XLAT, FEAT_QUIC = 1, 2, 4, 8, 16, 32


class Stats(C.Structure):
    _fields_ = [
        ("records_ingested", C.c_uint64), ("records_skipped", C.c_uint64), ("entries", C.c_uint64),
        ("evictions", C.c_uint64 * 3), ("evicted_flows", C.c_uint64 * 3), ("epoch_seq", C.c_uint64),
        ("table_slots", C.c_uint64), ("table_bytes", C.c_uint64),
        ("ingest_launches", C.c_uint64), ("ingest_kernel_ms", C.c_double),
        ("evict_launches", C.c_uint64), ("evict_kernel_ms", C.c_double),
        ("sketch_launches", C.c_uint64), ("sketch_kernel_ms", C.c_double), ("max_probe", C.c_uint64),
        ("records_bypassed", C.c_uint64),
        ("optimistic_folds", C.c_uint64), ("optimistic_rollbacks", C.c_uint64), ("sequence_rebases", C.c_uint64),
        ("account_epochs_first", C.c_uint64), ("account_chain", C.c_uint64), ("account_declined", C.c_uint64),
    ]


_vp, _sz, _psz = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)

# every symbol include/nfagg.h declares, with its signature
SIGNATURES = {
    "nfagg_abi_version": (C.c_uint32, []),
    "nfagg_create": (C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    "nfagg_destroy": (None, [_vp]),
    "nfagg_last_error": (C.c_char_p, [_vp]),
    "nfagg_ingest": (C.c_int, [_vp, _vp, _sz, _psz]),
    "nfagg_ingest_device": (C.c_int, [_vp, _vp, _sz, _psz]),
    "nfagg_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "nfagg_host_free": (None, [_vp]),
    "nfagg_host_threads": (C.c_int, [C.c_uint, C.c_int]),
    "nfagg_host_info": (C.c_int, [_vp]),
    "nfagg_device_numa_node": (C.c_int, [C.c_int]),
    "nfagg_staging_acquire": (C.c_int, [_vp, C.POINTER(_vp), _psz]),
    "nfagg_staging_commit": (C.c_int, [_vp, _sz, _psz]),
    "nfagg_len": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "nfagg_evict": (C.c_int, [_vp, C.c_int, _vp, _sz, _psz]),
    "nfagg_evict_device": (C.c_int, [_vp, C.c_int, _vp, _sz, _psz]),
    "nfagg_account": (C.c_int, [_vp, _vp, _sz, _vp, _sz, C.POINTER(C.c_uint64), _sz, _psz, _psz]),
    "nfagg_account_device": (C.c_int, [_vp, _vp, _sz, _vp, _sz, C.POINTER(C.c_uint64), _sz, _psz, _psz]),
    "nfagg_record_times": (None, [C.c_int64, C.c_uint64, _vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "nfagg_rollup_additional": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "nfagg_rollup_dns": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "nfagg_rollup_drops": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "nfagg_rollup_network_events": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "nfagg_rollup_xlat": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "nfagg_rollup_quic": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "nfagg_sketch_snapshot": (C.c_int, [_vp, C.c_int, _vp, _sz]),
    "nfagg_sketch_device_ptr": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _psz]),
    "nfagg_sketch_reset": (C.c_int, [_vp]),
    "nfagg_hll_estimate": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    "nfagg_cm_query": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(C.c_uint64)]),
    "nfagg_hll_estimate_from_histogram": (C.c_double, [_vp, C.c_uint32]),
    "nfagg_ringbuf_drain": (C.c_int, [C.POINTER(RingBuf), _vp, _sz, _psz, _psz, _vp]),
    "nfagg_limit_batches": (C.c_size_t, [C.POINTER(C.c_uint64), _sz, _sz, _sz, _vp, C.POINTER(C.c_uint64)]),
    "nfagg_encode_pb": (C.c_int, [_vp, _vp, _sz, C.POINTER(PbOptions), _vp, _sz, _vp, _vp, _vp, _psz]),
    "nfagg_encode_pb_device": (C.c_int, [_vp, _vp, _sz, C.POINTER(PbOptions), _vp, _sz, _vp, _vp, _vp, _psz]),
    "nfagg_cm_topk": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _vp, _psz]),
    "nfagg_cm_topk_device": (C.c_int, [_vp, C.c_int, _vp, _sz, _sz, _vp, _psz]),
    "nfagg_map_merge": (C.c_int, [_vp, C.POINTER(MapView), C.POINTER(MapView), _sz, C.POINTER(MergedFlows), _sz, _psz, _psz]),
    "nfagg_map_merge_device": (C.c_int, [_vp, C.POINTER(MapView), C.POINTER(MapView), _sz, C.POINTER(MergedFlows), _sz, _psz, _psz]),
    "nfagg_encode_pb_content": (C.c_int, [_vp, _vp, _sz, C.POINTER(PbFeatures), C.POINTER(PbOptions), _vp, _sz, _vp, _vp, _vp, _psz]),
    "nfagg_encode_pb_content_device": (C.c_int, [_vp, _vp, _sz, C.POINTER(PbFeatures), C.POINTER(PbOptions), _vp, _sz, _vp, _vp, _vp, _psz]),
    "nfagg_shard_of": (C.c_uint32, [_vp, C.c_uint32]),
    "nfagg_shard_ids": (None, [_vp, _sz, C.c_uint32, _vp]),
    "nfagg_key_hash": (C.c_uint64, [_vp]),
    "nfagg_ip_hash": (C.c_uint64, [_vp, C.c_uint32]),
    "nfagg_stats_get": (C.c_int, [_vp, C.POINTER(Stats)]),
    "nfagg_stats_reset_profile": (C.c_int, [_vp]),
    "nfagg_sync": (C.c_int, [_vp]),
    "nfagg_stream": (_vp, [_vp]),
    "nfagg_debug_skip_sequence": (C.c_int, [_vp, C.c_uint64]),
    "nfagg_set_sequence": (C.c_int, [_vp, C.c_uint64]),
    "nfagg_partial_bytes": (C.c_size_t, [_vp]),
    "nfagg_partials_export_device": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, _sz, C.POINTER(C.c_uint64), _psz]),
    "nfagg_partials_merge_device": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, _sz]),
    "nfagg_window_restart_device": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, _sz, C.c_uint64]),
    "nfagg_evict_owned_device": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint32, _vp, _sz, _psz]),
    "nfagg_group_debug_skip_sequence": (C.c_int, [_vp, C.c_uint64]),
    "nfagg_group_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(_vp)]),
    "nfagg_group_destroy": (None, [_vp]),
    "nfagg_group_last_error": (C.c_char_p, [_vp]),
    "nfagg_group_size": (C.c_uint32, [_vp]),
    "nfagg_group_member": (_vp, [_vp, C.c_uint32]),
    "nfagg_group_ingest": (C.c_int, [_vp, _vp, _sz, _psz]),
    "nfagg_group_ingest_device": (C.c_int, [_vp, C.c_uint32, _vp, _sz, _psz]),
    "nfagg_group_len": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "nfagg_group_merge_sketches": (C.c_int, [_vp]),
    "nfagg_group_evict": (C.c_int, [_vp, C.c_int, _vp, _sz, _psz]),
    "nfagg_group_evict_device": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _psz, _psz]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = the .so does not export the ABI
    _fn.restype = _res
    _fn.argtypes = _args


def load_synth():
    """libnfagg_synth.so: device-side synthetic stream generator (bench/test support)."""
    if not os.path.exists(SYNTH_PATH):
        raise ImportError(f"{SYNTH_PATH} is missing: build with make -C {os.path.join(_HERE, 'csrc')}")
    s = C.CDLL(SYNTH_PATH)
    u64 = C.c_uint64
    s.nfagg_synth_stream.restype = C.c_int
    s.nfagg_synth_stream.argtypes = [_vp, u64, u64, u64, u64, _vp, C.c_uint32, C.c_uint32, _vp, _vp]
    s.nfagg_synth_zipf_thresholds.restype = None
    s.nfagg_synth_zipf_thresholds.argtypes = [u64, C.c_double, _vp]
    s.nfagg_synth_shard_population.restype = None
    s.nfagg_synth_shard_population.argtypes = [u64, C.c_uint32, C.c_uint32, _vp]
    s.nfagg_synth_stream_host.restype = None
    s.nfagg_synth_stream_host.argtypes = [_vp, u64, u64, u64, u64, _vp, C.c_uint32, C.c_uint32, _vp]
    s.nfagg_synth_yardstick.restype = C.c_double
    s.nfagg_synth_yardstick.argtypes = [C.c_int, _vp, _vp, u64, _vp, C.c_int]
    return s
