"""netobserv-ebpf-agent_amd — MI355X-native flow aggregation behind
netobserv-ebpf-agent's pkg/flow.Accounter / MapTracer path.

The product is the HIP library lib/libnfagg.so (C ABI: include/nfagg.h).
This package is the host-side mirror of the reference's interface for that
path (accounter.py) plus ctypes plumbing. Importing it fails loudly when the
library has not been built — there is no CPU or pure-Python fallback.
"""
from . import _lib
from ._lib import (OK, FULL, TRUNCATED, REASON_TIMEOUT, REASON_FULL, REASON_CLOSING, SKETCH_CM, SKETCH_HLL,
                   CM_SRC, CM_DST, HLL_SRC, HLL_DST, MODE_ACCOUNTER, MODE_KERNEL_DEDUP,
                   FEAT_ADDITIONAL, FEAT_DNS, FEAT_DROPS, FEAT_NETWORK_EVENTS, FEAT_XLAT, FEAT_QUIC)
from .records import (FLOW_ID, FLOW_METRICS, FLOW_RECORD, ADDITIONAL, DNS, PKT_DROP, NETWORK_EVENTS, XLAT, QUIC,
                      ROLLUP_KINDS, sort_by_key, INTF_NAME, intf_table)
from .table import (FlowTable, FlowGroup, NfaggError, PinnedRecords, key_hash, shard_of, ip_hash, hll_estimate_from_histogram, record_times,
                    host_threads, host_info, device_numa_node)
from .accounter import (Accounter, NewAccounter, NewRecord, Record, IntfDirUdn, NewIntfDirUdn, Metrics, NoOp, CLOSE,
                        SetInterfaceNamer, SetGlobalIP)
from . import synth
from . import pipeline
from .pipeline import (CapacityLimiter, RecordToMap, DirectFLPStdout, BpfFlowContent, GPUMapFetcher, MapTracer, NewMapTracer,
                       FlowsToPBMessages)
from . import distributed
