/*
 * nfagg.h — C ABI of libnfagg, the MI355X (gfx950) flow-aggregation backend.
 *
 * This header is the drop-in boundary behind netobserv-ebpf-agent's userspace
 * flow stage. Every entry point names the reference interface it replaces
 * (paths relative to the reference repository root). The reference-side cgo
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Rules of the boundary:
 *   - plain C, plain pointers and sizes; no callbacks into the caller, no
 *     pointer retained after a call returns (cgo pointer rules);
 *   - the caller owns input buffers until the call returns, the library owns
 *     the flow table, eviction output goes into caller-provided buffers;
 *   - one producer per handle; calls on a handle are synchronous and must not
 *     overlap (exactly one goroutine runs Accounter.Account, account.go:58);
 *   - return value 0 = NFAGG_OK, >0 = a condition the caller must act on,
 *     <0 = error (nfagg_last_error gives the text). No Go-visible panics.
 *
 * There is NO CPU fallback behind this ABI: if the HIP runtime or a gfx950
 * device is missing, nfagg_create fails with NFAGG_ENODEV.
 */
#ifndef NFAGG_H
#define NFAGG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFAGG_ABI_VERSION 2u   /* 2 (round 6): nfagg_stats grew by account_epochs_first / account_chain / account_declined in round 5 without a bump — a
                                   caller built against version 1's header must not be handed the longer struct; cfg.copy_threads = 0 means "the
                                   calibrated number of parts" (was: 4) */

/* ------------------------------------------------------------------ */
/* Record ABI — byte-for-byte the structs of bpf/types.h               */
/* ------------------------------------------------------------------ */

/* bpf/types.h:191-204 flow_id; pkg/ebpf/bpf_x86_bpfel.go:108-120 BpfFlowId.
 * 40 bytes, alignment 2. Byte 39 is padding: the library zeroes it on ingest
 * (the kernel memsets it, bpf/flows.c:177-178; Go ignores it). */
typedef struct nfagg_flow_id {
    uint8_t  src_ip[16];          /* IPv4 as ::ffff:a.b.c.d */
    uint8_t  dst_ip[16];
    uint16_t src_port;            /* host endian */
    uint16_t dst_port;
    uint8_t  transport_protocol;
    uint8_t  icmp_type;
    uint8_t  icmp_code;
    uint8_t  pad_;
} nfagg_flow_id;

/* bpf/types.h:94-126 flow_metrics; bpf_x86_bpfel.go:122-153 BpfFlowMetrics.
 * 104 bytes, alignment 8. */
typedef struct nfagg_flow_metrics {
    uint64_t start_mono_time_ts;  /* @0  */
    uint64_t end_mono_time_ts;    /* @8  */
    uint64_t bytes;               /* @16 */
    uint32_t packets;             /* @24 */
    uint16_t eth_protocol;        /* @28 */
    uint16_t flags;               /* @30 */
    uint8_t  src_mac[6];          /* @32 */
    uint8_t  dst_mac[6];          /* @38 */
    uint32_t if_index_first_seen; /* @44 */
    uint32_t lock;                /* @48 struct bpf_spin_lock */
    uint32_t sampling;            /* @52 */
    uint8_t  direction_first_seen;/* @56 */
    uint8_t  errno_;              /* @57 */
    uint8_t  dscp;                /* @58 */
    uint8_t  nb_observed_intf;    /* @59 */
    uint8_t  observed_direction[6];/* @60 */
    uint8_t  pad2_[2];            /* @66 */
    uint32_t observed_intf[6];    /* @68 */
    uint16_t ssl_version;         /* @92 */
    uint16_t tls_cipher_suite;    /* @94 */
    uint16_t tls_key_share;       /* @96 */
    uint8_t  tls_types;           /* @98 */
    uint8_t  misc_flags;          /* @99 */
    uint8_t  pad4_[4];            /* @100 */
} nfagg_flow_metrics;

/* bpf/types.h:212-215 flow_record; pkg/model/record.go:63 RawRecord.
 * 144 bytes — the unit on the ring buffer (byte-exact vector:
 * pkg/model/record_test.go:19-102). */
typedef struct nfagg_flow_record {
    nfagg_flow_id      id;        /* @0  */
    nfagg_flow_metrics metrics;   /* @40 */
} nfagg_flow_record;

/* bpf/types.h:174-181 additional_metrics (RTT / IPsec). 32 bytes. */
typedef struct nfagg_additional_metrics {
    uint64_t start_mono_time_ts;
    uint64_t end_mono_time_ts;
    uint64_t flow_rtt;
    int32_t  ipsec_encrypted_ret;
    uint16_t eth_protocol;
    uint8_t  ipsec_encrypted;     /* bool */
    uint8_t  pad_;
} nfagg_additional_metrics;

/* bpf/types.h:131-140 dns_metrics. 64 bytes (name is NOT 2-aligned). */
typedef struct nfagg_dns_metrics {
    uint64_t start_mono_time_ts;
    uint64_t end_mono_time_ts;
    uint64_t latency;
    uint16_t id;
    uint16_t flags;
    uint16_t eth_protocol;
    uint8_t  errno_;
    char     name[32];
    uint8_t  pad_;
} nfagg_dns_metrics;

/* bpf/types.h:142-151 pkt_drop_metrics. 32 bytes. */
typedef struct nfagg_pkt_drop_metrics {
    uint64_t start_mono_time_ts;
    uint64_t end_mono_time_ts;
    uint16_t bytes;
    uint16_t packets;
    uint32_t latest_drop_cause;
    uint16_t latest_flags;
    uint16_t eth_protocol;
    uint8_t  latest_state;
    uint8_t  pad_[3];
} nfagg_pkt_drop_metrics;

/* bpf/types.h:153-161 network_events_metrics. 72 bytes. */
typedef struct nfagg_network_events_metrics {
    uint64_t start_mono_time_ts;
    uint64_t end_mono_time_ts;
    uint8_t  network_events[4][8];
    uint16_t bytes[4];
    uint16_t packets[4];
    uint16_t eth_protocol;
    uint8_t  network_events_idx;
    uint8_t  pad_[5];
} nfagg_network_events_metrics;

/* bpf/types.h:163-172 xlat_metrics. 56 bytes. */
typedef struct nfagg_xlat_metrics {
    uint64_t start_mono_time_ts;
    uint64_t end_mono_time_ts;
    uint8_t  saddr[16];
    uint8_t  daddr[16];
    uint16_t sport;
    uint16_t dport;
    uint16_t zone_id;
    uint16_t eth_protocol;
} nfagg_xlat_metrics;

/* bpf/types.h quic_metrics_t. 24 bytes. */
typedef struct nfagg_quic_metrics {
    uint64_t start_mono_time_ts;
    uint64_t end_mono_time_ts;
    uint32_t version;
    uint16_t eth_protocol;
    uint8_t  seen_long_hdr;
    uint8_t  seen_short_hdr;
} nfagg_quic_metrics;

/* ------------------------------------------------------------------ */
/* Status codes                                                         */
/* ------------------------------------------------------------------ */
enum {
    NFAGG_OK       = 0,
    /* >0: the caller must act, nothing went wrong */
    NFAGG_FULL     = 1,  /* ingest stopped before a record whose NEW key would
                            exceed max_entries (account.go:85): evict with
                            NFAGG_REASON_FULL, then resubmit the remainder.
                            (An epoch is never ended for lack of sequence
                            numbers: the slots' 32-bit sequence tags are
                            window-relative and the window moves,
                            stats.sequence_rebases.) */
    NFAGG_TRUNCATED = 2, /* output buffer smaller than the result */
    /* <0: errors */
    NFAGG_EINVAL   = -1,
    NFAGG_ENODEV   = -2, /* no HIP runtime / no gfx950 device: there is no CPU path */
    NFAGG_ENOMEM   = -3,
    NFAGG_EDEVICE  = -4, /* a HIP call failed; see nfagg_last_error */
    NFAGG_ESTATE   = -5, /* call not valid in the handle's current state */
    NFAGG_ERANGE   = -6, /* an argument is out of the supported range (e.g. map merge over more than 2^30 rows) */
};

/* Eviction reasons — the label values of the reference's Prometheus counters
 * (pkg/flow/account.go:71,78,90; pkg/metrics/metrics.go). */
enum {
    NFAGG_REASON_TIMEOUT = 0,   /* "timeout" */
    NFAGG_REASON_FULL    = 1,   /* "full"    */
    NFAGG_REASON_CLOSING = 2,   /* "closing" */
};

/* Accumulation semantics of the flow table. */
enum {
    /* pkg/model/flow_content.go:28-61 AccumulateBase, first record of a key
     * stored whole (account.go:95). This is what Accounter does. */
    NFAGG_MODE_ACCOUNTER = 0,
    /* bpf/flows.c:98-143 update_existing_flow + :76-96 add_observed_intf:
     * bytes/packets counted only on if_index_first_seen, other interfaces are
     * appended to observed_intf[] ("direction dedup"). */
    NFAGG_MODE_KERNEL_DEDUP = 1,
};

enum {
    NFAGG_SKETCH_CM  = 1u,  /* Count-Min over src IP and dst IP, adding bytes */
    NFAGG_SKETCH_HLL = 2u,  /* HyperLogLog over src IP and dst IP */
};

/* Sketch identifiers for snapshot / device-pointer / estimate calls. */
enum {
    NFAGG_CM_SRC  = 0,   /* uint64_t[cm_depth << cm_log2_width] */
    NFAGG_CM_DST  = 1,
    NFAGG_HLL_SRC = 2,   /* uint8_t [1 << hll_p], on the device and in snapshots */
    NFAGG_HLL_DST = 3,
};

/* ------------------------------------------------------------------ */
/* Handle and configuration                                             */
/* ------------------------------------------------------------------ */
typedef struct nfagg_handle nfagg_handle;

/* Replaces the arguments of flow.NewAccounter (pkg/flow/account.go:34-53;
 * call site pkg/agent/agent.go:208-212). Clocks, Prometheus metrics and the
 * OVN sample decoder stay on the Go side. Zero = default for every field
 * except struct_size. */
typedef struct nfagg_config {
    uint32_t struct_size;        /* sizeof(nfagg_config); ABI guard */
    int32_t  device;             /* HIP device ordinal */
    uint64_t max_entries;        /* CACHE_MAX_FLOWS (config.go:146); 0 -> 5000 */
    uint32_t table_log2_slots;   /* 0 -> smallest 2^k >= 2*max_entries (min 2^16) */
    uint32_t mode;               /* NFAGG_MODE_* */
    uint32_t sketch_flags;       /* NFAGG_SKETCH_* */
    uint32_t cm_depth;           /* 0 -> 4  (1..8) */
    uint32_t cm_log2_width;      /* 0 -> 20 */
    uint32_t hll_p;              /* 0 -> 14 (4..18) */
    uint64_t staging_records;    /* pinned staging ring, records per buffer; 0 -> 1<<20 */
    uint32_t n_shards;           /* 0/1 -> unsharded. Records whose
                                    nfagg_shard_of(id,n_shards) != shard_id are
                                    skipped on ingest and counted in stats */
    uint32_t shard_id;
    uint32_t profile;            /* 1 -> bracket kernels with HIP events (stats) */
    uint32_t ingest_variant;     /* 0 -> default kernels by batch size (direct kernel below 6144 records,
                                    single-pass LDS-cached kernel below 384 Ki, two-pass partitioned fold from
                                    there); others are A/B and diagnostic builds, see DESIGN.md §4.1b */
    /* Optional caller-owned DEVICE buffers for the sketches (so that another
     * library, e.g. RCCL via torch.distributed, can all-reduce them in place).
     * NULL -> the library allocates. Sizes as listed under NFAGG_CM_* above
     * (HLL: one byte per register, buffer 4-byte aligned). */
    void*    ext_sketch[4];
    uint32_t copy_threads;       /* parts a caller buffer is cut into for the copy workers (nfagg_host_threads) on its way into
                                    the pinned staging ring (one core moves ~28 GB/s, PCIe Gen5 x16 takes ~57); 0 -> what the
                                    workers' calibration found best on this host, 1 -> inline */
    uint32_t group_flags;        /* nfagg_group_create only: NFAGG_GROUP_* */
    uint32_t local_fold;         /* nfagg_create only (a NFAGG_GROUP_LOCAL_FOLD group sets it for its members): 1 = this handle is
                                    one rank of a local-fold job (nfagg_set_sequence / nfagg_partials_* below): it folds whatever
                                    part of ONE record stream arrives at it and other tables hold other records of the same flows.
                                    NFAGG_MODE_ACCOUNTER: no effect (a slot is a mergeable partial as it is).
                                    NFAGG_MODE_KERNEL_DEDUP: what a table counts for a flow depends on the flow's FIRST interface
                                    (bpf/flows.c:100-126), and that is the interface of the earliest record anywhere in the job —
                                    so the table is keyed by the SUB-FLOW (flow key, if_index_first_seen) and keeps, per
                                    interface, both what a counted and what a side interface needs; the flow itself is put
                                    together when the epoch ends (the sub-flows of all ranks having met at the flow's owner):
                                    nfagg_evict* / nfagg_evict_owned_device deliver exactly what ONE table over all the records
                                    would. Differences on such a handle: max_entries and nfagg_len count (flow, interface) pairs
                                    — NFAGG_FULL comes when a NEW pair finds max_entries of them; partials are
                                    NFAGG_PARTIAL_BYTES_DEDUP bytes; the 32-bit sequence window does not move: an epoch ends
                                    (NFAGG_FULL) once 2^32 - 16 sequence numbers have gone by in the job, and
                                    nfagg_window_restart_device is refused. */
} nfagg_config;

enum {
    /* Local-fold ("combiner") mode of a group: no per-record routing. Every member folds the chunks that arrive on ITS
     * device, whatever their keys, with sequence numbers global to the group; a flow may live on several members at once. At
     * the eviction the members' raw slots — mergeable partials, sequence tags included — travel to the member that owns the
     * flow (nfagg_shard_of), are merged there exactly (sums, ORs, maxima, earliest-tag-wins words) and the owner evicts the
     * flow. xGMI carries 192 bytes per (flow, member) instead of 144 bytes per record, and one hot flow (BASELINE configs[4])
     * is folded by all GPUs instead of by the one that owns it. Every eviction is bit-identical to one Accounter that saw
     * the same records between the same two evictions. Differences to the routed mode: max_entries bounds every member's
     * table, undivided — NFAGG_FULL is returned when ONE member holds max_entries flows and meets a new one, which is never
     * earlier and can be later than one Accounter over the whole stream would (an eviction may deliver up to
     * N x max_entries flows); use it where evictions are timeout-driven (CACHE_ACTIVE_TIMEOUT) and max_entries is the
     * safety net. nfagg_group_len is an upper bound (a flow counts once per member that saw it). After an eviction call
     * that returned NFAGG_TRUNCATED the group only accepts the repeated eviction (ingest returns NFAGG_FULL): the members'
     * slots have been merged already. Both modes: in NFAGG_MODE_KERNEL_DEDUP the members are created with
     * nfagg_config.local_fold (tables keyed by (flow, interface), see there), every eviction is bit-identical to ONE
     * kernel-dedup table (bpf/flows.c:76-143) over the same records — BASELINE configs[4]'s hot flow alternating over two
     * interfaces is counted on the interface of its earliest record whichever member saw that record — and an epoch ends
     * (NFAGG_FULL) after 2^32 - 16 records. */
    NFAGG_GROUP_LOCAL_FOLD = 1u,
};

typedef struct nfagg_stats {
    uint64_t records_ingested;   /* accepted into the table (this shard) */
    uint64_t records_skipped;    /* not this shard */
    uint64_t entries;            /* live keys now  (accounter-entries gauge, account.go:98) */
    uint64_t evictions[3];       /* per NFAGG_REASON_* (evictions_total) */
    uint64_t evicted_flows[3];   /* per reason (evicted_flows_total) */
    uint64_t epoch_seq;          /* records ingested since the last eviction */
    uint64_t table_slots;
    uint64_t table_bytes;
    /* filled when cfg.profile != 0 (HIP events on the handle's stream) */
    uint64_t ingest_launches;
    double   ingest_kernel_ms;   /* sum of ingest-kernel durations */
    uint64_t evict_launches;
    double   evict_kernel_ms;
    uint64_t sketch_launches;
    double   sketch_kernel_ms;
    uint64_t max_probe;          /* longest probe sequence seen */
    uint64_t records_bypassed;   /* records that found no entry in a pass-1 LDS flow cache (spilled to the
                                    second pass, or merged into HBM one by one by the single-pass kernel) */
    uint64_t optimistic_folds;   /* batches with live + batch > max_entries folded whole and checked afterwards */
    uint64_t optimistic_rollbacks; /* ... of which crossed max_entries (account.go:85) and were rolled back and split */
    uint64_t sequence_rebases;     /* times the 32-bit window of the slots' sequence tags was moved (once per ~2^32 records of an
                                      epoch; the epoch itself goes on: account.go:58-100 has no maximum length) */
    uint64_t account_epochs_first; /* nfagg_account[_device]: launches that found their epochs first and folded them from the sorted call */
    uint64_t account_chain;        /* ... launches of the kernel chain (short calls; calls the first path declined) */
    uint64_t account_declined;     /* ... of which were calls the first path declined (too many other flows' records between a
                                      record and its previous occurrence among equal hash bits) */
} nfagg_stats;

uint32_t nfagg_abi_version(void);

/* Replaces flow.NewAccounter (pkg/flow/account.go:34-53). */
int nfagg_create(const nfagg_config* cfg, nfagg_handle** out);

/* Replaces nothing in the reference (Go GC frees the Accounter); required by
 * C ownership. Pending flows are discarded: evict with NFAGG_REASON_CLOSING
 * first (account.go:73-80). */
void nfagg_destroy(nfagg_handle* h);

/* Text of the last error on this handle (NULL handle: last create error). */
const char* nfagg_last_error(const nfagg_handle* h);

/* ------------------------------------------------------------------ */
/* Ingest — replaces the `case record, ok := <-in` arm of               */
/* Accounter.Account (pkg/flow/account.go:72-96), batched.              */
/* ------------------------------------------------------------------ */

/* records: n x 144-byte flow_record_t in HOST memory (what
 * model.ReadFrom decodes, pkg/model/record.go:227-231), in arrival order.
 * The records are folded in exactly that order. *consumed = number of
 * leading records folded; returns NFAGG_FULL when it stopped early. */
int nfagg_ingest(nfagg_handle* h, const void* records, size_t n, size_t* consumed);

/* Same, records already in DEVICE memory of cfg.device (16-byte aligned). ASYNCHRONOUS when the batch cannot fill the table
 * (live + n <= max_entries): the kernels run on the handle's stream after the call returned and read d_records (the fold,
 * and k_finalize's copy of the new flows' first records), so the buffer must stay valid and unmodified until the handle
 * synchronises — nfagg_sync, nfagg_len, nfagg_evict*, nfagg_stats_get, or work ordered after nfagg_stream(h). (The host
 * variants copy into the library's own staging ring and have no such requirement.) */
int nfagg_ingest_device(nfagg_handle* h, const void* d_records, size_t n, size_t* consumed);

/* Page-locked host memory (hipHostMalloc). Record buffers handed to nfagg_ingest / nfagg_account and output buffers handed to
 * nfagg_evict / nfagg_account that lie in page-locked memory — from here, or registered by the caller (hipHostRegister) — cross
 * PCIe by DMA straight from / into them; pageable buffers go through the library's pinned staging ring (one more host copy:
 * ~30 GB/s instead of the link's ~50). A Go caller keeps its batch in such a buffer instead of a Go slice. */
int nfagg_host_alloc(size_t bytes, void** p);
void nfagg_host_free(void* p);

/* The host side's copy workers (one pool per process; csrc/nfagg_hostpool.h): what moves records with host cores — a pageable
 * caller buffer into the pinned staging ring (nfagg_ingest / nfagg_account), evictions out of the pinned bounce buffers, the BPF
 * ring buffer into the staging buffer (nfagg_ringbuf_drain, the batch form of RingBufTracer's loop,
 * pkg/flow/tracer_ringbuf.go:112-134). Created by the first nfagg_create: 16 workers bound to the CPUs of that handle's GPU's
 * NUMA node, non-temporal copies, and the number of parts a large copy is cut into MEASURED on this host (a calibration of a few
 * milliseconds) rather than assumed. nfagg_host_threads re-shapes the pool: `threads` workers (0 = keep the number), bound to
 * `numa_node` (-1 = unbound); returns the workers running (>= 0) or NFAGG_EINVAL. An agent that wants its cores left alone
 * calls nfagg_host_threads(2, -1) — or sets nfagg_config.copy_threads = 1 and drains the ring itself. */
typedef struct nfagg_host_pool_info {
    uint32_t struct_size;        /* sizeof(nfagg_host_pool_info) */
    uint32_t workers;            /* threads in the pool (the calling thread always works too) */
    uint32_t parts;              /* parts a large copy is cut into: the calibration's choice */
    uint32_t bound;              /* 1: the workers are bound to numa_node's CPUs */
    int32_t  numa_node;          /* -1: unbound */
    uint32_t pad_;
    double   calibrated_gbs;     /* what the best setting copied in the calibration, GB/s (host memory to host memory) */
} nfagg_host_pool_info;
int nfagg_host_threads(unsigned threads, int numa_node);
int nfagg_host_info(nfagg_host_pool_info* out);
/* The NUMA node `device` hangs off (/sys/bus/pci/devices/<bdf>/numa_node); -1 when unknown. */
int nfagg_device_numa_node(int device);

/* Zero-copy producer path: borrow the next pinned staging buffer
 * (capacity = cfg.staging_records), fill it (e.g. straight from the eBPF ring,
 * pkg/flow/tracer_ringbuf.go:112-134), then commit the first n records.
 * commit has nfagg_ingest semantics. */
int nfagg_staging_acquire(nfagg_handle* h, void** buf, size_t* capacity_records);
int nfagg_staging_commit(nfagg_handle* h, size_t n, size_t* consumed);

/* len(c.entries) (account.go:85,98). Synchronises with the device. */
int nfagg_len(nfagg_handle* h, uint64_t* entries);

/* ------------------------------------------------------------------ */
/* Evict — replaces Accounter.evict (pkg/flow/account.go:102-124) up to */
/* the point where model.NewRecord is called per entry.                 */
/* ------------------------------------------------------------------ */

/* Writes every live flow as one 144-byte flow_record_t {id, folded metrics}
 * into `out` (HOST memory, room for `cap` records), clears the table, starts
 * a new epoch. Order of records is unspecified (Go map order is random;
 * the reference's tests compare by key, account_test.go:94-98).
 * If cap < live entries nothing is evicted: *n_out = live entries and
 * NFAGG_TRUNCATED is returned. The caller turns each record into a
 * model.Record with model.NewRecord(key, &content, now, mono, ...)
 * exactly as account.go:116-119 (helper: nfagg_record_times). */
int nfagg_evict(nfagg_handle* h, int reason, void* out, size_t cap, size_t* n_out);

/* Same with `d_out` in DEVICE memory. */
int nfagg_evict_device(nfagg_handle* h, int reason, void* d_out, size_t cap, size_t* n_out);

/* ------------------------------------------------------------------ */
/* Account — the record arm of Accounter.Account WITH its evictions on   */
/* "full" (pkg/flow/account.go:81-96) in one call.                      */
/* ------------------------------------------------------------------ */

/* Folds records in arrival order exactly as nfagg_ingest does, but does not stop at a record whose NEW key finds
 * len(entries) >= max_entries (account.go:85): as the reference does inline (:86-94) it evicts every live flow — appended to
 * `out`, reason "full" — restarts the epoch and inserts that record. On return *n_epochs evictions have taken place; the
 * e-th delivered out[epoch_end[e-1] .. epoch_end[e]) (epoch_end[-1] = 0), the caller turns each into one `[]*model.Record`
 * for the exporter (account.go:111-123, one channel send per eviction); the table holds the epoch in progress, as after
 * nfagg_ingest. *consumed = leading records folded. Returns NFAGG_OK when all n are, NFAGG_TRUNCATED when `out` has no room
 * for another eviction (out_cap - records written < live flows; keep out_cap >= max_entries) or max_epochs are used up:
 * drain `out`, then call again with the rest (a pending eviction is delivered first).
 * With a small CACHE_MAX_FLOWS (the reference ships 5000, pkg/config/config.go:146, deploys 10 000, scripts/agent.yml:35-36, and
 * benchmarks 1 k / 10 k / 100 k, pkg/flow/tracer_map_bench_test.go:64-111) the stream stops on "full" every few thousand records;
 * here that whole loop runs on the device (NFAGG_MODE_ACCOUNTER, max_entries <= 2^22). A call of more than five chain windows'
 * worth of records (n >= 5 / (1 / 16384 + 1 / (2 max_entries)): 31 k at 5000 entries, never more than 80 k) has its epochs FOUND
 * FIRST (csrc/nfagg_epoch_par.hip): previous-occurrence links from one sort of (key hash, index) keys, one prefix
 * count per epoch — WHERE the loop of account.go:81-96 evicts does not need the map — and every complete epoch is then folded on
 * its own, all of them at once, each flow's records gathered in arrival order and folded as flow_content.go:28-61 folds them,
 * straight into `out`; only the call's first epoch (it continues what the table holds) and its last (it stays live) touch the
 * table. Sketches are fed along. Shorter calls take a chain of small kernels driven by a control block in device memory, replayed
 * from hipGraphs of 2 / 6 / 24 windows (csrc/nfagg_epoch_chain.hip; max_entries <= 32768; beyond: the optimistic fold of
 * nfagg_ingest); ingest_variant 30 forces that chain for every call (tests). Same evictions, in the same order, either way.
 * A call costs ~80 us whatever it holds: gather records (64 Ki, or what 1 ms brings) before calling — INTEGRATION.md section 3.
 * All pointers HOST memory: */
int nfagg_account(nfagg_handle* h, const void* records, size_t n, void* out, size_t out_cap, uint64_t* epoch_end,
                  size_t max_epochs, size_t* n_epochs, size_t* consumed);
/* Same with d_records / d_out in DEVICE memory (16-byte aligned); epoch_end stays in HOST memory. Synchronous: d_records
 * may be reused when the call returns. */
int nfagg_account_device(nfagg_handle* h, const void* d_records, size_t n, void* d_out, size_t out_cap, uint64_t* epoch_end,
                         size_t max_epochs, size_t* n_epochs, size_t* consumed);

/* ------------------------------------------------------------------ */
/* Capacity limiter — replaces, for the evictions of ONE nfagg_account   */
/* call, the decision of CapacityLimiter.Limit                            */
/* (pkg/flow/limiter.go:28-38): `if len(out) < cap(out) || cap(out) == 0  */
/* { out <- i } else { dropped += len(i) }`.                              */
/* ------------------------------------------------------------------ */

/* nfagg_account delivers n_epochs evictions at once (out[epoch_end[e-1] .. epoch_end[e])); the reference's limiter sees them
 * one by one on a channel and drops a whole batch when the exporter's channel is full. The shim takes this decision BEFORE it
 * builds a single model.Record (model.NewRecord per flow is the hottest allocation site, pkg/model/record_bench_test.go:10-13):
 * queue_len / queue_cap = len(out) / cap(out) of the exporter's channel now. keep[e] = 1: batch e is forwarded (it then occupies
 * one more slot of the channel; nothing is assumed to drain meanwhile: the exporter can only make the outcome better), 0: dropped.
 * *dropped_flows = the flows of the dropped batches — what the shim adds to DroppedFlowsCounter("limiter", "full") and to the
 * droppedFlows the limiter's periodic warning reports (limiter.go:33-34,41-57). queue_cap == 0 (unbuffered channel) never
 * drops (limiter.go:30). Pure host arithmetic, no device. Returns the number of batches kept. */
size_t nfagg_limit_batches(const uint64_t* epoch_end, size_t n_epochs, size_t queue_len, size_t queue_cap, uint8_t* keep,
                           uint64_t* dropped_flows);

/* pkg/model/record.go:90-97: TimeFlowStart = now - (mono_now - start_mono),
 * TimeFlowEnd likewise; uint64 wrap then signed nanoseconds, as Go does.
 * now_unix_ns is the wall clock in ns since the Unix epoch. */
void nfagg_record_times(int64_t now_unix_ns, uint64_t mono_now_ns,
                        const nfagg_flow_metrics* m,
                        int64_t* time_flow_start_unix_ns,
                        int64_t* time_flow_end_unix_ns);

enum {   /* the per-CPU feature maps, in the order of the nfagg_rollup_* entries */
    NFAGG_ROLLUP_ADDITIONAL = 0, NFAGG_ROLLUP_DNS = 1, NFAGG_ROLLUP_DROPS = 2,
    NFAGG_ROLLUP_NETWORK_EVENTS = 3, NFAGG_ROLLUP_XLAT = 4, NFAGG_ROLLUP_QUIC = 5
};
/* bits of a "present" byte: which parts of model.BpfFlowContent are non-nil */
enum {
    NFAGG_FEAT_ADDITIONAL     = 1 << NFAGG_ROLLUP_ADDITIONAL,
    NFAGG_FEAT_DNS            = 1 << NFAGG_ROLLUP_DNS,
    NFAGG_FEAT_DROPS          = 1 << NFAGG_ROLLUP_DROPS,
    NFAGG_FEAT_NETWORK_EVENTS = 1 << NFAGG_ROLLUP_NETWORK_EVENTS,
    NFAGG_FEAT_XLAT           = 1 << NFAGG_ROLLUP_XLAT,
    NFAGG_FEAT_QUIC           = 1 << NFAGG_ROLLUP_QUIC
};

/* ------------------------------------------------------------------ */
/* Per-CPU map rollup — replaces lookupAndDeletePerCPUMap's accumulator */
/* closures (pkg/tracer/tracer.go:1057-1110,1118-1146).                 */
/* ------------------------------------------------------------------ */

/* For each of n_flows flows: fold n_cpu per-CPU partials (CPU index
 * ascending; element 0 adopted whole, elements 1.. folded into it) with the
 * matching model.Accumulate* (pkg/model/flow_content.go), and apply
 * buildBaseFromAdditional (flow_content.go:63-74) to base[i] for every
 * partial. `base` is in/out: the caller passes the flow's base metrics from
 * the main map, or zeroes when the main map had no entry
 * (tracer.go:1136-1139). All pointers are HOST memory.
 *   partials: n_flows * n_cpu structs, flow-major
 *   folded:   n_flows structs */
int nfagg_rollup_additional(nfagg_handle* h, const nfagg_additional_metrics* partials,
                            size_t n_flows, size_t n_cpu,
                            nfagg_flow_metrics* base, nfagg_additional_metrics* folded);
int nfagg_rollup_dns(nfagg_handle* h, const nfagg_dns_metrics* partials,
                     size_t n_flows, size_t n_cpu,
                     nfagg_flow_metrics* base, nfagg_dns_metrics* folded);
int nfagg_rollup_drops(nfagg_handle* h, const nfagg_pkt_drop_metrics* partials,
                       size_t n_flows, size_t n_cpu,
                       nfagg_flow_metrics* base, nfagg_pkt_drop_metrics* folded);
int nfagg_rollup_network_events(nfagg_handle* h, const nfagg_network_events_metrics* partials,
                                size_t n_flows, size_t n_cpu,
                                nfagg_flow_metrics* base, nfagg_network_events_metrics* folded);
int nfagg_rollup_xlat(nfagg_handle* h, const nfagg_xlat_metrics* partials,
                      size_t n_flows, size_t n_cpu,
                      nfagg_flow_metrics* base, nfagg_xlat_metrics* folded);
int nfagg_rollup_quic(nfagg_handle* h, const nfagg_quic_metrics* partials,
                      size_t n_flows, size_t n_cpu,
                      nfagg_flow_metrics* base, nfagg_quic_metrics* folded);

/* ------------------------------------------------------------------ */
/* Map merge — replaces FlowFetcher.LookupAndDeleteMap's join            */
/* (pkg/tracer/tracer.go:1022-1116) including lookupAndDeletePerCPUMap    */
/* (:1118-1146): the caller drains the eBPF maps (syscalls stay in Go)    */
/* and hands the raw arrays over; the join by flow id, the per-CPU folds  */
/* and buildBaseFromAdditional happen on the device in one call.          */
/* ------------------------------------------------------------------ */

/* One drained map: n keys and their values. Main map (aggregated_flows):
 * values = nfagg_flow_metrics[n]. Feature map k (NFAGG_ROLLUP_*): values =
 * that map's struct[n * n_cpu], flow-major (what cilium's per-CPU Lookup
 * returns per key). A key listed twice in one map keeps its first row (the
 * reference's second LookupAndDelete fails and is skipped, :1048-1052,
 * :1130-1134); such rows are counted in *n_duplicate_keys. */
typedef struct nfagg_map_view {
    const nfagg_flow_id* ids;
    const void*          values;
    size_t               n;
} nfagg_map_view;

/* Caller-allocated outputs, `cap` entries each. records[i] = {id, base metrics
 * after every buildBaseFromAdditional}; present[i] = NFAGG_FEAT_* bits (the
 * non-nil parts of model.BpfFlowContent); part arrays hold the folded part
 * (zeroes when absent) and may be NULL when not wanted. The layout is what
 * nfagg_encode_pb_content consumes. Flows come out in order of first
 * appearance: main map first, then the feature maps in the order the reference
 * walks them (dns, drops, network events, xlat, additional, quic) — Go's map
 * order is random, so any order is valid. id byte 39 (a blank field in Go) is
 * not part of the key and is written as zero. */
typedef struct nfagg_merged_flows {
    nfagg_flow_record*            records;
    uint8_t*                      present;
    nfagg_additional_metrics*     additional;
    nfagg_dns_metrics*            dns;
    nfagg_pkt_drop_metrics*       drops;
    nfagg_network_events_metrics* network_events;
    nfagg_xlat_metrics*           xlat;
    nfagg_quic_metrics*           quic;
} nfagg_merged_flows;

/* feature_maps[k], k = NFAGG_ROLLUP_*; n = 0 for a map that is not enabled.
 * NFAGG_TRUNCATED (nothing written, *n_out = flows) when cap is too small.
 * All pointers HOST memory: */
int nfagg_map_merge(nfagg_handle* h, const nfagg_map_view* main_map, const nfagg_map_view feature_maps[6],
                    size_t n_cpu, const nfagg_merged_flows* out, size_t cap, size_t* n_out, size_t* n_duplicate_keys);
/* Same with every data pointer in DEVICE memory (8-byte aligned). */
int nfagg_map_merge_device(nfagg_handle* h, const nfagg_map_view* d_main_map, const nfagg_map_view d_feature_maps[6],
                           size_t n_cpu, const nfagg_merged_flows* d_out, size_t cap, size_t* n_out, size_t* n_duplicate_keys);

/* ------------------------------------------------------------------ */
/* Sketches — new functionality (no reference counterpart; spec in      */
/* DESIGN.md §sketches, scalar oracle in oracle/).                      */
/* ------------------------------------------------------------------ */

/* Copy a sketch to HOST memory. CM: uint64_t[depth<<log2w]; HLL: uint8_t[1<<p]. */
int nfagg_sketch_snapshot(nfagg_handle* h, int which, void* out, size_t out_bytes);
/* Device pointer and byte size of a sketch array (for in-place RCCL all-reduce:
 * CM sum uint64, HLL max uint8 — 16 KiB per array at p = 14). */
int nfagg_sketch_device_ptr(nfagg_handle* h, int which, void** d_ptr, size_t* bytes);
/* Zero all sketches (start of a sketch window). */
int nfagg_sketch_reset(nfagg_handle* h);
/* Cardinality from the device registers: integer histogram of register values
 * computed on the GPU, FP64 estimate from the histogram in a fixed order.
 * which = NFAGG_HLL_SRC / NFAGG_HLL_DST. */
int nfagg_hll_estimate(nfagg_handle* h, int which, double* estimate);
/* Count-Min point query for one 16-byte IP: min over rows. Host-side read of
 * d counters. which = NFAGG_CM_SRC / NFAGG_CM_DST. */
int nfagg_cm_query(nfagg_handle* h, int which, const uint8_t ip[16], uint64_t* estimate);
/* Heavy hitters: the k endpoints with the largest Count-Min byte estimate among the addresses that occur in `records`
 * — Count-Min stores no keys, so the candidates come from a record batch, typically the one nfagg_evict just returned
 * (the sketch is keyed by src address for NFAGG_CM_SRC, by dst address for NFAGG_CM_DST, and the same side of each
 * record is looked up). Order: estimate descending, then the 16 address bytes ascending; *n_out = min(k, distinct
 * addresses). Estimates are computed and sorted on the device; `out` is HOST memory. */
typedef struct nfagg_heavy_hitter {
    uint8_t  ip[16];
    uint64_t estimate;
} nfagg_heavy_hitter;            /* 24 bytes */
int nfagg_cm_topk(nfagg_handle* h, int which, const void* records, size_t n, size_t k,
                  nfagg_heavy_hitter* out, size_t* n_out);
/* Same with d_records in DEVICE memory (16-byte aligned), e.g. straight from nfagg_evict_device. */
int nfagg_cm_topk_device(nfagg_handle* h, int which, const void* d_records, size_t n, size_t k,
                         nfagg_heavy_hitter* out, size_t* n_out);
/* The HLL estimator itself, on a host histogram hist[0..64] of register
 * values for m = 1<<p registers (exposed so callers can estimate after a
 * cross-GPU max-merge). */
double nfagg_hll_estimate_from_histogram(const uint32_t* hist, uint32_t p);

/* ------------------------------------------------------------------ */
/* Ring-buffer drain — replaces the per-sample loop of                   */
/* RingBufTracer.listenAndForwardRingBuffer (pkg/flow/tracer_ringbuf.go:  */
/* 112-134) over ringbuf.Reader / ringReader.readRecord                   */
/* (vendor/github.com/cilium/ebpf/ringbuf/ring.go:44-101): one bulk copy  */
/* of every committed sample into a staging buffer instead of one         */
/* reflection decode + channel send per record. Host-only, no device.     */
/* ------------------------------------------------------------------ */

/* A BPF_MAP_TYPE_RINGBUF as user space maps it (kernel/bpf/ringbuf.c):
 * data = the data pages (size mask+1; the second mapping cilium relies on is
 * not required), producer_pos / consumer_pos = the two position pages. */
typedef struct nfagg_ringbuf {
    const uint8_t* data;
    uint64_t mask;                        /* data size - 1, size a power of two */
    const volatile uint64_t* producer_pos;/* written by the kernel */
    volatile uint64_t* consumer_pos;      /* written by this call   */
} nfagg_ringbuf;

/* Copies committed 144-byte samples, in ring order, into dst (room for
 * cap_records; e.g. the buffer of nfagg_staging_acquire) until the ring is
 * empty, the next sample is still busy (ring.go:67-72), or dst is full; then
 * publishes the consumer position once. Discarded samples (ring.go:84-90) and
 * samples whose length is not 144 (model.ReadFrom would fail,
 * tracer_ringbuf.go:119-122) are skipped and counted in *n_skipped.
 * errno_counts (optional, 256 entries, ADDED to): records per metrics.errno,
 * for EvictedPacketsCounter("ringbuffer", errno) (tracer_ringbuf.go:128-130).
 * Returns NFAGG_OK, or NFAGG_EINVAL when the ring content is truncated
 * (io.ErrUnexpectedEOF in the reference); nothing is consumed past that point. */
int nfagg_ringbuf_drain(const nfagg_ringbuf* rb, void* dst, size_t cap_records,
                        size_t* n_records, size_t* n_skipped, uint64_t* errno_counts);

/* ------------------------------------------------------------------ */
/* Export encode — replaces, for evicted records, model.NewRecord's time */
/* and interface derivation (pkg/model/record.go:82-125), pbflow.FlowToPB */
/* / FlowsToPB (pkg/pbflow/proto.go:18-149), proto.Marshal of each        */
/* pbflow.Record (pkg/exporter/kafka_proto.go:53; gRPC marshalling of     */
/* pbflow.Records) and getFlowKey (pkg/exporter/kafka_proto.go:37-47).    */
/* ------------------------------------------------------------------ */

/* One row of the interface namer as a table: what
 * registerer.IfaceNameForIndexAndMAC (pkg/agent/interfaces_listener.go:74-80)
 * returns for (if_index, mac), and the UDN NewIntfDirUdn (record.go:167-183)
 * resolves for that name ("" = none). Lookup: the row with this index and MAC,
 * else the first row with this index and has_mac == 0, else unknown_name. */
typedef struct nfagg_intf_name {
    uint32_t if_index;
    uint8_t  mac[6];
    uint8_t  has_mac;
    uint8_t  name_len;           /* <= 16 */
    char     name[16];
    uint8_t  udn_len;            /* <= 63 */
    char     udn[63];
} nfagg_intf_name;               /* 92 bytes */

typedef struct nfagg_pb_options {
    uint32_t struct_size;        /* sizeof(nfagg_pb_options) */
    uint32_t n_names;
    int64_t  now_unix_ns;        /* currentTime   (account.go:103 c.clock())     */
    uint64_t mono_now_ns;        /* monotonicCurrentTime (account.go:104)        */
    uint8_t  agent_ip[16];       /* Record.AgentIP as a 16-byte net.IP           */
    const nfagg_intf_name* names;/* HOST memory, n_names rows (copied per call)  */
    char     unknown_name[16];   /* the namer's answer for an unknown interface  */
    uint8_t  unknown_len;
    uint8_t  pad_[7];
} nfagg_pb_options;

/* Serialise n evicted flow_record_t. Frame i = 0x0A varint(body_len[i]) body
 * starts at frame_offsets[i]; frame_offsets[n] = *out_bytes. Any run of frames
 * [a,b) is a serialized pbflow.Records{entries a..b-1}; the last body_len[i]
 * bytes of frame i are the serialized pbflow.Record (the Kafka message value).
 * kafka_keys (optional): n x 32 bytes, getFlowKey. Records carry only
 * BpfFlowMetrics (what Accounter evicts); the per-feature messages of the
 * MapTracer branch are not produced here. Returns NFAGG_TRUNCATED with
 * *out_bytes = bytes needed when out_cap is too small (nothing written).
 * All pointers HOST memory: */
int nfagg_encode_pb(nfagg_handle* h, const void* records, size_t n, const nfagg_pb_options* opt,
                    void* out, size_t out_cap, uint64_t* frame_offsets, uint32_t* body_len,
                    void* kafka_keys, size_t* out_bytes);
/* Same with d_records / d_out / d_frame_offsets / d_body_len / d_kafka_keys in
 * DEVICE memory (16-byte aligned), e.g. straight from nfagg_evict_device. */
int nfagg_encode_pb_device(nfagg_handle* h, const void* d_records, size_t n, const nfagg_pb_options* opt,
                           void* d_out, size_t out_cap, uint64_t* d_frame_offsets, uint32_t* d_body_len,
                           void* d_kafka_keys, size_t* out_bytes);

/* The MapTracer branch (pkg/flow/tracer_map.go:103-146): every flow is a full
 * model.BpfFlowContent (pkg/model/flow_content.go:9-17) — base metrics plus the
 * optional per-feature parts that LookupAndDeleteMap merged
 * (pkg/tracer/tracer.go:1057-1110; here: the `folded` outputs of nfagg_rollup_*).
 * Struct of arrays indexed like the records; part k is present for flow i (the Go
 * pointer is non-nil) when its array is non-NULL and present[i] has
 * NFAGG_FEAT_* set. Encoded per NewRecord (record.go:116-125: DNSLatency,
 * TimeFlowRtt) and FlowToPB (proto.go:79-118,129-138: dns_id/flags/errno/name
 * via utils.DNSRawNameToDotted, dns_latency only when non-zero, pkt_drop_*, xlat
 * with the address family of the FLOW's eth_protocol, ipsec_encrypted[_ret],
 * quic). Network events (NFAGG_FEAT_NETWORK_EVENTS) need the OVN sample decoder:
 * they are encoded as NewRecord does with a nil decoder (record.go:126) — field 27
 * empty, no drop injected; a caller with a decoder routes those flows through Go.
 * dns.name bytes are copied as they are (Go's Marshal rejects a non-UTF-8 string). */
typedef struct nfagg_pb_features {
    uint32_t struct_size;        /* sizeof(nfagg_pb_features) */
    uint32_t reserved_;
    const uint8_t* present;                      /* n bytes of NFAGG_FEAT_* bits (NULL: none) */
    const nfagg_additional_metrics* additional;  /* n entries or NULL */
    const nfagg_dns_metrics*        dns;
    const nfagg_pkt_drop_metrics*   drops;
    const nfagg_xlat_metrics*       xlat;
    const nfagg_quic_metrics*       quic;
} nfagg_pb_features;

/* nfagg_encode_pb over (records[i].id, BpfFlowContent{records[i].metrics, features[i]}).
 * Same outputs and return codes. All pointers HOST memory: */
int nfagg_encode_pb_content(nfagg_handle* h, const void* records, size_t n, const nfagg_pb_features* features,
                            const nfagg_pb_options* opt, void* out, size_t out_cap, uint64_t* frame_offsets,
                            uint32_t* body_len, void* kafka_keys, size_t* out_bytes);
/* Same with every data pointer (also those inside d_features) in DEVICE memory;
 * the nfagg_pb_features struct itself is in host memory. */
int nfagg_encode_pb_content_device(nfagg_handle* h, const void* d_records, size_t n, const nfagg_pb_features* d_features,
                                   const nfagg_pb_options* opt, void* d_out, size_t out_cap, uint64_t* d_frame_offsets,
                                   uint32_t* d_body_len, void* d_kafka_keys, size_t* out_bytes);

/* ------------------------------------------------------------------ */
/* Sharding, stats, sync                                                */
/* ------------------------------------------------------------------ */

/* Shard of a flow key: the function that routes records to GPUs
 * (hash of the 40 key bytes with byte 39 forced to 0). */
uint32_t nfagg_shard_of(const nfagg_flow_id* id, uint32_t n_shards);
/* Host-side router: shard id of each of n 144-byte records (HOST memory), for a
 * caller that feeds one handle per GPU. Pure CPU; no device needed. */
void nfagg_shard_ids(const void* records, size_t n, uint32_t n_shards, uint32_t* out_shard);
/* The 64-bit key hash itself (table index / fingerprint / shard all derive from it). */
uint64_t nfagg_key_hash(const nfagg_flow_id* id);
/* The 64-bit hash of a 16-byte IP with the given seed index (0..3), as used by
 * the sketches. */
uint64_t nfagg_ip_hash(const uint8_t ip[16], uint32_t seed_index);

int nfagg_stats_get(nfagg_handle* h, nfagg_stats* out);
int nfagg_stats_reset_profile(nfagg_handle* h);
/* Wait until all submitted work of this handle has finished. */
int nfagg_sync(nfagg_handle* h);
/* The hipStream_t the handle launches on (as void*), for event timing by the caller. */
void* nfagg_stream(nfagg_handle* h);
/* ------------------------------------------------------------------ */
/* Multi-GPU group — N devices behind ONE process.                      */
/* The agent is one process with one pipeline (pkg/agent/agent.go:387-442;  */
/* the Accounter is built at agent.go:208-212), so its N GPUs are driven   */
/* from that process: flows shard by key hash (nfagg_shard_of), member i    */
/* is an ordinary handle owning shard i. A batch enters on one member's     */
/* device, is partitioned there in arrival order (stable device partition)  */
/* and the buckets go to their owners over xGMI; every owner folds its      */
/* bucket. Flow state needs no collective; the Count-Min / HyperLogLog      */
/* arrays are all-reduced with RCCL at the eviction tick (librccl is        */
/* loaded on demand; a single-GPU process never loads it).                  */
/* ------------------------------------------------------------------ */
typedef struct nfagg_group nfagg_group;

/* cfg as for nfagg_create, except: device / n_shards / shard_id / ext_sketch are set per member, and max_entries is
 * CACHE_MAX_FLOWS for the whole group — every shard holds at most ceil(max_entries / n_devices) flows.
 * devices: HIP ordinals, one per member; all distinct (production), or all equal (several members on ONE GPU: lets a
 * single-GPU box exercise partition, routing and the stop-on-full logic; the sketch merge then runs as local kernels). */
int nfagg_group_create(const nfagg_config* cfg, const int32_t* devices, uint32_t n_devices, nfagg_group** out);
void nfagg_group_destroy(nfagg_group* g);
const char* nfagg_group_last_error(const nfagg_group* g);   /* NULL: last create error */
uint32_t nfagg_group_size(const nfagg_group* g);
/* Member i (shard i): sketch queries, stats, device-side export (nfagg_encode_pb_device ...) go through its handle.
 * Do not ingest into or evict a member directly. */
nfagg_handle* nfagg_group_member(nfagg_group* g, uint32_t i);

/* The record arm of Accounter.Account (account.go:81-96) for the group: records in HOST memory, in arrival order.
 * Chunks go up the members' PCIe links in turn (pinned staging ring of the member), are partitioned on that device
 * and routed. Returns NFAGG_FULL with *consumed = the leading records folded when the next record's NEW key finds
 * its shard full (account.go:85): evict the group with NFAGG_REASON_FULL, then resubmit the rest. */
int nfagg_group_ingest(nfagg_group* g, const void* records, size_t n, size_t* consumed);
/* Same, records already in DEVICE memory of member `src_member`'s device (16-byte aligned, < 2^31 records). In local-fold
 * mode the chunk is folded by that member, asynchronously: the buffer must stay valid until the group synchronises
 * (nfagg_group_len, nfagg_group_evict*), as for nfagg_ingest_device.
 * THREADS: distinct source members may be fed concurrently, one host thread per source member (how N PCIe links are kept
 * busy from one process). Routed mode partitions every chunk on its source's own stream — the partitions of concurrent calls
 * overlap — and folds the buckets one call at a time (arrival order between concurrent calls = the order in which they get
 * there); local-fold mode reserves the chunk's sequence numbers and folds concurrently. Every other group call needs the
 * ingest threads to have returned. */
int nfagg_group_ingest_device(nfagg_group* g, uint32_t src_member, const void* d_records, size_t n, size_t* consumed);
/* len(c.entries) over all shards. */
int nfagg_group_len(nfagg_group* g, uint64_t* entries);
/* The per-tick collective: ncclAllReduce(sum, uint64) over each Count-Min array and ncclAllReduce(max, uint8) over each
 * HLL register array, in place on every member, on the members' streams. Afterwards every member answers
 * nfagg_hll_estimate / nfagg_cm_query / nfagg_cm_topk for the whole node. IN PLACE means: call it ONCE per window, then
 * nfagg_sketch_reset every member (nfagg_group_member) before the next window's records arrive — a second call, or the next
 * tick's call without the reset, would sum N copies of the already merged counters (Count-Min inflated N x per call). */
int nfagg_group_merge_sketches(nfagg_group* g);
/* Accounter.evict (account.go:102-124) for every shard: the members' flows back to back in `out` (HOST memory). */
int nfagg_group_evict(nfagg_group* g, int reason, void* out, size_t cap, size_t* n_out);
/* Same, shard i's flows into d_out[i] (DEVICE memory of member i's device, cap[i] records, 16-byte aligned), n_out[i] of
 * them: the input of nfagg_encode_pb_device on member i. NFAGG_TRUNCATED (nothing evicted, n_out = sizes needed) when a
 * buffer is too small. */
int nfagg_group_evict_device(nfagg_group* g, int reason, void* const* d_out, const size_t* cap, size_t* n_out);

/* ------------------------------------------------------------------ */
/* Local fold across GPUs with ONE PROCESS PER GPU (ranks of a           */
/* torch.distributed / MPI job; the in-process form is                   */
/* NFAGG_GROUP_LOCAL_FOLD above). Every rank owns an unsharded handle     */
/* (n_shards = 1) and folds the part of the ONE record stream that        */
/* arrives at it, whatever its keys — no per-record routing. Sequence     */
/* numbers must be global to the job: before folding a chunk the rank     */
/* tells the handle the arrival position of its first record              */
/* (nfagg_set_sequence). At the eviction tick a table's slots are         */
/* mergeable partials of their flows (sums, ORs, maxima, sequence-tagged  */
/* words where the earlier / later record wins):                          */
/*   1. nfagg_partials_export_device  the live flows as 192-byte          */
/*      partials grouped by owner, owner = nfagg_shard_of(key, n_shards)  */
/*   2. the caller moves segment o to rank o (RCCL all-to-all over xGMI,  */
/*      hipMemcpyPeerAsync, ...)                                          */
/*   3. nfagg_partials_merge_device   rank o merges what it received      */
/*   4. nfagg_evict_owned_device      rank o evicts the flows it owns;    */
/*      everything else in its table expires with the epoch.              */
/* The union of the ranks' evictions is bit-identical to ONE sequential   */
/* Accounter (pkg/flow/account.go:58-124) over the records folded since   */
/* the last eviction, in the order of their sequence numbers.             */
/* NFAGG_MODE_KERNEL_DEDUP (bpf/flows.c:76-143): the handle must have been  */
/* created with nfagg_config.local_fold = 1; its partials are SUB-FLOWS      */
/* (flow, interface) of NFAGG_PARTIAL_BYTES_DEDUP bytes, owned by the owner  */
/* of their FLOW; step 4 joins the sub-flows of each flow (first interface = */
/* interface of the earliest record in the job) and the union of the ranks'  */
/* evictions is bit-identical to ONE kernel-dedup table over those records.  */
/* ------------------------------------------------------------------ */
#define NFAGG_PARTIAL_BYTES 192u
#define NFAGG_PARTIAL_BYTES_DEDUP 256u
/* Bytes per partial on this handle: NFAGG_PARTIAL_BYTES, or NFAGG_PARTIAL_BYTES_DEDUP in kernel-dedup mode. */
size_t nfagg_partial_bytes(const nfagg_handle* h);
#define NFAGG_SHARD_NONE 0xFFFFFFFFu

/* The next record folded by this handle carries sequence number next_seq (epoch-relative: every eviction restarts the
 * epoch at 0; 64 bits). Must not be smaller than the number the handle has reached. Gaps are harmless: only the order
 * matters. Marks the handle as sharing its numbering with other tables (see nfagg_window_restart_device). */
int nfagg_set_sequence(nfagg_handle* h, uint64_t next_seq);

/* Step 1. d_out: DEVICE memory, 64-byte aligned, room for `cap` partials of nfagg_partial_bytes(h). Segment o (the flows shard
 * o owns) starts at partial sum(counts[0..o)) and holds counts[o] partials; counts: HOST array of n_shards (<= 64) words.
 * The flows of self_shard stay in the table and are not exported (counts[self_shard] = 0); NFAGG_SHARD_NONE exports all.
 * *n_out = partials written. NFAGG_TRUNCATED (nothing written, nothing changed, *n_out = partials needed; an upper bound
 * known in advance is nfagg_len) when cap is too small. Synchronous: the partials are complete when the call returns.
 * Afterwards the handle accepts no records (nfagg_ingest* return NFAGG_FULL) until nfagg_evict_owned_device ran: the
 * exported flows still sit in the table and would be exported twice. */
int nfagg_partials_export_device(nfagg_handle* h, uint32_t n_shards, uint32_t self_shard, void* d_out, size_t cap,
                                 uint64_t* counts, size_t* n_out);
/* Step 3. d_partials: n partials in DEVICE memory of this handle's device (16-byte aligned), all owned by shard_id of
 * n_shards (a partial of another shard fails the next synchronising call). Asynchronous on the handle's stream: the buffer
 * must stay valid until the handle synchronises. May be called several times (one call per source). The table needs room
 * for the flows it receives: size it with table_log2_slots (about 4 slots per max_entries, as the group does). */
int nfagg_partials_merge_device(nfagg_handle* h, uint32_t n_shards, uint32_t shard_id, const void* d_partials, size_t n);
/* Step 4. nfagg_evict_device restricted to the flows shard_id of n_shards owns; ends the epoch of the whole table.
 * NFAGG_TRUNCATED (nothing evicted, *n_out = records needed) when cap is too small. (Kernel-dedup mode: *n_out counts FLOWS,
 * which only the join of the sub-flows tells: call with cap = 0 first, or keep cap >= nfagg_len.) */
int nfagg_evict_owned_device(nfagg_handle* h, int reason, uint32_t n_shards, uint32_t shard_id, void* d_out, size_t cap,
                             size_t* n_out);

/* The sequence window of ranks that share one numbering. The slots carry sequence numbers relative to a 32-bit window; a
 * handle that alone holds its flows moves the window by itself when ~2^32 records of an epoch have gone by. Ranks of a
 * local-fold job cannot (the order between the ranks' partials of one flow would be lost): nfagg_ingest* fails with
 * NFAGG_ERANGE on such a handle (one that had nfagg_set_sequence) when its window is used up. Before that happens — every
 * rank knows the job's position — all ranks bring the flows together at their owners WITHOUT evicting:
 *   nfagg_partials_export_device(h, n_shards, NFAGG_SHARD_NONE, ...)   every flow, the rank's own included
 *   exchange: segment o to rank o (the own segment stays)
 *   nfagg_window_restart_device(h, n_shards, shard_id, d_partials, n, next_seq)     (NFAGG_MODE_ACCOUNTER only)
 * which empties the table (the epoch tag; nothing is written), merges the n partials this rank owns and rebases their tags;
 * the next record folded carries next_seq (>= every number used in the job so far). The epoch goes on: sketches untouched,
 * nfagg_len = the flows this rank owns. The in-process group (NFAGG_GROUP_LOCAL_FOLD) does all of this by itself. */
int nfagg_window_restart_device(nfagg_handle* h, uint32_t n_shards, uint32_t shard_id, const void* d_partials, size_t n,
                                uint64_t next_seq);

/* Testing aid: account for `records` more records in the current eviction epoch without folding any (their sequence
 * numbers are skipped), so that the moves of the 32-bit sequence window (every ~2^32 records) can be exercised without
 * feeding 600 GB. */
int nfagg_debug_skip_sequence(nfagg_handle* h, uint64_t records);
int nfagg_group_debug_skip_sequence(nfagg_group* g, uint64_t records);   /* the group's common position (local fold) / every member's */

#ifdef __cplusplus
}
#endif
#endif /* NFAGG_H */
