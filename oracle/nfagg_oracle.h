/*
 * nfagg_oracle.h — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the reference's userspace flow
 * aggregation path, used ONLY by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker / reported baseline. Nothing under
 * netobserv-ebpf-agent_amd/ (the product) includes, links or calls this.
 *
 * Pinning status:
 *   PINNED by the reference's own known-answer tests (encoded in
 *   tests/test_oracle_kat.py): record decode (pkg/model/record_test.go:19-102,
 *   193-347), Accounter (pkg/flow/account_test.go:47-217), AccumulateDNS/Drops/
 *   NetworkEvents/Xlat/Additional/Quic and base-from-additional
 *   (pkg/model/flow_content_test.go:11-380).
 *   PINNED to the reference's own C compiled in place (oracle/_ref, built by
 *   `make -C oracle ref` from /root/reference/bpf/flows.c:76-143 + bpf/types.h;
 *   tests/test_oracle_ref.py): the kernel dedup merge (mode 1).
 *   PARITY UNPINNED (no reference test exercises them; the oracle follows the
 *   source text): AccumulateBase's order-dependent fields (eth_protocol/dscp/
 *   sampling last-non-zero, MAC first-non-zero), u32/u64 wrap-around, and the
 *   Count-Min / HyperLogLog sketches (which do not exist in the reference at
 *   all — our own spec).
 *
 * The Go toolchain is absent, so the Go side of the reference cannot be built
 * here; tools/go/accounter_parity_test.go dumps the reference Accounter's
 * evictions on the seeded streams for anyone with a toolchain.
 */
#ifndef NFAGG_ORACLE_H
#define NFAGG_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- record ABI, restated from bpf/types.h (NOT shared with include/nfagg.h
 * on purpose: tests cross-check sizes and offsets of the two) ---- */
typedef struct {            /* bpf/types.h:191-204 */
    uint8_t  src_ip[16], dst_ip[16];
    uint16_t src_port, dst_port;
    uint8_t  proto, icmp_type, icmp_code, pad;
} orc_flow_id;

typedef struct {            /* bpf/types.h:94-126 */
    uint64_t start, end, bytes;
    uint32_t packets;
    uint16_t eth_protocol, flags;
    uint8_t  src_mac[6], dst_mac[6];
    uint32_t if_index_first_seen, lock, sampling;
    uint8_t  direction_first_seen, err_no, dscp, nb_observed_intf;
    uint8_t  observed_direction[6];
    uint8_t  pad2[2];
    uint32_t observed_intf[6];
    uint16_t ssl_version, tls_cipher_suite, tls_key_share;
    uint8_t  tls_types, misc_flags;
    uint8_t  pad4[4];
} orc_flow_metrics;

typedef struct { orc_flow_id id; orc_flow_metrics metrics; } orc_flow_record; /* :212-215 */

typedef struct { uint64_t start, end, flow_rtt; int32_t ipsec_ret; uint16_t eth_protocol;
                 uint8_t ipsec_encrypted, pad; } orc_additional;              /* :174-181 */
typedef struct { uint64_t start, end, latency; uint16_t id, flags, eth_protocol;
                 uint8_t err_no; char name[32]; uint8_t pad; } orc_dns;       /* :131-140 */
typedef struct { uint64_t start, end; uint16_t bytes, packets; uint32_t latest_drop_cause;
                 uint16_t latest_flags, eth_protocol; uint8_t latest_state, pad[3]; } orc_drops; /* :142-151 */
typedef struct { uint64_t start, end; uint8_t network_events[4][8]; uint16_t bytes[4], packets[4];
                 uint16_t eth_protocol; uint8_t network_events_idx, pad[5]; } orc_netev;        /* :153-161 */
typedef struct { uint64_t start, end; uint8_t saddr[16], daddr[16];
                 uint16_t sport, dport, zone_id, eth_protocol; } orc_xlat;    /* :163-172 */
typedef struct { uint64_t start, end; uint32_t version; uint16_t eth_protocol;
                 uint8_t seen_long_hdr, seen_short_hdr; } orc_quic;           /* quic_metrics_t */

/* model.BpfFlowContent (pkg/model/flow_content.go:9-17): base + optional parts.
 * has_* mirrors the nil-ness of the Go pointers. */
typedef struct {
    orc_flow_metrics base;
    int has_dns, has_drops, has_netev, has_xlat, has_additional, has_quic;
    orc_dns dns; orc_drops drops; orc_netev netev; orc_xlat xlat;
    orc_additional additional; orc_quic quic;
} orc_content;

/* ---- pkg/model/flow_content.go ---- */
void orc_accumulate_base(orc_flow_metrics* p, const orc_flow_metrics* other);      /* :28-61  */
void orc_accumulate_dns(orc_content* p, const orc_dns* other);                     /* :76-96  */
void orc_accumulate_drops(orc_content* p, const orc_drops* other);                 /* :98-118 */
void orc_accumulate_netev(orc_content* p, const orc_netev* other);                 /* :120-137 */
void orc_accumulate_xlat(orc_content* p, const orc_xlat* other);                   /* :139-152 */
void orc_accumulate_additional(orc_content* p, const orc_additional* other);       /* :154-177 */
void orc_accumulate_quic(orc_content* p, const orc_quic* other);                   /* :179-198 */
uint16_t orc_add_uint16(uint16_t a, uint16_t b);                                   /* :209-215 */

/* ---- pkg/tracer/tracer.go:1118-1146 per-CPU fold (one kind at a time) ----
 * kind: 0 additional, 1 dns, 2 drops, 3 netev, 4 xlat, 5 quic.
 * partials: n_flows*n_cpu structs flow-major; base in/out; folded out. */
void orc_rollup(int kind, const void* partials, size_t n_flows, size_t n_cpu,
                orc_flow_metrics* base, void* folded);

/* ---- pkg/tracer/tracer.go:1022-1146 LookupAndDeleteMap: merge of the drained maps (nfagg_oracle_maps.c).
 * feat_*[k]: kind k as in orc_rollup (NULL / 0 = map not enabled); feat_vals[k] holds feat_n[k]*n_cpu
 * partials, flow-major. out_ids/out: room for n_main + sum(feat_n) entries; returns the number of flows,
 * sorted by key bytes. */
size_t orc_map_merge(const orc_flow_id* main_ids, const orc_flow_metrics* main_vals, size_t n_main,
                     const orc_flow_id* const feat_ids[6], const void* const feat_vals[6], const size_t feat_n[6],
                     size_t n_cpu, orc_flow_id* out_ids, orc_content* out);

/* ---- pkg/flow/account.go ---- */
typedef struct orc_accounter orc_accounter;
/* mode 0: Accounter (AccumulateBase). mode 1: kernel dedup merge (bpf/flows.c:76-143). */
orc_accounter* orc_acc_new(uint64_t max_entries, int mode);
void   orc_acc_free(orc_accounter*);
/* account.go:81-96 for records[0..n): stops BEFORE the record whose new key
 * finds len(entries) >= max_entries (:85) and returns how many it consumed
 * (n when it never stopped). The caller then evicts ("full") and resubmits. */
size_t orc_acc_ingest(orc_accounter*, const void* records, size_t n);
size_t orc_acc_len(const orc_accounter*);
/* multi-core baseline helper (not a reference path): fold only the records of one key-hash shard; returns how many */
size_t orc_acc_ingest_shard(orc_accounter*, const void* records, size_t n, uint32_t n_shards, uint32_t shard);
/* nfagg_oracle_mt.c: partition-then-fold on T threads (bench.py cpu_baseline.multicore); seconds[0] partition, [1] fold */
size_t orc_partition_fold_mt(const void* records, size_t n, uint32_t T, uint64_t max_entries, int mode, size_t* flows, double seconds[3]);
size_t orc_local_fold_mt(const void* records, size_t n, uint32_t T, uint64_t max_entries, size_t* flows, double seconds[3], void* out, size_t out_cap);
void orc_mt_set_pinning(int on);   /* 1: thread t of orc_local_fold_mt binds to the t-th CPU in NUMA-node order (bench.py's cpu_baseline) */
/* account.go:102-124 up to NewRecord: all entries as 144-byte records sorted by
 * the 40 key bytes (memcmp), table cleared. Returns the count (writes at most cap). */
size_t orc_acc_evict(orc_accounter*, void* out, size_t cap);

/* pkg/model/record.go:90-97 */
void orc_record_times(int64_t now_unix_ns, uint64_t mono_now, const orc_flow_metrics* m,
                      int64_t* start_unix_ns, int64_t* end_unix_ns);

/* ---- sketches: our own spec. PARITY UNPINNED. ---- */
uint64_t orc_key_hash(const void* key40);                 /* byte 39 treated as 0 */
uint64_t orc_ip_hash(const uint8_t ip[16], uint32_t seed_index);
uint32_t orc_shard_of(const void* key40, uint32_t n_shards);
/* cm: uint64[depth<<log2w]; adds `add` to one counter per row */
void   orc_cm_update(uint64_t* cm, uint32_t depth, uint32_t log2w, const uint8_t ip[16], uint64_t add);
uint64_t orc_cm_query(const uint64_t* cm, uint32_t depth, uint32_t log2w, const uint8_t ip[16]);
/* the k heaviest distinct src/dst addresses of `records` by Count-Min estimate (estimate desc, address bytes asc);
 * out: k rows of {uint8_t ip[16]; uint64_t estimate}; returns the number of rows */
size_t orc_cm_topk(const uint64_t* cm, uint32_t depth, uint32_t log2w, const void* records, size_t n, int side, size_t k, void* out);
void   orc_hll_update(uint8_t* regs, uint32_t p, const uint8_t ip[16]);
/* scalar HLL estimate straight from the registers, register order */
double orc_hll_estimate(const uint8_t* regs, uint32_t p);
/* feed records into sketches: cm_src/cm_dst add metrics.bytes; hll_src/hll_dst. NULL = skip */
void   orc_sketch_ingest(const void* records, size_t n, uint64_t* cm_src, uint64_t* cm_dst,
                         uint32_t depth, uint32_t log2w, uint8_t* hll_src, uint8_t* hll_dst, uint32_t p);

/* ---- synthetic streams; formulas from
 * pkg/model/bench_fixtures_test.go:19-50 ---- */
/* population member i -> key */
void orc_bench_flow_id(uint64_t i, orc_flow_id* out);
/* record j of the stream, drawn for population member i */
void orc_bench_record(uint64_t i, uint64_t j, orc_flow_record* out);
/* Zipf(s) rank table: thresholds[k] = floor(2^64 * CDF(k)) (last = 2^64-1). */
void orc_zipf_thresholds(uint64_t n_keys, double s, uint64_t* thresholds);
/* counter-based RNG shared with the device generator */
uint64_t orc_splitmix64(uint64_t x);
/* stream record j -> population index: hot_permille of records go to key 0, the
 * rest Zipf over the table (hot_permille = 0 for plain Zipf). uniform if thresholds==NULL. */
uint64_t orc_stream_key_index(uint64_t seed, uint64_t j, uint64_t n_keys,
                              const uint64_t* thresholds, uint32_t hot_permille);
/* fill out[0..n) with records j0..j0+n of the stream. variant 0: bench fixture
 * metrics; variant 1: every order-dependent field scrambled per record.
 * pop_index (optional, n_keys entries) maps a drawn rank to a population member
 * (used to restrict a stream to the members of one hash shard). */
void orc_gen_stream(uint64_t seed, uint64_t j0, size_t n, uint64_t n_keys,
                    const uint64_t* thresholds, uint32_t hot_permille, uint32_t variant,
                    const uint64_t* pop_index, void* out);

/* ---- record -> protobuf (oracle/nfagg_oracle_pb.c): pkg/pbflow/proto.go:40-149 FlowToPB over the
 * model.Record that model.NewRecord (pkg/model/record.go:82-125) builds from one evicted
 * flow_record_t, serialised as google.golang.org/protobuf does (fields in number order, proto3
 * zero-value omission, message fields present whenever the Go pointer is non-nil). ---- */
typedef struct {            /* one row of the interface namer table (registerer.IfaceNameForIndexAndMAC) */
    uint32_t if_index;
    uint8_t  mac[6];
    uint8_t  has_mac;       /* 0: matches any MAC */
    uint8_t  name_len;
    char     name[16];
    uint8_t  udn_len;       /* the value NewIntfDirUdn (record.go:167-183) resolves for this name */
    char     udn[63];
} orc_intf_name;            /* 92 bytes */
typedef struct {
    int64_t  now_unix_ns;   /* currentTime */
    uint64_t mono_now_ns;   /* monotonicCurrentTime */
    uint8_t  agent_ip[16];  /* net.IP in 16-byte form */
    const orc_intf_name* names;
    uint32_t n_names;
    char     unknown_name[16];  /* what the namer returns for an unknown interface ("unknown", interfaces_listener.go:77) */
    uint8_t  unknown_len;
} orc_pb_options;
/* Serialise pbflow.Record for one evicted flow into out (>= 1024 bytes); returns the length. */
size_t orc_pb_encode_record(const orc_flow_record* r, const orc_pb_options* o, uint8_t* out);
/* Same for the MapTracer branch: the full BpfFlowContent (DNS, drops, xlat, RTT/IPsec, QUIC;
 * network events need the OVN sample decoder and are encoded as with a nil decoder, record.go:126). */
size_t orc_pb_encode_content(const orc_flow_id* id, const orc_content* c, const orc_pb_options* o, uint8_t* out);
/* pkg/utils/utils.go:18-58 DNSRawNameToDotted over the 32-byte kernel copy; out >= 32 bytes; returns the length */
size_t orc_dns_name_dotted(const char raw[32], char* out);
/* pkg/exporter/kafka_proto.go:37-47 getFlowKey: the two IPs, smaller first */
void orc_kafka_key(const orc_flow_record* r, uint8_t out[32]);

#ifdef __cplusplus
}
#endif
#endif
