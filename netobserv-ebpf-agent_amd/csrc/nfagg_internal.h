// nfagg_internal.h — device-side layout of the flow table and the launch
// interface between the C ABI (nfagg_api.hip) and the kernels.
//
// The table replaces Accounter.entries (pkg/flow/account.go:22,
// map[BpfFlowId]*BpfFlowMetrics): open addressing, linear probing, one
// 128-byte "hot" line per slot (tag + key + every field a record updates) and
// one 128-byte "cold" line (fields that only the first record of a flow, or
// the first record with a non-zero MAC, ever writes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nfagg_hash.h"
#include "../../include/nfagg.h"

namespace nfagg {

constexpr int kRecordBytes = 144;   // bpf/types.h:212-215
constexpr int kRecordDwords = 36;
constexpr int kKeyBytes = 40;       // bpf/types.h:191-204

// All 64-bit words of a slot are accessed with agent-scope atomics while an
// ingest kernel runs (per-XCD L2s are not coherent with each other; see
// DESIGN.md §coherence). Identities are zero. A slot belongs to the eviction epoch
// its tag names (bits 48..63): eviction does not touch the table at all, it bumps
// TableView.epoch_bits, and whoever claims a slot with an older tag zeroes its
// value words before publishing it. Sequence numbers are epoch-relative and fit 32
// bits (an epoch that reaches 2^32-16 records ends in an eviction).
//
// "Tagged word": (~seq32) << 32 | data32, combined with atomic max: the word of
// the record with the SMALLEST sequence number wins, independently per word, so
// a group of tagged words written by the same records ends up holding the
// earliest record's data with no lock (all words see the same set of tags).
struct alignas(128) SlotHot {
    uint64_t tag;        // epoch<<48 | fp46<<2 | state: 2 claimed, key and zeroes being written; 3 ready. Another epoch = free
    uint64_t key[5];     // flow_id, byte 39 zero
    uint64_t bytes;      // sum, wraps (flow_content.go:42)
    uint64_t end;        // max          (:39-41)
    uint64_t start_inv;  // max of ~start over non-zero starts; 0 = unset (:36-38)
    uint32_t packets;    // sum, wraps (:43)
    uint32_t flags;      // OR (:44), low 16 bits
    uint64_t eth_tag;    // max of (seq+1)<<16 | eth_protocol  over non-zero eth   (:45-47 last non-zero)
    uint64_t dscp_tag;   // max of (seq+1)<<8  | dscp          over non-zero dscp  (:54-56)
    uint64_t samp_tag;   // max of (seq+1)<<32 | sampling      over non-zero sampling (:57-59)
    uint64_t id0;        // tagged: record dword 21 (if_index_first_seen) of the FIRST record (account.go:95);
                         // its tag is the flow's first sequence number
    uint64_t smac_lo;    // tagged: low 32 bits of the first non-zero src_mac (:48-50)
    uint64_t dmac_lo;    // tagged: low 32 bits of the first non-zero dst_mac (:51-53)
};
static_assert(sizeof(SlotHot) == 128, "hot line");

// Half a line of "first record" data per slot. The MAC halves are tagged words like their low halves in the hot line
// (several workgroups may race for "first non-zero MAC"). The identity dwords are PLAIN: no fold kernel writes them;
// k_finalize (nfagg_kernels.hip), the last launch of every ingest call, copies them from the batch record that the
// slot's id0 tag names as the flow's first (the batch is still in HBM: the caller owns it until the call returns).
struct alignas(64) SlotCold {
    uint64_t smac_hi;    // tagged: high 16 bits of the first non-zero src_mac
    uint64_t dmac_hi;
    uint32_t id[12];     // first record's dwords 22, 24..34 (23 = sampling comes from samp_tag, 35 = pad4 is zero)
};
static_assert(sizeof(SlotCold) == 64, "cold half line");

// Kernel-dedup mode only (NFAGG_MODE_KERNEL_DEDUP, bpf/flows.c:76-143): two more
// lines per slot. In this mode the hot line's start_inv holds the FIRST record's raw
// start, eth_tag its raw eth_protocol (update_existing_flow never touches them),
// `end` is unused, flags carries tls_types in bits 16..23, and dscp/sampling tags
// are fed by every counted record (zero values included: "last", not "last non-zero").
//
// cand[]: the (up to) seven interfaces with the earliest first appearance in the
// epoch, one word (~seq)<<32 | if_index each, maintained by a lock-free CAS protocol
// (nfagg_dedup.hip topk_insert) — the flow's own if_index_first_seen is always one
// of them, which leaves the six that observed_intf[] can take (flows.c:79).
// dir[j][]: for candidate j the two earliest DISTINCT direction values,
// (~seq)<<8 | direction — all add_observed_intf needs to replay the merge to
// OBSERVED_DIRECTION_BOTH (flows.c:84-88) in arrival order at eviction.
struct alignas(128) SlotAux {
    uint64_t cand[7];
    uint64_t endl_lo;     // max of (seq+1)<<32 | low  32 bits of end : end is ASSIGNED by the last record (flows.c:108,127)
    uint64_t endl_hi;     // max of (seq+1)<<32 | high 32 bits
    uint64_t ssl_first;   // tagged: first non-zero ssl_version among the counted records (flows.c:112-119)
    uint32_t ssl_max;     // max / min over the non-zero ssl versions: they differ <=> MISC_FLAGS_SSL_MISMATCH
    uint32_t ssl_minv;    // max of 0x10000 - ssl (0 = none seen)
    uint64_t cs_tag;      // max of (seq+1)<<16 | tls_cipher_suite over server hellos (flows.c:120-122)
    uint64_t ks_tag;      // same for tls_key_share (flows.c:123-125)
    uint64_t dir[7][2];
    uint64_t pad[5];
};
static_assert(sizeof(SlotAux) == 256, "aux lines");

// Device-resident counters, mirrored to pinned host memory on demand.
struct DevCounters {
    unsigned long long n_live;     // claimed slots this epoch (== len(c.entries) unless a split is pending)
    unsigned long long n_out;      // records written by the evict kernel
    unsigned long long n_skipped;  // records of other shards
    unsigned long long n_bypassed; // records merged one by one (no LDS cache entry for their flow)
    unsigned int error;            // non-zero: a kernel bailed out (probe overflow)
    unsigned int max_probe;
    unsigned long long n_direct;   // two-pass ingest: records merged one by one in pass 2 / pass 3 (no LDS entry, queue overflow)
    unsigned long long n_finalized;// live_list[0..n_finalized) have their identity dwords written (k_finalize)
    unsigned int aborted;          // a claim was refused because n_live reached TableView.claim_limit: the fold of this
                                   // batch is incomplete and the API rolls it back (optimistic fold, nfagg_api.hip)
    unsigned int fin_ticket;       // k_finalize: blocks done (the last one publishes n_finalized and resets this)
    unsigned long long phase[8];   // diagnostic builds only: per-phase wave-cycle sums
};

// Two-pass partitioned ingest (nfagg_ingest_part.hip): per-partition queues of spilled record indices.
constexpr int kSpillParts = 2048;
struct SpillView {
    uint32_t* queue;               // kSpillParts x qcap record indices (0xffffffff = padding)
    uint32_t* qtail;               // kSpillParts reserved-entry counters; zero between launches
    uint32_t qcap;                 // entries per partition, multiple of 4
    uint32_t* ovf;                 // overflow list (partition queue or staging group full): groups of 4 and single items, read item by item
    uint32_t* ovf_tail;
    uint32_t ovf_cap;
    unsigned int* error;           // = &DevCounters.error
    uint32_t part_shift;           // partition of a key = (hash >> part_shift) & (n_parts - 1); see nfagg_create
    uint32_t n_parts;              // partitions in use (power of two, <= kSpillParts): the two-pass accounter fold scales them to the
                                   // batch (launch_ingest_part), the kernel-dedup passes always use kSpillParts
    uint4* xp;                     // kernel-dedup mode: kDedupXpEntries exported cache entries of 144 bytes (nfagg_dedup_cached.hip)
    uint32_t sort_first;           // kernel-dedup partition pass: many flows per partition expected (the API's guess from the last epoch):
                                   // sort the items by sub-partition before the first round (both ways are exact)
};
constexpr uint64_t kDedupXpEntries = 256ull * 1024ull;   // streaming workgroups x their cache entries
constexpr uint64_t kDedupXpBytes = kDedupXpEntries * 144ull;

struct TableView {
    SlotHot* hot;
    SlotCold* cold;
    SlotAux* aux;                  // non-null only in kernel-dedup mode
    uint32_t* live_list;           // slot indices claimed this epoch, in claim order
    DevCounters* ctr;
    uint64_t mask;                 // slots - 1
    uint64_t claim_limit;          // hard bound on claimed slots per epoch (3/4 of the table): claims beyond it are refused
    uint64_t epoch_bits;           // current eviction epoch (1..65535) << 48: the tags of this epoch's slots carry it
    uint32_t n_shards, shard_id;
    uint32_t defer_claims;         // 1: tables of 2^21 slots or more (see nfagg_create)
    uint32_t subflow;              // kernel-dedup mode of a local-fold rank: the table is keyed by the SUB-FLOW (flow key,
                                   // if_index_first_seen); SlotHot.end holds the sixth key word (nfagg_dedup.h)
    SpillView spill;               // set by the API for the two-pass fold
};

struct SketchView {
    uint64_t* cm[2];               // src, dst : depth << log2w counters
    uint8_t* hll[2];               // src, dst : 1 << p registers of ONE BYTE each (the spec's and the collective's layout; the buffers
                                   // are 4-byte aligned: a register is raised by a CAS on the word that holds it)
    uint32_t cm_depth, cm_log2w, hll_p, flags;
};

// ---- launch wrappers (defined in nfagg_kernels.hip) ----
// Fold records[0..n) into the table; record i carries sequence seq_base + i.
// variant: 0 = default, see DESIGN.md.
// When ingest_fuses_sketches(variant), the kernel also applies the sketch updates of sk
// (pass sk.flags = 0 to disable); otherwise the caller launches launch_sketch_update itself.
hipError_t launch_ingest(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, uint64_t seq_base,
                         int mode, int variant, hipStream_t s);
bool ingest_fuses_sketches(int mode, int variant, uint64_t n, uint32_t sketch_flags);
bool ingest_variant_supported(int variant);
bool ingest_needs_spill(int mode, int variant, uint64_t n);
hipError_t launch_ingest_part(const TableView& t, const SketchView& sk, const SpillView& q, const void* d_records, uint64_t n,
                              uint64_t seq_base, int variant, hipStream_t s);
// nfagg_epoch_par.hip / nfagg_account_par.inc: the evict-on-full loop of nfagg_account with its epochs found first — sort keys
// ((top 40 hash bits) << 24 | index), their sort, previous-occurrence links, live flags, the cut walk (resumable: run in parts);
// then the complete epochs of the
// middle: positions and segment folds (no table)
hipError_t launch_par_hash(const void* d_records, uint64_t n, uint64_t* d_keys, hipStream_t s);
hipError_t launch_par_sort(void* temp, size_t* temp_bytes, const uint64_t* k_in, uint64_t* k_out, uint64_t n, hipStream_t s);
hipError_t launch_par_links(const void* d_records, const uint64_t* d_keys_sorted, uint64_t n, int32_t* d_prev, uint32_t* d_overflow, hipStream_t s);
hipError_t launch_par_live(const TableView& t, const void* d_records, int32_t* d_prev, uint64_t n, hipStream_t s);
hipError_t launch_par_cuts(const int32_t* d_prev, uint64_t n, uint32_t max_entries, uint32_t live0, uint32_t* d_cuts, uint32_t max_cuts,
                           uint32_t* d_ctl, uint64_t b_begin, uint64_t b_end, uint32_t* host_cuts, uint32_t* host_state, hipStream_t s);
uint64_t par_cut_span();
uint64_t par_walk_blocks(uint64_t n);
uint64_t par_prev_entries(uint64_t n);
int32_t par_prev_pad_value();
hipError_t launch_par_middle(const void* d_records, const uint64_t* d_keys_sorted, uint64_t n, const int32_t* d_prev, const uint32_t* d_cuts,
                             uint32_t t_lo, uint32_t t_hi, uint32_t i_lo, uint32_t i_hi, uint32_t max_entries, const SketchView& sk, uint32_t* d_pos,
                             void* d_out, uint32_t* d_long, uint32_t long_cap, uint32_t* d_huge, uint32_t* d_tiles, uint32_t* d_n_long, uint32_t* d_bad, hipStream_t s);
#ifdef NFAGG_DIAG
hipError_t diag_set_dense(void* p);       // nfagg_kernels.hip: the dense-identity timing experiment (libnfagg_diag.so only)
#endif
uint32_t par_seg_short();
uint32_t par_huge_cap();
uint64_t par_rank_tiles(uint64_t records);
// Kernel-dedup mode: nfagg_dedup.hip (direct: a claim pass, then a fold pass) / nfagg_dedup_cached.hip (one streaming pass + partitions).
hipError_t launch_ingest_dedup(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base, hipStream_t s);
hipError_t launch_ingest_dedup_cached(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, uint64_t seq_base, int variant, hipStream_t s);
hipError_t launch_evict_dedup(const TableView& t, uint64_t n_live, uint64_t seq_limit, void* d_out, hipStream_t s);
// Careful path, phase A: claim slots only; writes the slot index of every record.
hipError_t launch_claim(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base,
                        uint32_t* d_slot_idx, hipStream_t s);
// Careful path: flags[i] = record i is the first occurrence of a key new in this epoch chunk;
// block_counts[b] = number of flags set in block b of kFlagBlock records.
constexpr int kFlagBlock = 1024;
hipError_t launch_first_flags(const TableView& t, const uint32_t* d_slot_idx, uint64_t n, uint64_t seq_base,
                              uint8_t* d_flags, uint32_t* d_block_counts, hipStream_t s);
// Write every live flow whose first record has seq < seq_limit as a 144-byte
// record (dense, order unspecified), reset n_live. The table is not touched: the caller bumps the epoch.
hipError_t launch_sort_slots(const uint32_t* d_in, uint32_t* d_out, uint64_t n, int end_bit, void* d_temp, size_t* temp_bytes, hipStream_t s);
hipError_t launch_evict(const TableView& t, uint64_t n_live, uint64_t seq_limit, void* d_out, hipStream_t s);
hipError_t launch_evict_filtered(const TableView& t, uint64_t n_live, uint64_t seq_limit, void* d_out, hipStream_t s);
// Optimistic fold support (nfagg_api.hip: a batch that MIGHT cross max_entries is folded whole and rolled back if it did).
// Raw copies of the claimed slots live_list[0..n) (hot, cold and — dedup mode — aux lines) into / out of a scratch area laid
// out as [n hot lines][n cold lines][n aux pairs]; zeroing of the slots live_list[from..to); the first sequence number of
// the slots live_list[from..to) (what the split point is selected from); counters rewritten after a rollback.
size_t snapshot_bytes(const TableView& t, uint64_t n);
hipError_t launch_snapshot(const TableView& t, uint64_t n, void* d_snap, bool restore, hipStream_t s);
hipError_t launch_discard(const TableView& t, uint64_t from, uint64_t to, hipStream_t s);
hipError_t launch_first_seqs(const TableView& t, uint64_t from, uint64_t to, uint32_t* d_out, hipStream_t s);
// Last launch of an ingest call: identity dwords of the slots claimed since the previous finalize, from records[0..n).
// Also resets the spill overflow tail for the next two-pass call (one launch less than a memset node).
hipError_t launch_finalize(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base, hipStream_t s);
hipError_t launch_sort_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n, void* d_temp, size_t* temp_bytes, hipStream_t s);
// Stable partition of a batch by key-hash shard (nfagg_partition.hip): buckets back to back in shard order in d_out, the
// original index of every record in d_orig, bucket sizes in d_count[0..n_shards). Scratch: partition_scratch_bytes(n, n_shards).
size_t partition_scratch_bytes(uint64_t n, uint32_t n_shards);
hipError_t launch_partition(const void* d_records, uint64_t n, uint32_t n_shards, void* d_out, uint32_t* d_orig,
                            uint64_t* d_count, void* d_scratch, hipStream_t s);
// d_cnt[s] = entries of bucket s with original index < m
hipError_t launch_partition_prefix_counts(const uint32_t* d_orig, const uint64_t* d_count, uint32_t n_shards, uint64_t m,
                                          uint64_t* d_cnt, hipStream_t s);
// Local-fold mode (nfagg_combine.hip): the live slots of a table as 192-byte PARTIALS grouped by owner shard, and their merge
// into the owner's table. launch_export_count: d_counts[0..64) = flows per owner (self_shard's are not exported);
// launch_export_scatter: d_cursor[o] = first position of segment o in d_out; launch_merge_raw: everything that combines, then
// the winner's plain identity dwords (t.n_shards / t.shard_id = the merging shard); launch_count_owned: owned flows of t itself.
constexpr size_t kPartialBytes = 192;
// Kernel-dedup mode (sub-flow tables, nfagg_dedup.h "local fold"): a partial is one SUB-FLOW's slot — hot line, cold half line and
// the eight words of the aux lines a sub-flow uses (last end, TLS words, its two earliest directions): 256 bytes.
constexpr size_t kPartialBytesDedup = 256;
// d_partials of launch_export_scatter / launch_merge_raw hold kPartialBytesDedup-byte partials when t.subflow is set.
// The join that ends a sub-flow table's epoch (nfagg_dedup_join.hip): every live sub-flow of S that (n_shards, shard_id) owns is
// folded into the flow-keyed table J by the two phases of the kernel-dedup merge (claim + first record + earliest interfaces,
// then update_existing_flow with F final); J is then evicted with launch_evict_dedup. d_slot_of: n_live words of scratch.
hipError_t launch_subflow_join(const TableView& S, const TableView& J, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards,
                               uint32_t shard_id, uint32_t* d_slot_of, hipStream_t s);
hipError_t launch_export_count(const TableView& t, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards, uint32_t self_shard,
                               unsigned long long* d_counts, hipStream_t s);
hipError_t launch_export_scatter(const TableView& t, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards, uint32_t self_shard,
                                 unsigned long long* d_cursor, void* d_out, hipStream_t s);
hipError_t launch_merge_raw(const TableView& t, const void* d_partials, uint64_t n, hipStream_t s);
hipError_t launch_count_owned(const TableView& t, uint64_t n_live, uint64_t seq_limit, unsigned long long* d_count, hipStream_t s);
// Accounter.Account with its evictions on "full" on the device, as a chain of four small kernels per window (nfagg_epoch_chain.hip;
// what nfagg_account runs for calls of a few epochs and whenever the epochs-found-first path declines): control block helpers and one
// window's launches. chain_ctl_read -> {pos, seq, live, out_pos, epoch_bits, n_epochs, stop, epoch_first, epoch_began_here}.
size_t chain_ctl_bytes();
uint32_t chain_window();
void chain_ctl_fill(void* h_ctl, const void* d_records, void* d_out, uint64_t n, uint64_t seq, uint64_t live, uint64_t list_fin, uint64_t out_cap,
                    uint64_t epoch_bits, uint64_t max_entries, uint32_t max_epochs);
void chain_ctl_read(const void* h_ctl, uint64_t out[9]);
hipError_t launch_epoch_chain_window(const TableView& t, const SketchView& sk, void* d_ctl, uint32_t* d_slot_idx, uint64_t* d_epoch_end, hipStream_t s);
// The sequence window (nfagg_rebase.hip): the tags of the live slots rebased in place; the new window starts at rebase_keep().
hipError_t launch_rebase(const TableView& t, hipStream_t s);
uint32_t rebase_keep();
// n_live = n_finalized = 0, aborted / max_probe cleared (what the eviction leaves behind), without an eviction
hipError_t launch_reset_counters(const TableView& t, hipStream_t s);
// Sketch update over a batch (nfagg_sketch.hip).
hipError_t launch_sketch_update(const SketchView& sk, const TableView& t, const void* d_records, uint64_t n, hipStream_t s);
hipError_t launch_cm_estimate(const uint64_t* d_cm, uint32_t depth, uint32_t log2w, int side, const void* d_records, uint64_t n,
                              uint64_t* d_est, uint32_t* d_idx, hipStream_t s);
hipError_t launch_cm_sort_desc(const uint64_t* d_est, uint64_t* d_est_sorted, const uint32_t* d_idx, uint32_t* d_idx_sorted, uint64_t n,
                               void* d_temp, size_t* temp_bytes, hipStream_t s);
hipError_t launch_cm_gather(const void* d_records, int side, const uint64_t* d_est_sorted, const uint32_t* d_idx_sorted, uint64_t m,
                            uint64_t* d_rows, hipStream_t s);
hipError_t launch_hll_histogram(const uint8_t* d_regs, uint32_t p, uint32_t* d_hist65, hipStream_t s);

// Per-CPU rollups (nfagg_rollup.hip). kind: 0 additional,1 dns,2 drops,3 netev,4 xlat,5 quic.
hipError_t launch_rollup(int kind, const void* d_partials, uint64_t n_flows, uint64_t n_cpu,
                         void* d_base, void* d_folded, hipStream_t s);
size_t rollup_struct_size(int kind);

// Merge of the drained maps (nfagg_rollup.hip). Index = position in Go's walk order:
// 0 main (aggregated_flows), 1 dns, 2 drops, 3 network events, 4 xlat, 5 additional, 6 quic.
struct MergeIn {
    const uint8_t* ids[7];        // 40-byte flow ids (DEVICE), 8-byte aligned
    const uint8_t* vals[7];       // [0]: nfagg_flow_metrics[n]; others: struct[n * n_cpu], flow-major
    uint32_t off[8];              // off[q] = global position of row 0 of map q; off[7] = total rows
    uint32_t n_cpu;
};
struct MergeOut {                 // dense, one entry per merged flow (DEVICE); part arrays may be null
    nfagg_flow_record* records;
    uint8_t* present;
    nfagg_additional_metrics* additional;
    nfagg_dns_metrics* dns;
    nfagg_pkt_drop_metrics* drops;
    nfagg_network_events_metrics* network_events;
    nfagg_xlat_metrics* xlat;
    nfagg_quic_metrics* quic;
};
size_t merge_slot_bytes();
// build the join (slots pre-set to 0xFF, n_slots a power of two >= 2 * rows), flag first occurrences, block-local scan
hipError_t launch_merge_build(const MergeIn& in, void* d_slots, uint32_t n_slots, uint32_t* d_slot_of, unsigned int* d_n_dup,
                              uint32_t* d_local_off, uint32_t* d_block_sum, hipStream_t s);
hipError_t launch_merge_fold(const MergeIn& in, const MergeOut& out, const void* d_slots, const uint32_t* d_slot_of,
                             const uint32_t* d_local_off, const uint64_t* d_block_base, hipStream_t s);
// exclusive scan of n_blocks block sums into d_block_base[0..n_blocks], total at [n_blocks] (nfagg_pb.hip)
hipError_t launch_scan_block_sums(const uint32_t* d_block_sum, uint32_t n_blocks, uint64_t* d_block_base, hipStream_t s);

}  // namespace nfagg
