// nfagg_api.hip — the C ABI of libnfagg (include/nfagg.h): handle, pinned
// staging ring, batch control flow of Accounter.Account
// (pkg/flow/account.go:58-100) around the kernels. No CPU fallback: without a
// gfx950 device nfagg_create fails.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nfagg.h"
#include "nfagg_internal.h"
#include "nfagg_hostpool.h"
#include "nfagg_pb.h"

using namespace nfagg;

static_assert(sizeof(nfagg_flow_id) == 40, "flow_id");
static_assert(sizeof(nfagg_flow_metrics) == 104, "flow_metrics");
static_assert(sizeof(nfagg_flow_record) == 144, "flow_record");
static_assert(sizeof(nfagg_additional_metrics) == 32, "additional");
static_assert(sizeof(nfagg_dns_metrics) == 64, "dns");
static_assert(sizeof(nfagg_pkt_drop_metrics) == 32, "drops");
static_assert(sizeof(nfagg_network_events_metrics) == 72, "netev");
static_assert(sizeof(nfagg_xlat_metrics) == 56, "xlat");
static_assert(sizeof(nfagg_quic_metrics) == 24, "quic");

namespace {

thread_local std::string g_create_error;

struct EventPair { hipEvent_t a, b; int kind; };

}  // namespace

struct nfagg_handle {
    nfagg_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t slots = 0;
    TableView tv{};
    SketchView sk{};
    bool own_sketch[4] = {false, false, false, false};
    DevCounters* h_ctr = nullptr;       // pinned mirror
    // staging ring (host ingest path)
    void* pinned[2] = {nullptr, nullptr};
    void* d_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_free[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;   // H2D copies of the staging ring: chunk k+1 goes up while chunk k is folded on `stream`
    hipEvent_t stage_up[2] = {nullptr, nullptr};
    int stage_next = 0;
    bool stage_acquired = false;
    // careful-path scratch
    uint64_t careful_chunk = 0;
    uint32_t* d_slot_idx = nullptr;
    uint8_t* d_flags = nullptr;
    uint32_t* d_block_counts = nullptr;
    std::vector<uint32_t> h_block_counts;
    // eviction buffer (device)
    void* d_evict = nullptr;
    uint64_t d_evict_cap = 0;
    // rollup scratch
    void* d_roll[3] = {nullptr, nullptr, nullptr};
    size_t d_roll_cap[3] = {0, 0, 0};
    uint32_t* d_hist = nullptr;
    // protobuf encode scratch: local offsets, block sums, block bases, namer table, (host variant) records/out/offsets/lens/keys
    // map merge scratch: [0] slots [1] slot_of [2] local_off [3] block_sum [4] block_base [5] dup counter,
    // (host variant) [6..12] ids, [13..19] values, [20..27] outputs
    void* d_mm[28] = {};
    size_t d_mm_cap[28] = {};
    void* d_hh[7] = {};            // heavy hitters: est, est sorted, idx, idx sorted, sort scratch, gathered rows, (host variant) records
    size_t d_hh_cap[7] = {};
    void* d_sort[2] = {};          // eviction: live list in slot order, radix-sort scratch
    size_t d_sort_cap[2] = {};
    int sort_bits = 0;
    bool epoch_unclustered = false; // a batch of this epoch claimed slots in arrival order (single-pass / direct / dedup kernels)
    std::vector<nfagg_intf_name> pb_names;   // host copy of the namer table, sorted (kept until the stream has consumed it)
    void* d_pb[15] = {};
    size_t d_pb_cap[15] = {};
    // optimistic fold: [0] raw slot snapshot, [1] sketch snapshot, [2] first sequence numbers (+ sorted), [3] sort scratch
    void* d_opt[4] = {};
    size_t d_opt_cap[4] = {};
    uint64_t epoch_len_hint = 0;   // records the last epoch that ended on "full" took (0 = this stream has not stopped on full)
    uint64_t last_epoch_flows = 0; // slots the last epoch had claimed when it ended (flows; sub-flows on a sub-flow table): what the next one is sized by
    uint64_t abort_cap = 0;        // largest chunk worth trying after the kernels refused claims (0 = no limit known)
    // spill queues of the two-pass ingest
    void* d_spill = nullptr;
    size_t d_spill_cap = 0;
    // accounting
    uint64_t epoch_seq = 0;         // sequence number of the next record, counted from the start of the epoch (64 bits: never runs out)
    uint64_t seq_origin = 0;        // the device sees epoch_seq - seq_origin: a 32-bit window, moved by a rebase (nfagg_rebase.hip)
    bool ext_sequenced = false;     // sequence numbers are handed in (nfagg_set_sequence, group local fold): other tables hold tags of the
                                    // same numbering, so this handle must not move its window on its own
    uint64_t live = 0;       // len(entries): exact after refresh_counters and on the paths that keep counters_exact
    uint64_t live_ub = 0;    // upper bound on the device's n_live (the claimed slots); equal to it while counters_exact
    // The host knows the device's n_live and len(entries) without asking (after an eviction; after a claim + flag chunk,
    // whose counts it has just read): the next "might this batch fill the table" / eviction needs no round trip. Every
    // asynchronous fold clears it. mirror_fresh: *h_ctr equals the device counters (what the optimistic fold's rollback restores).
    bool counters_exact = false;
    bool mirror_fresh = false;
    uint8_t* h_careful = nullptr;   // pinned: block counts + flags of a claim + flag chunk of up to kCarefulMaxBatch records
    bool must_evict = false; // a "full" split is pending (account.go:85-94)
    uint64_t split_seq = 0;
    // local fold across GPUs (nfagg_partials_*): the flows of other shards have been exported to their owners; nothing may be
    // folded on top of them before the eviction (they would be exported twice)
    bool exported = false;
    void* d_exp = nullptr;          // 64 owner counts + 64 segment cursors + the owned-flows count
    size_t d_exp_cap = 0;
    unsigned long long* h_exp = nullptr;   // pinned mirror of the counts
    // nfagg_account*: control block of the epoch kernel chain (device + pinned mirror), epoch ends, slot scratch
    void* d_ep[3] = {};             // [0] control block, [1] epoch ends, [2] live-list scratch
    size_t d_ep_cap[3] = {};
    void* d_ep_out = nullptr;       // second device buffer for evictions (nfagg_account alternates with d_evict)
    size_t d_ep_out_cap = 0;
    void* h_ep = nullptr;           // pinned: control block, then the epoch ends
    size_t h_ep_cap = 0;
    // device -> pageable host memory through two pinned bounce buffers (d2h_copy): a plain hipMemcpy to pageable memory runs at
    // ~13 GB/s here, this at the PCIe rate
    void* h_bounce[2] = {nullptr, nullptr};
    hipStream_t d2h_stream = nullptr;
    hipStream_t d2h_small = nullptr;     // small / one-off downloads (d2h_copy): not the stream the pipelined page-locked downloads use
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    hipGraphExec_t ep_graph[3] = {nullptr, nullptr, nullptr};   // kChainWindows[k] windows of the epoch kernel chain, captured once (their arguments never change)
    bool ep_graph_off = false;           // the capture or the instantiation failed once: this handle launches its windows eagerly
    void* ep_graph_key[3] = {};          // the buffers the captured launches point at: re-capture when one was re-allocated
    // sub-flow table (kernel-dedup mode of a local-fold rank, nfagg_dedup.h): the flow-keyed table its epochs are joined into
    // (nfagg_dedup_join.hip), allocated at the first eviction; scratch; the join that has been made and not yet evicted
    TableView jv{};
    DevCounters* h_jctr = nullptr;  // pinned mirror of jv.ctr
    void* d_join = nullptr;         // slot_of[]: the J slot of every live sub-flow
    size_t d_join_cap = 0;
    struct { bool valid = false, dirty = false; uint32_t n_shards = 0, shard_id = 0; uint64_t flows = 0, claimed = 0; } join;   // dirty: J may hold claims
    // the evict-on-full loop with its epochs found first (nfagg_account_par.inc): analysis arrays, pinned mirror, the stream the
    // middle epochs are folded on while the table takes the first and the last
    uint32_t* h_par = nullptr;      // pinned: control words, then the cuts
    void* d_par[8] = {};            // sort keys, sorted keys, prev, pos, long segments, sort scratch, control + cuts, rank tile counts
    size_t d_par_cap[8] = {};
    hipStream_t par_stream = nullptr;
    hipEvent_t par_done = nullptr;
    hipEvent_t par_part[16] = {};   // one behind every part of the cut walk (kParWalkPartsMax)
    nfagg_stats stats{};
    std::vector<EventPair> ev_pending;
    std::vector<EventPair> ev_free;
    std::mutex err_mu;              // nfagg_account's helper threads report through fail() too
    std::string err;
};

namespace {

int fail(nfagg_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) { std::lock_guard<std::mutex> lk(h->err_mu); h->err = buf; } else g_create_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((h), NFAGG_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ... with something to do before the error is reported (a second stream to join)
#define HIP_TRY_DO(h, expr, cleanup)                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            cleanup;                                                                          \
            return fail((h), NFAGG_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                     \
    } while (0)

uint64_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

// --- profiling: HIP events on the handle's stream around the dominant kernels
void prof_begin(nfagg_handle* h, EventPair& ep, int kind) {
    if (h->ev_free.empty()) {
        hipEventCreate(&ep.a); hipEventCreate(&ep.b);
    } else { ep = h->ev_free.back(); h->ev_free.pop_back(); }
    ep.kind = kind;
    hipEventRecord(ep.a, h->stream);
}
void prof_end(nfagg_handle* h, EventPair& ep) {
    hipEventRecord(ep.b, h->stream);
    h->ev_pending.push_back(ep);
}
void prof_resolve(nfagg_handle* h) {
    for (auto& ep : h->ev_pending) {
        hipEventSynchronize(ep.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, ep.a, ep.b);
        if (ep.kind == 0) { h->stats.ingest_kernel_ms += ms; h->stats.ingest_launches++; }
        else if (ep.kind == 1) { h->stats.evict_kernel_ms += ms; h->stats.evict_launches++; }
        else { h->stats.sketch_kernel_ms += ms; h->stats.sketch_launches++; }
        h->ev_free.push_back(ep);
    }
    h->ev_pending.clear();
}

int ensure_bytes(nfagg_handle* h, void** p, size_t* cap, size_t need);

int launch_ingest_profiled(nfagg_handle* h, const void* d, uint64_t n, uint64_t seq_base) {
    EventPair ep{};
    h->counters_exact = false; h->mirror_fresh = false;      // callers that know the outcome set counters_exact again
    const bool prof = h->cfg.profile != 0;
    if (!ingest_needs_spill((int)h->cfg.mode, (int)h->cfg.ingest_variant, n)) h->epoch_unclustered = true;
    if (ingest_needs_spill((int)h->cfg.mode, (int)h->cfg.ingest_variant, n)) {
        // room for twice the even share of a batch in which every record spills; beyond that the kernel merges directly
        uint64_t qcap = (2 * n / kSpillParts + 1024 + 3) & ~3ull;
        // overflow list: every item (record; kernel-dedup mode: or exported cache entry) may overflow, ONE slot each
        // (overflow_push_one), plus one padded group per partition per workgroup
        const uint64_t ovf_cap = ((n + 3) & ~3ull) + kDedupXpEntries + 256ull * kSpillParts * 4;
        if (qcap > h->tv.spill.qcap || ovf_cap > h->tv.spill.ovf_cap) {
            if (qcap < h->tv.spill.qcap) qcap = h->tv.spill.qcap;
            const size_t qbytes = (size_t)kSpillParts * qcap * sizeof(uint32_t);
            int rc = ensure_bytes(h, &h->d_spill, &h->d_spill_cap, qbytes + ovf_cap * sizeof(uint32_t));
            if (rc != NFAGG_OK) return rc;
            h->tv.spill.queue = (uint32_t*)h->d_spill;
            h->tv.spill.qcap = (uint32_t)qcap;
            h->tv.spill.ovf = (uint32_t*)((char*)h->d_spill + qbytes);
            h->tv.spill.ovf_cap = (uint32_t)((h->d_spill_cap - qbytes) / sizeof(uint32_t) > 0xfffffff0ull ? 0xfffffff0ull
                                             : ((h->d_spill_cap - qbytes) / sizeof(uint32_t)) & ~3ull);
        }
        if (h->cfg.mode == NFAGG_MODE_KERNEL_DEDUP) {
            // the partition pass sorts its items by sub-partition first when a partition's flows will not fit one cache (1024
            // sub-flow entries x 2048 partitions): the last epoch's flow count is the guess, the table's size before there is one
            const uint64_t guess = h->last_epoch_flows ? h->last_epoch_flows * (h->tv.subflow ? 1 : 2) : (h->cfg.max_entries > (1ull << 22) ? ~0ull : 0ull);
            h->tv.spill.sort_first = h->cfg.ingest_variant == 16 ? 2u : (guess > 2500000ull ? 1u : 0u);    // 2: whatever the partition's size (tests)      // ~1200 sub-flows per partition: one cache (and its retry round) still holds them
        }
        if (h->cfg.mode == NFAGG_MODE_KERNEL_DEDUP && !h->tv.spill.xp) {     // exported cache entries of the streaming pass (38 MB)
            size_t cap = 0;
            void* p = nullptr;
            int rc = ensure_bytes(h, &p, &cap, (size_t)kDedupXpBytes);
            if (rc != NFAGG_OK) return rc;
            h->tv.spill.xp = (uint4*)p;
        }
    }
    if (prof) { if (h->ev_pending.size() >= 8192) prof_resolve(h); prof_begin(h, ep, 0); }
    const uint64_t rel = seq_base - h->seq_origin;              // what the slots' 32-bit tags carry (callers keep rel + n inside the window)
    hipError_t e = launch_ingest(h->tv, h->sk, d, n, rel, (int)h->cfg.mode, (int)h->cfg.ingest_variant, h->stream);
    // identity dwords of the flows this batch created, from the batch (the caller owns it until the call returns)
    if (e == hipSuccess) e = launch_finalize(h->tv, d, n, rel, h->stream);
    if (prof) prof_end(h, ep);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "ingest launch failed: %s", hipGetErrorString(e));
    if (h->sk.flags && !ingest_fuses_sketches((int)h->cfg.mode, (int)h->cfg.ingest_variant, n, h->sk.flags)) {
        if (prof) prof_begin(h, ep, 2);
        e = launch_sketch_update(h->sk, h->tv, d, n, h->stream);
        if (prof) prof_end(h, ep);
        if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "sketch launch failed: %s", hipGetErrorString(e));
    }
    return NFAGG_OK;
}

// Bring the device counters to the host (synchronises the stream). In two halves for callers that wait for something else on the
// same stream anyway: enqueue the copy, wait once, take the counters in.
int refresh_counters_enqueue(nfagg_handle* h) {
    HIP_TRY(h, hipMemcpyAsync(h->h_ctr, h->tv.ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, h->stream));
    return NFAGG_OK;
}
int refresh_counters_taken(nfagg_handle* h) {
    if (h->h_ctr->error)
        return fail(h, NFAGG_EDEVICE, "flow table kernel bailed out (code %u: 1 = probe overflow / table too small, 2 = claim spin limit, 5 = spill overflow list full, 6 = claimed slot without a record, 7 = partial of another shard merged)", h->h_ctr->error);
    h->live_ub = h->h_ctr->n_live;
    if (!h->must_evict) h->live = h->h_ctr->n_live;
    h->counters_exact = true; h->mirror_fresh = true;
    h->stats.records_skipped = h->h_ctr->n_skipped;
    h->stats.records_bypassed = h->h_ctr->n_bypassed;
    if (h->h_ctr->max_probe > h->stats.max_probe) h->stats.max_probe = h->h_ctr->max_probe;
    return NFAGG_OK;
}
int refresh_counters(nfagg_handle* h) {
    int rc = refresh_counters_enqueue(h);
    if (rc != NFAGG_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return refresh_counters_taken(h);
}

// ---- optimistic fold -------------------------------------------------------------------------------------------
// A batch that MIGHT reach `len(entries) >= maxEntries` at a new key (account.go:85) — live + batch > max_entries — is
// folded whole, by the same kernels as any other batch, and checked afterwards: the sequential Accounter finds the table
// full iff the number of distinct keys (live + new ones of the batch) exceeds max_entries, so `n_live <= max_entries`
// after the fold proves that no record of the batch triggered the eviction, and the fold stands as it is. Otherwise
// the fold is rolled back — the slots claimed by the batch are zeroed, the slots that were live before it are restored
// from a raw copy taken before the fold, the sketches from theirs — and the exact split point is read off the table: every
// slot carries its flow's first sequence number (the tag of id0), so the record whose new key finds the table full is the
// (room+1)-th smallest first sequence number among the slots the batch claimed. The prefix before it is folded again
// (now exactly `room` new keys: it fits), the caller gets NFAGG_FULL + consumed as the reference's eviction demands.
// Claims are bounded by TableView.claim_limit whatever the batch holds: a batch with more new keys than the table has
// room for is `aborted` by the kernels (some claims refused), rolled back the same way and retried with a quarter of it.
constexpr size_t kCarefulCountsBytes = 256;    // h_careful: block counts of a small chunk (16 x 4 bytes), then its flags
constexpr uint64_t kCarefulMaxBatch = 16384;   // below this the claim + flag path decides (two small launches, one host round trip)

// ---- the sequence window (nfagg_rebase.hip) --------------------------------------------------------------------
constexpr uint64_t kSeqWindow = 0xFFFFFFF0ull;     // relative sequence numbers stay below this
constexpr uint64_t kMaxFoldChunk = 1ull << 31;     // records per pass of the ingest loop: always fits a fresh window

// Make room for `n` (<= kMaxFoldChunk) more sequence numbers. A table that holds the only copy of its flows rebases its tags in
// place; a handle whose numbering is shared with other tables (local fold across GPUs) cannot: *blocked tells the caller.
int ensure_seq_window(nfagg_handle* h, uint64_t n, bool* blocked) {
    if (blocked) *blocked = false;
    if (h->epoch_seq - h->seq_origin + n < kSeqWindow) return NFAGG_OK;
    // (a sub-flow table cannot either: the order BETWEEN the slots of one flow matters until the join, ranks inside a slot lose it)
    if (h->ext_sequenced || h->tv.subflow) { if (blocked) *blocked = true; return NFAGG_OK; }
    const hipError_t e = launch_rebase(h->tv, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "rebase launch failed: %s", hipGetErrorString(e));
    h->seq_origin = h->epoch_seq - rebase_keep();
    h->stats.sequence_rebases++;
    h->mirror_fresh = h->mirror_fresh;                          // counters untouched: the live list and n_live stay as they are
    return NFAGG_OK;
}

int snapshot_sketches(nfagg_handle* h, bool restore) {
    if (!h->sk.flags) return NFAGG_OK;
    const size_t cmb = (h->sk.flags & NFAGG_SKETCH_CM) ? ((size_t)h->sk.cm_depth << h->sk.cm_log2w) * sizeof(uint64_t) : 0;
    const size_t hlb = (h->sk.flags & NFAGG_SKETCH_HLL) ? ((size_t)1 << h->sk.hll_p) : 0;      // one byte per register
    int rc = ensure_bytes(h, &h->d_opt[1], &h->d_opt_cap[1], 2 * (cmb + hlb));
    if (rc != NFAGG_OK) return rc;
    char* snap = (char*)h->d_opt[1];
    for (int k = 0; k < 2; k++) {
        if (cmb) HIP_TRY(h, restore ? hipMemcpyAsync(h->sk.cm[k], snap + k * cmb, cmb, hipMemcpyDeviceToDevice, h->stream)
                                    : hipMemcpyAsync(snap + k * cmb, h->sk.cm[k], cmb, hipMemcpyDeviceToDevice, h->stream));
        if (hlb) HIP_TRY(h, restore ? hipMemcpyAsync(h->sk.hll[k], snap + 2 * cmb + k * hlb, hlb, hipMemcpyDeviceToDevice, h->stream)
                                    : hipMemcpyAsync(snap + 2 * cmb + k * hlb, h->sk.hll[k], hlb, hipMemcpyDeviceToDevice, h->stream));
    }
    return NFAGG_OK;
}

// The pieces of an optimistic fold (also driven member by member by the multi-GPU group, nfagg_group.inc):
//   opt_begin    raw copy of the live slots and of the sketches (h->live exact on entry: the caller refreshed the counters)
//   opt_fold     the fold itself, asynchronous, sequence numbers from h->epoch_seq
//   opt_result   waits; did the chunk cross max_entries / was it aborted; if it crossed: the index (relative to the
//                chunk) of the record whose new key found the table full — the (room+1)-th smallest first sequence
//                number among the slots the chunk claimed
//   opt_commit   the fold stands: account for it
//   opt_rollback table, sketches and counters as opt_begin found them
struct OptState {
    uint64_t old_live = 0, seq0 = 0, room = 0, n_after = 0;
    DevCounters before{};
    bool was_unclustered = false;
};

int opt_begin(nfagg_handle* h, OptState& st) {
    const uint64_t maxe = h->cfg.max_entries;
    st.old_live = h->live; st.seq0 = h->epoch_seq;
    st.room = maxe > st.old_live ? maxe - st.old_live : 0;
    st.before = *h->h_ctr;
    st.was_unclustered = h->epoch_unclustered;
    int rc;
    if (st.old_live) {
        if ((rc = ensure_bytes(h, &h->d_opt[0], &h->d_opt_cap[0], snapshot_bytes(h->tv, st.old_live))) != NFAGG_OK) return rc;
        hipError_t e = launch_snapshot(h->tv, st.old_live, h->d_opt[0], false, h->stream);
        if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "snapshot launch failed: %s", hipGetErrorString(e));
    }
    return snapshot_sketches(h, false);
}

int opt_fold(nfagg_handle* h, const OptState& st, const void* d, uint64_t chunk) {
    return chunk ? launch_ingest_profiled(h, d, chunk, st.seq0) : NFAGG_OK;
}

int opt_result(nfagg_handle* h, OptState& st, uint64_t chunk, bool* crossed, bool* aborted, uint64_t* split) {
    int rc = refresh_counters(h);
    if (rc != NFAGG_OK) return rc;
    const uint64_t maxe = h->cfg.max_entries;
    st.n_after = h->h_ctr->n_live;
    *aborted = h->h_ctr->aborted != 0;
    *crossed = *aborted || st.n_after > maxe;
    *split = 0;
    h->stats.optimistic_folds++;
    if (!*crossed || *aborted) return NFAGG_OK;
    const uint64_t m = st.n_after - st.old_live;               // slots claimed by the chunk; m > room
    size_t temp_bytes = 0;
    hipError_t e = launch_sort_u32(nullptr, nullptr, m, nullptr, &temp_bytes, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "sort size query failed: %s", hipGetErrorString(e));
    if ((rc = ensure_bytes(h, &h->d_opt[2], &h->d_opt_cap[2], 2 * m * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_opt[3], &h->d_opt_cap[3], temp_bytes + 16)) != NFAGG_OK) return rc;
    uint32_t* seqs = (uint32_t*)h->d_opt[2];
    e = launch_first_seqs(h->tv, st.old_live, st.n_after, seqs, h->stream);
    if (e == hipSuccess) e = launch_sort_u32(seqs, seqs + m, m, h->d_opt[3], &temp_bytes, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "split search failed: %s", hipGetErrorString(e));
    uint32_t split_seq = 0;
    HIP_TRY(h, hipMemcpyAsync(&split_seq, seqs + m + st.room, sizeof split_seq, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const uint64_t rel0 = st.seq0 - h->seq_origin;              // the slots carry window-relative numbers
    if ((uint64_t)split_seq < rel0 || (uint64_t)split_seq - rel0 >= chunk)
        return fail(h, NFAGG_EDEVICE, "optimistic fold: split sequence %u outside the batch [%llu, %llu)", split_seq,
                    (unsigned long long)rel0, (unsigned long long)(rel0 + chunk));
    *split = (uint64_t)split_seq - rel0;
    return NFAGG_OK;
}

void opt_commit(nfagg_handle* h, const OptState& st, uint64_t chunk) {
    h->live = h->live_ub = h->h_ctr->n_live;
    h->counters_exact = true; h->mirror_fresh = true;          // opt_result read the counters after the fold
    h->epoch_seq = st.seq0 + chunk;
    h->stats.records_ingested += chunk;
}

int opt_rollback(nfagg_handle* h, const OptState& st) {
    h->stats.optimistic_rollbacks++;
    hipError_t e = launch_discard(h->tv, st.old_live, st.n_after, h->stream);
    if (e == hipSuccess && st.old_live) e = launch_snapshot(h->tv, st.old_live, h->d_opt[0], true, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "rollback launch failed: %s", hipGetErrorString(e));
    int rc = snapshot_sketches(h, true);
    if (rc != NFAGG_OK) return rc;
    *h->h_ctr = st.before;                                     // n_live, n_finalized, the diagnostic counters: as before the chunk
    HIP_TRY(h, hipMemcpyAsync(h->tv.ctr, h->h_ctr, sizeof(DevCounters), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));               // h_ctr is reused by the next refresh
    h->epoch_unclustered = st.was_unclustered;
    h->live = h->live_ub = st.old_live;
    return NFAGG_OK;
}

// Fold d[0..chunk) optimistically (h->live exact on entry). On return: *folded = records folded for good (the epoch's
// sequence and record counters advanced by them), *full = the record after them found the table full (account.go:85),
// *retry = nothing folded, try a shorter chunk.
int fold_optimistic(nfagg_handle* h, const void* d, uint64_t chunk, uint64_t* folded, bool* full, bool* retry) {
    *folded = 0; *full = false; *retry = false;
    OptState st;
    int rc;
    if ((rc = opt_begin(h, st)) != NFAGG_OK) return rc;
    if ((rc = opt_fold(h, st, d, chunk)) != NFAGG_OK) return rc;
    bool crossed = false, aborted = false;
    uint64_t split = 0;
    if ((rc = opt_result(h, st, chunk, &crossed, &aborted, &split)) != NFAGG_OK) return rc;
    if (!crossed) { opt_commit(h, st, chunk); *folded = chunk; return NFAGG_OK; }   // no record of the chunk found the table full
    if ((rc = opt_rollback(h, st)) != NFAGG_OK) return rc;
    if (aborted) { *retry = true; return NFAGG_OK; }
    if (split > 0 && (rc = launch_ingest_profiled(h, d, split, st.seq0)) != NFAGG_OK) return rc;
    h->live = h->live_ub = h->cfg.max_entries;                 // exactly `room` new keys in the prefix: len(entries) == maxEntries
    h->epoch_seq = st.seq0 + split;
    h->stats.records_ingested += split;
    *folded = split; *full = true;
    return NFAGG_OK;
}

int subflow_join_discard(nfagg_handle* h);

// The record arm of Accounter.Account (account.go:81-96) for a device-resident batch.
int ingest_device_core(nfagg_handle* h, const void* d_records, size_t n, size_t* consumed_out) {
    size_t consumed = 0;
    const uint64_t maxe = h->cfg.max_entries;
    const char* base = static_cast<const char*>(d_records);
    int rc = NFAGG_OK;
    if (h->must_evict || h->exported) { if (consumed_out) *consumed_out = 0; return n ? NFAGG_FULL : NFAGG_OK; }
    if (n && (rc = subflow_join_discard(h)) != NFAGG_OK) { if (consumed_out) *consumed_out = 0; return rc; }
    while (consumed < n) {
        uint64_t rem = n - consumed;
        if (rem > kMaxFoldChunk) rem = kMaxFoldChunk;
        {   // the epoch has no maximum length (account.go:58-100): when the 32-bit window of the tags is used up it is moved
            bool blocked = false;
            if ((rc = ensure_seq_window(h, rem, &blocked)) != NFAGG_OK) break;
            if (blocked && h->tv.subflow) {
                // kernel-dedup mode of a local-fold rank: the window does not move (include/nfagg.h, nfagg_config.local_fold) —
                // the epoch ends here, as it does when the table is full: the caller evicts and resubmits the rest
                h->must_evict = true; h->split_seq = h->epoch_seq;
                rc = NFAGG_FULL;
                break;
            }
            if (blocked) {
                // numbering shared with other tables (nfagg_set_sequence): the window can only move when the flows have been
                // brought together at their owners (nfagg_window_restart_device; the in-process group does it by itself)
                rc = fail(h, NFAGG_ERANGE, "sequence window used up on an externally sequenced handle: restart the window (nfagg_window_restart_device) or evict");
                break;
            }
        }
        const void* d = base + consumed * kRecordBytes;
        if (h->live_ub + rem <= maxe) {
            // no record of this batch can find len(entries) >= maxEntries (account.go:85)
            if ((rc = launch_ingest_profiled(h, d, rem, h->epoch_seq)) != NFAGG_OK) break;
            h->live_ub += rem; h->epoch_seq += rem; consumed += rem;
            h->stats.records_ingested += rem;
            continue;
        }
        if (!h->counters_exact && (rc = refresh_counters(h)) != NFAGG_OK) break;
        const uint64_t room = maxe > h->live ? maxe - h->live : 0;
        if (room >= rem) { h->live_ub = h->live; continue; }          // fits after all: the fast path above takes it
        // epochs shorter than a claim + flag chunk (tiny CACHE_MAX_FLOWS, the reference's default of 5000 among them): that path
        // finds the split with two small launches and one round trip, an optimistic fold would be rolled back every time
        const bool tiny_epochs = h->epoch_len_hint != 0 && h->epoch_len_hint <= kCarefulMaxBatch;
        if (rem > kCarefulMaxBatch && !tiny_epochs) {
            if (!h->mirror_fresh && (rc = refresh_counters(h)) != NFAGG_OK) break;      // the rollback restores *h_ctr
            // ---- optimistic path (see fold_optimistic). How much to try: everything, unless this stream has been stopping on
            // full. Then an epoch is about epoch_len_hint records long: stay clearly inside it while far from its end (the chunk
            // fits, nothing is thrown away), and go for the split with a short chunk when close — a chunk that crosses
            // max_entries is folded, rolled back and its prefix folded again, so it should not be much longer than that prefix.
            uint64_t chunk = rem;
            if (h->epoch_len_hint) {
                const uint64_t E = h->epoch_len_hint;
                uint64_t want;
                if (E >= (1ull << 22)) {
                    want = 2 * E;                                  // long epochs: one fold at full two-pass speed, the split, the prefix again
                } else {                                           // (measured: 2.6 against 1.9 G records/s at 6.6 M-record epochs)
                    const uint64_t target = E - E / 8;
                    const uint64_t tail = E > h->epoch_seq ? E - h->epoch_seq : 0;
                    uint64_t shortc = 2 * tail;                    // go for the split: about twice what is left of the epoch
                    const uint64_t floor_ = 2 * E < (1ull << 17) ? 2 * E : (1ull << 17);
                    if (shortc < floor_) shortc = floor_;
                    want = h->epoch_seq + (1ull << 16) < target ? target - h->epoch_seq : shortc;   // (0.66 against 0.33 G records/s at 0.6 M)
                }
                if (want <= kCarefulMaxBatch) want = kCarefulMaxBatch + 1;
                if (chunk > want) chunk = want;
            }
            if (h->abort_cap && chunk > h->abort_cap) chunk = h->abort_cap;
            uint64_t folded = 0; bool full = false, retry = false;
            if ((rc = fold_optimistic(h, d, chunk, &folded, &full, &retry)) != NFAGG_OK) break;
            if (retry) {
                // more new keys than the table takes (claim_limit): a quarter of it next, but never less than what cannot
                // abort at all — a chunk of claim_limit - live records claims at most that many slots
                const uint64_t safe = h->tv.claim_limit > h->live_ub ? h->tv.claim_limit - h->live_ub : 0;
                if (chunk <= safe) { rc = fail(h, NFAGG_EDEVICE, "optimistic fold aborted although the chunk fits the table"); break; }
                h->abort_cap = chunk / 4 > safe ? chunk / 4 : safe;
                continue;
            }
            consumed += folded;                                   // epoch_seq and the record counter advanced inside
            if (full) {
                h->epoch_len_hint = h->epoch_seq > 1 ? h->epoch_seq : 1;      // this epoch took that many records
                h->must_evict = true; h->split_seq = h->epoch_seq;
                rc = NFAGG_FULL;
                break;
            }
            continue;
        }
        if (room >= 65536 || (room > 0 && room >= rem / 4)) {
            // the first `room` records cannot overflow either
            if ((rc = launch_ingest_profiled(h, d, room, h->epoch_seq)) != NFAGG_OK) break;
            h->live_ub += room; h->epoch_seq += room; consumed += room;
            h->stats.records_ingested += room;
            continue;
        }
        // ---- careful path (small batches): the split point may lie inside this chunk
        const uint64_t chunk = rem < h->careful_chunk ? rem : h->careful_chunk;
        const uint64_t seq0 = h->epoch_seq;
        h->epoch_unclustered = true;
        hipError_t e = launch_claim(h->tv, d, chunk, seq0 - h->seq_origin, h->d_slot_idx, h->stream);
        if (e == hipSuccess) e = launch_first_flags(h->tv, h->d_slot_idx, chunk, seq0 - h->seq_origin, h->d_flags, h->d_block_counts, h->stream);
        if (e != hipSuccess) { rc = fail(h, NFAGG_EDEVICE, "claim launch failed: %s", hipGetErrorString(e)); break; }
        const uint64_t nblk = (chunk + kFlagBlock - 1) / kFlagBlock;
        h->h_block_counts.resize(nblk);
        h->mirror_fresh = false;
        // small chunks (every chunk of a stream with tiny epochs): counts and flags come back together, one round trip
        const bool one_trip = chunk <= kCarefulMaxBatch;
        uint32_t* counts_dst = one_trip ? reinterpret_cast<uint32_t*>(h->h_careful) : h->h_block_counts.data();
        e = hipMemcpyAsync(counts_dst, h->d_block_counts, nblk * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && one_trip) e = hipMemcpyAsync(h->h_careful + kCarefulCountsBytes, h->d_flags, chunk, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { rc = fail(h, NFAGG_EDEVICE, "claim sync failed: %s", hipGetErrorString(e)); break; }
        if (one_trip) memcpy(h->h_block_counts.data(), counts_dst, nblk * sizeof(uint32_t));
        uint64_t total_new = 0;
        for (uint64_t b = 0; b < nblk; b++) total_new += h->h_block_counts[b];
        if (h->live + total_new <= maxe) {
            if ((rc = launch_ingest_profiled(h, d, chunk, seq0)) != NFAGG_OK) break;
            h->live += total_new; h->live_ub = h->live; h->epoch_seq += chunk; consumed += chunk;
            h->stats.records_ingested += chunk;
            h->counters_exact = true;                              // n_live = live: every claimed slot is a flow of the epoch
            continue;
        }
        // The (room+1)-th new key of the chunk arrives with the table full:
        // everything before it is folded, then the caller must evict ("full").
        uint64_t want = room + 1, blk = 0, before = 0;
        while (before + h->h_block_counts[blk] < want) { before += h->h_block_counts[blk]; blk++; }
        uint8_t fl_buf[kFlagBlock];
        const uint64_t off = blk * kFlagBlock;
        const uint64_t cnt = (chunk - off) < (uint64_t)kFlagBlock ? (chunk - off) : (uint64_t)kFlagBlock;
        const uint8_t* fl = fl_buf;
        if (one_trip) fl = h->h_careful + kCarefulCountsBytes + off;
        else {
            e = hipMemcpy(fl_buf, h->d_flags + off, cnt, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { rc = fail(h, NFAGG_EDEVICE, "flag copy failed: %s", hipGetErrorString(e)); break; }
        }
        uint64_t split = off;
        for (uint64_t k = 0; k < cnt; k++) {
            if (fl[k]) { before++; if (before == want) { split = off + k; break; } }
        }
        if (split > 0 && (rc = launch_ingest_profiled(h, d, split, seq0)) != NFAGG_OK) break;
        h->epoch_seq += split; consumed += split;
        h->stats.records_ingested += split;
        h->live = maxe > h->live ? maxe : h->live;   // len(entries) == maxEntries now
        h->live_ub = h->live + (total_new - room);   // = the device's n_live: the chunk's new keys all hold a slot
        h->counters_exact = true;
        h->epoch_len_hint = h->epoch_seq > 1 ? h->epoch_seq : 1;
        h->must_evict = true;
        h->split_seq = seq0 + split;
        rc = NFAGG_FULL;
        break;
    }
    if (consumed_out) *consumed_out = consumed;
    return rc;
}

int ensure_bytes(nfagg_handle* h, void** p, size_t* cap, size_t need) {
    if (*cap >= need) return NFAGG_OK;
    if (*p) { hipFree(*p); *p = nullptr; *cap = 0; }
    size_t want = need + need / 4 + 4096;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) return fail(h, NFAGG_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    *cap = want;
    return NFAGG_OK;
}

// ---- sub-flow tables (kernel-dedup mode of a local-fold rank): the join that ends an epoch (nfagg_dedup_join.hip) ----------------
// J: a flow-keyed table of the ordinary kernel-dedup layout, as large as the sub-flow table (a join claims at most one slot per
// live sub-flow). It lives from the first eviction on and is emptied, like the main table, by its epoch tag.
int subflow_join_table(nfagg_handle* h) {
    TableView& J = h->jv;
    if (J.ctr) return NFAGG_OK;
    const uint64_t slots = h->slots;
    if (!J.hot) HIP_TRY(h, hipMalloc((void**)&J.hot, slots * sizeof(SlotHot)));
    if (!J.cold) HIP_TRY(h, hipMalloc((void**)&J.cold, slots * sizeof(SlotCold)));
    if (!J.aux) HIP_TRY(h, hipMalloc((void**)&J.aux, slots * sizeof(SlotAux)));
    if (!J.live_list) HIP_TRY(h, hipMalloc((void**)&J.live_list, slots * sizeof(uint32_t)));
    if (!h->h_jctr) HIP_TRY(h, hipHostMalloc((void**)&h->h_jctr, sizeof(DevCounters), hipHostMallocDefault));
    DevCounters* c = nullptr;
    HIP_TRY(h, hipMalloc((void**)&c, sizeof(DevCounters)));
    hipError_t e = hipMemsetAsync(c, 0, sizeof(DevCounters), h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(J.hot, 0, slots * sizeof(SlotHot), h->stream);       // the tags: every slot free
    if (e != hipSuccess) { hipFree(c); return fail(h, NFAGG_EDEVICE, "join table: %s", hipGetErrorString(e)); }
    J.mask = slots - 1; J.claim_limit = h->tv.claim_limit; J.epoch_bits = 1ull << 48;
    J.n_shards = 1; J.shard_id = 0; J.defer_claims = 0; J.subflow = 0;
    J.ctr = c;                                                  // last: marks the table as complete
    h->stats.table_bytes += slots * (sizeof(SlotHot) + sizeof(SlotCold) + sizeof(SlotAux));
    return NFAGG_OK;
}

// J's slots belong to a past epoch from now on (nothing is written); its counters are reset by the launches that emptied it
int subflow_join_table_next_epoch(nfagg_handle* h) {
    uint64_t next_epoch = (h->jv.epoch_bits >> 48) + 1;
    if (next_epoch > 0xFFFFull) {
        HIP_TRY(h, hipMemsetAsync(h->jv.hot, 0, h->slots * sizeof(SlotHot), h->stream));
        next_epoch = 1;
    }
    h->jv.epoch_bits = next_epoch << 48;
    return NFAGG_OK;
}

// A join that was made and not delivered (the caller's buffer was too small) is void once the table changes again.
int subflow_join_discard(nfagg_handle* h) {
    if (!h->join.dirty) { h->join.valid = false; return NFAGG_OK; }      // (dirty without valid: a join that failed part-way)
    h->join.valid = false; h->join.dirty = false;
    const hipError_t e = launch_reset_counters(h->jv, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "join table reset failed: %s", hipGetErrorString(e));
    return subflow_join_table_next_epoch(h);
}

// Join the live sub-flows that (n_shards, shard_id) owns (n_shards = 1: all of them) into J. Synchronises; h->join holds the outcome.
int subflow_join(nfagg_handle* h, uint32_t n_shards, uint32_t shard_id) {
    int rc;
    if (h->join.valid && h->join.n_shards == n_shards && h->join.shard_id == shard_id) return NFAGG_OK;
    if ((rc = subflow_join_discard(h)) != NFAGG_OK) return rc;
    if ((rc = refresh_counters(h)) != NFAGG_OK) return rc;
    if (h->h_ctr->aborted) return fail(h, NFAGG_EDEVICE, "table too small for the flows this shard owns (claims refused while merging)");
    const uint64_t claimed = h->h_ctr->n_live;
    if ((rc = subflow_join_table(h)) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_join, &h->d_join_cap, (size_t)claimed * sizeof(uint32_t) + 16)) != NFAGG_OK) return rc;
    const uint64_t seq_limit = h->must_evict ? h->split_seq - h->seq_origin : ~0ull;
    h->join.dirty = true;
    const hipError_t e = launch_subflow_join(h->tv, h->jv, claimed, seq_limit, n_shards, shard_id, (uint32_t*)h->d_join, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "sub-flow join launch failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipMemcpyAsync(h->h_jctr, h->jv.ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->h_jctr->error || h->h_jctr->aborted)
        return fail(h, NFAGG_EDEVICE, "sub-flow join bailed out (code %u, aborted %u)", h->h_jctr->error, h->h_jctr->aborted);
    h->join.valid = true; h->join.n_shards = n_shards; h->join.shard_id = shard_id;
    h->join.flows = h->h_jctr->n_live; h->join.claimed = claimed;
    return NFAGG_OK;
}

}  // namespace

// The NUMA node a device hangs off: /sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node (-1: unknown / not a NUMA host).
static int device_numa_node(int device) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* c = bdf; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// First handle of the process: the copy workers, bound next to its GPU. Later handles (other GPUs of a group) share them.
static void host_pool_for_device(int device) {
    static std::once_flag once;
    std::call_once(once, [device] { (void)HostPool::get().configure(0, device_numa_node(device), /*explicit_call=*/false); });
}

extern "C" {

uint32_t nfagg_abi_version(void) { return NFAGG_ABI_VERSION; }

const char* nfagg_last_error(const nfagg_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int nfagg_create(const nfagg_config* cfg_in, nfagg_handle** out) {
    if (!cfg_in || !out) return fail(nullptr, NFAGG_EINVAL, "null argument");
    if (cfg_in->struct_size != sizeof(nfagg_config))
        return fail(nullptr, NFAGG_EINVAL, "nfagg_config.struct_size %u != %zu (ABI mismatch)", cfg_in->struct_size, sizeof(nfagg_config));
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NFAGG_ENODEV, "no HIP device available (%s); libnfagg has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    nfagg_config cfg = *cfg_in;
    if (cfg.device < 0 || cfg.device >= ndev) return fail(nullptr, NFAGG_EINVAL, "device %d out of range (0..%d)", cfg.device, ndev - 1);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg.device) != hipSuccess) return fail(nullptr, NFAGG_ENODEV, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, NFAGG_ENODEV, "device %d is %s; libnfagg is built for gfx950 (MI355X) only", cfg.device, prop.gcnArchName);
    if (cfg.max_entries == 0) cfg.max_entries = 5000;   // CACHE_MAX_FLOWS default (pkg/config/config.go:146)
    if (cfg.mode != NFAGG_MODE_ACCOUNTER && cfg.mode != NFAGG_MODE_KERNEL_DEDUP)
        return fail(nullptr, NFAGG_EINVAL, "unknown mode %u", cfg.mode);
    if (!ingest_variant_supported((int)cfg.ingest_variant))
        return fail(nullptr, NFAGG_EINVAL, "ingest_variant %u is not part of this build (phase-timing builds live in libnfagg_diag.so)", cfg.ingest_variant);
    if (cfg.cm_depth == 0) cfg.cm_depth = 4;
    if (cfg.cm_log2_width == 0) cfg.cm_log2_width = 20;
    if (cfg.hll_p == 0) cfg.hll_p = 14;
    if (cfg.cm_depth > 8 || cfg.cm_log2_width < 4 || cfg.cm_log2_width > 28 || cfg.hll_p < 4 || cfg.hll_p > 18)
        return fail(nullptr, NFAGG_EINVAL, "sketch parameters out of range");
    if (cfg.staging_records == 0) cfg.staging_records = 1ull << 20;
    if (cfg.copy_threads > 64) return fail(nullptr, NFAGG_EINVAL, "copy_threads > 64");      // 0: what the copy workers' calibration found best
    if (cfg.n_shards == 0) cfg.n_shards = 1;
    if (cfg.shard_id >= cfg.n_shards) return fail(nullptr, NFAGG_EINVAL, "shard_id %u >= n_shards %u", cfg.shard_id, cfg.n_shards);
    if (cfg.local_fold > 1) return fail(nullptr, NFAGG_EINVAL, "local_fold must be 0 or 1");
    if (cfg.local_fold && cfg.n_shards > 1) return fail(nullptr, NFAGG_EINVAL, "a local-fold rank folds whatever arrives at it: n_shards must be 0 or 1");
    uint64_t slots = cfg.table_log2_slots ? (1ull << cfg.table_log2_slots) : next_pow2(2 * cfg.max_entries);
    if (slots < (1ull << 16)) slots = 1ull << 16;
    if (slots < 2 * cfg.max_entries) return fail(nullptr, NFAGG_EINVAL, "table_log2_slots too small: need >= 2*max_entries slots");
    if (slots > (1ull << 31)) return fail(nullptr, NFAGG_EINVAL, "table larger than 2^31 slots not supported");

    // caller-owned sketch buffers: Count-Min counters are 64-bit words; HyperLogLog registers are ONE BYTE each (round 4) and are
    // raised by a CAS on the 32-bit word that holds them — a buffer that is not 4-byte aligned cannot be updated
    for (int k = 0; k < 2; k++) {
        if (((uintptr_t)cfg.ext_sketch[k] & 7u) != 0) return fail(nullptr, NFAGG_EINVAL, "ext_sketch[%d] (Count-Min, uint64 counters) must be 8-byte aligned", k);
        if (((uintptr_t)cfg.ext_sketch[2 + k] & 3u) != 0)
            return fail(nullptr, NFAGG_EINVAL, "ext_sketch[%d] (HyperLogLog: 1 << hll_p registers of one byte each) must be 4-byte aligned", 2 + k);
    }
    nfagg_handle* h = new (std::nothrow) nfagg_handle();
    if (!h) return fail(nullptr, NFAGG_ENOMEM, "out of host memory");
    h->cfg = cfg; h->device = cfg.device; h->slots = slots;
#define CREATE_TRY(expr)                                                                         \
    do {                                                                                         \
        hipError_t e2_ = (expr);                                                                 \
        if (e2_ != hipSuccess) {                                                                 \
            int rc_ = fail(nullptr, (e2_ == hipErrorOutOfMemory) ? NFAGG_ENOMEM : NFAGG_EDEVICE, \
                           "%s failed: %s", #expr, hipGetErrorString(e2_));                      \
            nfagg_destroy(h);                                                                    \
            return rc_;                                                                          \
        }                                                                                        \
    } while (0)
    CREATE_TRY(hipSetDevice(h->device));
    host_pool_for_device(h->device);            // the copy workers: created with the first handle, next to its GPU (nfagg_hostpool.h)
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipMalloc((void**)&h->tv.hot, slots * sizeof(SlotHot)));
    CREATE_TRY(hipMalloc((void**)&h->tv.cold, slots * sizeof(SlotCold)));
    // the live list can transiently hold bogus claims of the careful path: size it by slots
    CREATE_TRY(hipMalloc((void**)&h->tv.live_list, slots * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc((void**)&h->tv.ctr, sizeof(DevCounters)));
    CREATE_TRY(hipMemsetAsync(h->tv.hot, 0, slots * sizeof(SlotHot), h->stream));
    CREATE_TRY(hipMemsetAsync(h->tv.cold, 0, slots * sizeof(SlotCold), h->stream));
    if (cfg.mode == NFAGG_MODE_KERNEL_DEDUP) {
        CREATE_TRY(hipMalloc((void**)&h->tv.aux, slots * sizeof(SlotAux)));
        CREATE_TRY(hipMemsetAsync(h->tv.aux, 0, slots * sizeof(SlotAux), h->stream));
    }
    CREATE_TRY(hipMemsetAsync(h->tv.ctr, 0, sizeof(DevCounters), h->stream));
    CREATE_TRY(hipHostMalloc((void**)&h->h_ctr, sizeof(DevCounters), hipHostMallocDefault));
    CREATE_TRY(hipHostMalloc((void**)&h->h_careful, kCarefulCountsBytes + kCarefulMaxBatch, hipHostMallocDefault));
    h->tv.mask = slots - 1; h->tv.n_shards = cfg.n_shards; h->tv.shard_id = cfg.shard_id;
    // a local-fold rank in kernel-dedup mode keys its table by (flow, interface): nfagg_dedup.h, include/nfagg.h (local_fold)
    h->tv.subflow = (cfg.mode == NFAGG_MODE_KERNEL_DEDUP && cfg.local_fold) ? 1u : 0u;
    h->tv.claim_limit = slots / 4 * 3 + 16;   // max_entries <= slots/2 plus a careful chunk <= slots/4 always fit
    h->tv.epoch_bits = 1ull << 48;            // eviction epoch 1; 0 is "never used"
    // pass 2 of the two-pass fold may collect its claims per workgroup (up to 1024 each, 256 workgroups resident) before
    // counting them: only where that many uncounted claims cannot fill the table
    h->tv.defer_claims = slots >= (1ull << 21) ? 1u : 0u;
    // careful path: never let claimed slots exceed 3/4 of the table
    h->careful_chunk = slots / 4;
    if (h->careful_chunk > (1ull << 22)) h->careful_chunk = 1ull << 22;
    CREATE_TRY(hipMalloc((void**)&h->d_slot_idx, h->careful_chunk * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc((void**)&h->d_flags, h->careful_chunk));
    CREATE_TRY(hipMalloc((void**)&h->d_block_counts, ((h->careful_chunk + kFlagBlock - 1) / kFlagBlock) * sizeof(uint32_t)));
    for (int b = 0; b < 2; b++) CREATE_TRY(hipEventCreateWithFlags(&h->stage_free[b], hipEventDisableTiming));
    for (int b = 0; b < 2; b++) CREATE_TRY(hipEventCreateWithFlags(&h->stage_up[b], hipEventDisableTiming));
    CREATE_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    // sketches
    h->sk.cm_depth = cfg.cm_depth; h->sk.cm_log2w = cfg.cm_log2_width; h->sk.hll_p = cfg.hll_p;
    h->sk.flags = cfg.sketch_flags & (NFAGG_SKETCH_CM | NFAGG_SKETCH_HLL);
    if (h->sk.flags & NFAGG_SKETCH_CM) {
        const size_t bytes = ((size_t)cfg.cm_depth << cfg.cm_log2_width) * sizeof(uint64_t);
        for (int k = 0; k < 2; k++) {
            if (cfg.ext_sketch[k]) h->sk.cm[k] = (uint64_t*)cfg.ext_sketch[k];
            else { CREATE_TRY(hipMalloc((void**)&h->sk.cm[k], bytes)); h->own_sketch[k] = true; CREATE_TRY(hipMemsetAsync(h->sk.cm[k], 0, bytes, h->stream)); }
        }
    }
    if (h->sk.flags & NFAGG_SKETCH_HLL) {
        const size_t bytes = ((size_t)1 << cfg.hll_p);              // uint8_t registers (hll_p >= 4: a multiple of 4 bytes)
        for (int k = 0; k < 2; k++) {
            if (cfg.ext_sketch[2 + k]) h->sk.hll[k] = (uint8_t*)cfg.ext_sketch[2 + k];
            else { CREATE_TRY(hipMalloc((void**)&h->sk.hll[k], bytes)); h->own_sketch[2 + k] = true; CREATE_TRY(hipMemsetAsync(h->sk.hll[k], 0, bytes, h->stream)); }
        }
    }
    CREATE_TRY(hipMalloc((void**)&h->d_hist, 65 * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc((void**)&h->tv.spill.qtail, (kSpillParts + 1) * sizeof(uint32_t)));
    CREATE_TRY(hipMemsetAsync(h->tv.spill.qtail, 0, (kSpillParts + 1) * sizeof(uint32_t), h->stream));
    h->tv.spill.ovf_tail = h->tv.spill.qtail + kSpillParts;
    h->tv.spill.error = &h->tv.ctr->error;
    {   // Partition = the TOP 11 bits of the home slot index: the flows one pass-2 workgroup merges live in one 1/2048 of the
        // table (a 2^17-slot window of a 2^28-slot table) instead of all over it, and what is appended to the live list at
        // about the same time comes from the ~256 windows being flushed at that time — table pages stay hot in the TLBs
        // during the flush and during eviction.
        int bits = 0;
        while ((1ull << bits) < slots) bits++;
        h->tv.spill.part_shift = (uint32_t)(bits - 11);
        h->tv.spill.n_parts = kSpillParts;
    }
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    h->stats.table_slots = slots;
    h->stats.table_bytes = slots * (sizeof(SlotHot) + sizeof(SlotCold) + (h->tv.aux ? sizeof(SlotAux) : 0));
    *out = h;
    return NFAGG_OK;
}

void nfagg_destroy(nfagg_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    prof_resolve(h);
    for (auto& ep : h->ev_free) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
    for (int b = 0; b < 2; b++) {
        if (h->pinned[b]) hipHostFree(h->pinned[b]);
        if (h->d_stage[b]) hipFree(h->d_stage[b]);
        if (h->stage_free[b]) hipEventDestroy(h->stage_free[b]);
        if (h->stage_up[b]) hipEventDestroy(h->stage_up[b]);
    }
    if (h->copy_stream) hipStreamDestroy(h->copy_stream);
    for (int k = 0; k < 2; k++) {
        if (h->own_sketch[k] && h->sk.cm[k]) hipFree(h->sk.cm[k]);
        if (h->own_sketch[2 + k] && h->sk.hll[k]) hipFree(h->sk.hll[k]);
    }
    for (int k = 0; k < 3; k++) if (h->d_roll[k]) hipFree(h->d_roll[k]);
    if (h->d_hist) hipFree(h->d_hist);
    if (h->d_spill) hipFree(h->d_spill);
    if (h->tv.spill.xp) hipFree(h->tv.spill.xp);
    for (int k = 0; k < 4; k++) if (h->d_opt[k]) hipFree(h->d_opt[k]);
    for (int k = 0; k < 15; k++) if (h->d_pb[k]) hipFree(h->d_pb[k]);
    for (int k = 0; k < 2; k++) if (h->d_sort[k]) hipFree(h->d_sort[k]);
    for (int k = 0; k < 7; k++) if (h->d_hh[k]) hipFree(h->d_hh[k]);
    for (int k = 0; k < 28; k++) if (h->d_mm[k]) hipFree(h->d_mm[k]);
    if (h->tv.spill.qtail) hipFree(h->tv.spill.qtail);
    if (h->d_evict) hipFree(h->d_evict);
    if (h->d_slot_idx) hipFree(h->d_slot_idx);
    if (h->d_flags) hipFree(h->d_flags);
    if (h->d_block_counts) hipFree(h->d_block_counts);
    if (h->tv.hot) hipFree(h->tv.hot);
    if (h->tv.cold) hipFree(h->tv.cold);
    if (h->tv.aux) hipFree(h->tv.aux);
    if (h->tv.live_list) hipFree(h->tv.live_list);
    if (h->tv.ctr) hipFree(h->tv.ctr);
    if (h->h_par) hipHostFree(h->h_par);
    for (int k = 0; k < 8; k++) if (h->d_par[k]) hipFree(h->d_par[k]);
    if (h->par_done) hipEventDestroy(h->par_done);
    for (int k = 0; k < 16; k++) if (h->par_part[k]) hipEventDestroy(h->par_part[k]);
    if (h->par_stream) hipStreamDestroy(h->par_stream);
    if (h->jv.hot) hipFree(h->jv.hot);
    if (h->jv.cold) hipFree(h->jv.cold);
    if (h->jv.aux) hipFree(h->jv.aux);
    if (h->jv.live_list) hipFree(h->jv.live_list);
    if (h->jv.ctr) hipFree(h->jv.ctr);
    if (h->h_jctr) hipHostFree(h->h_jctr);
    if (h->d_join) hipFree(h->d_join);
    if (h->h_ctr) hipHostFree(h->h_ctr);
    if (h->h_careful) hipHostFree(h->h_careful);
    if (h->h_exp) hipHostFree(h->h_exp);
    if (h->d_exp) hipFree(h->d_exp);
    for (int b = 0; b < 2; b++) { if (h->h_bounce[b]) hipHostFree(h->h_bounce[b]); if (h->bounce_ev[b]) hipEventDestroy(h->bounce_ev[b]); }
    if (h->d2h_stream) hipStreamDestroy(h->d2h_stream);
    if (h->d2h_small) hipStreamDestroy(h->d2h_small);
    for (int k = 0; k < 3; k++) if (h->ep_graph[k]) hipGraphExecDestroy(h->ep_graph[k]);
    if (h->h_ep) hipHostFree(h->h_ep);
    if (h->d_ep_out) hipFree(h->d_ep_out);
    for (int k = 0; k < 3; k++) if (h->d_ep[k]) hipFree(h->d_ep[k]);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int nfagg_ingest_device(nfagg_handle* h, const void* d_records, size_t n, size_t* consumed) {
    if (!h || (!d_records && n)) return fail(h, NFAGG_EINVAL, "null argument");
    if (((uintptr_t)d_records & 15u) != 0) return fail(h, NFAGG_EINVAL, "device records must be 16-byte aligned");
    HIP_TRY(h, hipSetDevice(h->device));
    return ingest_device_core(h, d_records, n, consumed);
}

static int staging_alloc(nfagg_handle* h) {
    if (h->pinned[0]) return NFAGG_OK;
    const size_t bytes = (size_t)h->cfg.staging_records * kRecordBytes;
    for (int b = 0; b < 2; b++) {
        HIP_TRY(h, hipHostMalloc(&h->pinned[b], bytes, hipHostMallocDefault));
        HIP_TRY(h, hipMalloc(&h->d_stage[b], bytes));
    }
    return NFAGG_OK;
}

// Caller buffer -> pinned staging buffer (and bounce buffer -> caller buffer). One core's memcpy (~28 GB/s) is slower than the PCIe
// link the pinned buffer feeds (~57 GB/s), so large copies are cut into parts for the process's copy workers (nfagg_hostpool.h:
// bound to the GPU's NUMA node, non-temporal stores). threads: cfg.copy_threads — 0 = the number of parts the workers' calibration
// found best on this host, 1 = inline.
static void staged_copy(void* dst, const void* src, size_t bytes, unsigned threads) {
    if (threads == 1) { memcpy(dst, src, bytes); return; }
    HostPool::get().copy(dst, src, bytes, threads);
}

// Device -> caller buffer (pageable). The data must be complete on the device (the caller synchronised the producing stream).
// Chunks go down into two pinned bounce buffers in turn on a stream of their own; while chunk k travels, chunk k-1 is copied
// out of its bounce buffer by cfg.copy_threads threads — the mirror image of the H2D staging ring.
// Is p page-locked host memory the device can read or write directly (nfagg_host_alloc, hipHostMalloc, hipHostRegister)? Such
// a buffer needs no trip through the library's own pinned buffers: the DMA engine takes it as it is.
static bool host_is_pinned(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc memory: "invalid value"
    return a.type == hipMemoryTypeHost;
}

constexpr size_t kBounceBytes = 16u << 20;
static int d2h_copy(nfagg_handle* h, void* dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return NFAGG_OK;
    if (bytes < (4u << 20) || host_is_pinned(dst)) {
        // NOT a blocking hipMemcpy: that one runs on the legacy stream, and nfagg_account calls this from a helper thread while the
        // calling thread may be CAPTURING the epoch chain's graph on the handle's stream — the capture was invalidated now and then
        // ("operation failed due to a previous error during capture", once in ~240 soak streams: profiles/r05_soak_account.txt).
        // On a stream of ITS OWN: a copy into pageable memory on d2h_stream left that stream's later page-locked downloads at
        // a third of the link (nfagg_account from page-locked buffers 22.0 -> 28.0 ms, same box: profiles/r05_account_host_pipeline.txt)
        if (!h->d2h_small) HIP_TRY(h, hipStreamCreateWithFlags(&h->d2h_small, hipStreamNonBlocking));
        HIP_TRY(h, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->d2h_small));
        HIP_TRY(h, hipStreamSynchronize(h->d2h_small));
        return NFAGG_OK;
    }
    if (!h->d2h_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->d2h_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) {                                // created on first use; a failure part-way is picked up by the next call
        if (!h->h_bounce[b]) HIP_TRY(h, hipHostMalloc(&h->h_bounce[b], kBounceBytes, hipHostMallocDefault));
        if (!h->bounce_ev[b]) HIP_TRY(h, hipEventCreateWithFlags(&h->bounce_ev[b], hipEventDisableTiming));
    }
    const size_t n_chunks = (bytes + kBounceBytes - 1) / kBounceBytes;
    for (size_t k = 0; k <= n_chunks; k++) {
        if (k < n_chunks) {
            const size_t off = k * kBounceBytes, len = bytes - off < kBounceBytes ? bytes - off : kBounceBytes;
            HIP_TRY(h, hipMemcpyAsync(h->h_bounce[k & 1], (const char*)d_src + off, len, hipMemcpyDeviceToHost, h->d2h_stream));
            HIP_TRY(h, hipEventRecord(h->bounce_ev[k & 1], h->d2h_stream));
        }
        if (k > 0) {
            const size_t off = (k - 1) * kBounceBytes, len = bytes - off < kBounceBytes ? bytes - off : kBounceBytes;
            HIP_TRY(h, hipEventSynchronize(h->bounce_ev[(k - 1) & 1]));
            staged_copy((char*)dst + off, h->h_bounce[(k - 1) & 1], len, h->cfg.copy_threads);
        }
    }
    return NFAGG_OK;
}

// How much of a host buffer goes up in one piece. A stream that keeps stopping on "full" (small CACHE_MAX_FLOWS: the reference's
// default is 5000, config.go:146) consumes one epoch per call — about epoch_len_hint records — and what was staged beyond that
// is staged again by the next call: copy what is left of the epoch and a quarter more (a longer epoch simply takes another
// chunk), not a whole staging buffer. Measured at max_entries 5000: 3.6 -> 36 M records/s through nfagg_ingest / nfagg_evict.
static size_t stage_chunk(const nfagg_handle* h, size_t remaining, size_t cap) {
    size_t m = remaining < cap ? remaining : cap;
    if (h->epoch_len_hint) {
        const uint64_t reach = h->epoch_len_hint + h->epoch_len_hint / 4;
        uint64_t want = reach > h->epoch_seq ? reach - h->epoch_seq : 0;
        if (want < kCarefulMaxBatch) want = kCarefulMaxBatch;
        if (m > want) m = (size_t)want;
    }
    return m;
}

int nfagg_ingest(nfagg_handle* h, const void* records, size_t n, size_t* consumed_out) {
    if (!h || (!records && n)) return fail(h, NFAGG_EINVAL, "null argument");
    if (h->stage_acquired) return fail(h, NFAGG_ESTATE, "a staging buffer is acquired; commit it first");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = staging_alloc(h);
    if (rc != NFAGG_OK) return rc;
    size_t consumed = 0;
    const char* src = static_cast<const char*>(records);
    const size_t cap = (size_t)h->cfg.staging_records;
    const bool pinned_src = host_is_pinned(records);            // page-locked caller buffer: sent up as it is, no host copy
    // ---- the whole call folded optimistically (round 5). A call of several staging buffers into a table it MIGHT fill (live + n >
    // max_entries) used to ask the device for the exact len(entries) before every chunk — a stream synchronisation that also stops
    // the host from copying the next chunk: host copy and upload took turns instead of overlapping (20 M records from pageable
    // memory: 88 ms, 227 M records/s, where the same call into a table that cannot fill ran at the link's 374 M). Here the chunks
    // are folded without a look in between and the proof comes ONCE, after the last: n_live <= max_entries and no refused claim
    // means no record of the call found the map full (account.go:85; DESIGN.md §2 "optimistic fold"). Otherwise everything is
    // rolled back (raw copy of the live slots, sketches, counters: taken before the first chunk) and the loop below does it step by step.
    if (n > cap && n <= kMaxFoldChunk && h->epoch_len_hint == 0 && !h->must_evict && !h->exported && !h->tv.subflow && !h->ext_sequenced) {
        if (!h->mirror_fresh && (rc = refresh_counters(h)) != NFAGG_OK) return rc;
        bool blocked = false;
        if (h->live + n > h->cfg.max_entries && (rc = ensure_seq_window(h, n, &blocked)) == NFAGG_OK && !blocked) {
            OptState st;
            if ((rc = opt_begin(h, st)) != NFAGG_OK) return rc;
            size_t off = 0;
            while (off < n && rc == NFAGG_OK) {
                const size_t m = (n - off) < cap ? (n - off) : cap;
                const int b = h->stage_next;
                HIP_TRY(h, hipEventSynchronize(h->stage_free[b]));
                const void* up = src + off * kRecordBytes;
                if (!pinned_src) { staged_copy(h->pinned[b], up, m * kRecordBytes, h->cfg.copy_threads); up = h->pinned[b]; }
                HIP_TRY(h, hipMemcpyAsync(h->d_stage[b], up, m * kRecordBytes, hipMemcpyHostToDevice, h->copy_stream));
                HIP_TRY(h, hipEventRecord(h->stage_up[b], h->copy_stream));
                HIP_TRY(h, hipStreamWaitEvent(h->stream, h->stage_up[b], 0));
                rc = launch_ingest_profiled(h, h->d_stage[b], m, st.seq0 + off);
                HIP_TRY(h, hipEventRecord(h->stage_free[b], h->stream));
                h->stage_next ^= 1;
                off += m;
            }
            if (rc != NFAGG_OK) return rc;
            if ((rc = refresh_counters(h)) != NFAGG_OK) return rc;       // the one wait of the call
            h->stats.optimistic_folds++;
            st.n_after = h->h_ctr->n_live;
            if (!h->h_ctr->aborted && st.n_after <= h->cfg.max_entries) {
                opt_commit(h, st, n);
                if (pinned_src) HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
                if (consumed_out) *consumed_out = n;
                return NFAGG_OK;
            }
            if ((rc = opt_rollback(h, st)) != NFAGG_OK) return rc;       // some record found the map full: step by step, below
        } else if (rc != NFAGG_OK) return rc;
    }
    // pinned ring, double buffered: the CPU fills buffer b+1 while the GPU
    // copies/folds buffer b (tracer_ringbuf.go:112-134 forwards one record at a time)
    while (consumed < n) {
        const size_t m = stage_chunk(h, n - consumed, cap);
        const int b = h->stage_next;
        HIP_TRY(h, hipEventSynchronize(h->stage_free[b]));
        const void* up = src + consumed * kRecordBytes;
        if (!pinned_src) { staged_copy(h->pinned[b], up, m * kRecordBytes, h->cfg.copy_threads); up = h->pinned[b]; }
        // the copy runs on its own stream: the fold of the previous chunk (other buffer) is still busy on h->stream
        HIP_TRY(h, hipMemcpyAsync(h->d_stage[b], up, m * kRecordBytes, hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(h, hipEventRecord(h->stage_up[b], h->copy_stream));
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->stage_up[b], 0));
        size_t c = 0;
        rc = ingest_device_core(h, h->d_stage[b], m, &c);
        HIP_TRY(h, hipEventRecord(h->stage_free[b], h->stream));
        h->stage_next ^= 1;
        consumed += c;
        if (rc != NFAGG_OK) break;
    }
    if (pinned_src) HIP_TRY(h, hipStreamSynchronize(h->copy_stream));   // the caller's buffer is its own again when the call returns
    if (consumed_out) *consumed_out = consumed;
    return rc;
}

int nfagg_staging_acquire(nfagg_handle* h, void** buf, size_t* capacity_records) {
    if (!h || !buf) return fail(h, NFAGG_EINVAL, "null argument");
    if (h->stage_acquired) return fail(h, NFAGG_ESTATE, "staging buffer already acquired");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = staging_alloc(h);
    if (rc != NFAGG_OK) return rc;
    const int b = h->stage_next;
    HIP_TRY(h, hipEventSynchronize(h->stage_free[b]));
    *buf = h->pinned[b];
    if (capacity_records) *capacity_records = (size_t)h->cfg.staging_records;
    h->stage_acquired = true;
    return NFAGG_OK;
}

int nfagg_staging_commit(nfagg_handle* h, size_t n, size_t* consumed) {
    if (!h) return NFAGG_EINVAL;
    if (!h->stage_acquired) return fail(h, NFAGG_ESTATE, "no staging buffer acquired");
    if (n > h->cfg.staging_records) return fail(h, NFAGG_EINVAL, "n exceeds staging capacity");
    HIP_TRY(h, hipSetDevice(h->device));
    const int b = h->stage_next;
    h->stage_acquired = false;
    HIP_TRY(h, hipMemcpyAsync(h->d_stage[b], h->pinned[b], n * kRecordBytes, hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(h, hipEventRecord(h->stage_up[b], h->copy_stream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->stage_up[b], 0));
    int rc = ingest_device_core(h, h->d_stage[b], n, consumed);
    HIP_TRY(h, hipEventRecord(h->stage_free[b], h->stream));
    h->stage_next ^= 1;
    return rc;
}

int nfagg_len(nfagg_handle* h, uint64_t* entries) {
    if (!h || !entries) return fail(h, NFAGG_EINVAL, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = refresh_counters(h);
    if (rc != NFAGG_OK) return rc;
    *entries = h->live;
    return NFAGG_OK;
}

// The table itself is never cleared: the slots of the flows that just left simply belong to a past epoch tag.
// Tags hold 16 bits of epoch; when they wrap (every 65 535 bumps) the tags are cleared once.
static int bump_epoch(nfagg_handle* h) {
    uint64_t next_epoch = (h->tv.epoch_bits >> 48) + 1;
    if (next_epoch > 0xFFFFull) {
        HIP_TRY(h, hipMemsetAsync(h->tv.hot, 0, h->slots * sizeof(SlotHot), h->stream));
        next_epoch = 1;
    }
    h->tv.epoch_bits = next_epoch << 48;
    return NFAGG_OK;
}

// The bookkeeping that ends an eviction epoch (the kernels have been launched; the device counters are reset by them).
static int finish_epoch(nfagg_handle* h, int reason, uint64_t flows) {
    h->last_epoch_flows = h->tv.subflow ? h->join.claimed : flows;
    h->stats.evictions[reason]++;
    h->stats.evicted_flows[reason] += flows;
    h->epoch_seq = 0; h->seq_origin = 0; h->live = 0; h->live_ub = 0; h->must_evict = false; h->split_seq = 0; h->exported = false;
    h->counters_exact = true;          // k_reset_after_evict left n_live = 0
    h->epoch_unclustered = false;
    h->abort_cap = 0;                  // one batch with more new keys than the table takes does not cap the chunks of later epochs
    return bump_epoch(h);
}

// Accounter.evict (account.go:102-124) for a sub-flow table: the join (nfagg_dedup_join.hip) of the sub-flows that (n_shards,
// shard_id) owns into the flow-keyed table, then k_evict_dedup over that. NFAGG_TRUNCATED (nothing evicted; the join is kept for
// the repeated call) when cap is too small.
static int subflow_evict(nfagg_handle* h, int reason, uint32_t n_shards, uint32_t shard_id, void* out, bool out_is_device, size_t cap, size_t* n_out) {
    int rc;
    *n_out = 0;
    if (!h->counters_exact && (rc = refresh_counters(h)) != NFAGG_OK) return rc;
    if (reason == NFAGG_REASON_TIMEOUT && h->live_ub == 0 && !h->exported) return NFAGG_OK;      // account.go:64-66
    if ((rc = subflow_join(h, n_shards, shard_id)) != NFAGG_OK) return rc;
    const uint64_t flows = h->join.flows;
    *n_out = (size_t)flows;
    if (flows > cap) return NFAGG_TRUNCATED;
    if (flows && !out) return fail(h, NFAGG_EINVAL, "null output buffer");
    void* d_out = out;
    if (!out_is_device) {
        size_t capb = (size_t)h->d_evict_cap;
        rc = ensure_bytes(h, &h->d_evict, &capb, (size_t)flows * kRecordBytes + 16);
        h->d_evict_cap = capb;
        if (rc != NFAGG_OK) return rc;
        d_out = h->d_evict;
    }
    HIP_TRY(h, hipMemsetAsync(&h->jv.ctr->n_out, 0, sizeof(unsigned long long), h->stream));
    EventPair ep{};
    if (h->cfg.profile) prof_begin(h, ep, 1);
    hipError_t e = launch_evict_dedup(h->jv, flows, ~0ull, d_out, h->stream);      // leaves J's counters reset
    if (e == hipSuccess) e = launch_reset_counters(h->tv, h->stream);
    if (h->cfg.profile) prof_end(h, ep);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "evict launch failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipMemcpyAsync(h->h_jctr, h->jv.ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!out_is_device && flows && (rc = d2h_copy(h, out, h->d_evict, (size_t)flows * kRecordBytes)) != NFAGG_OK) return rc;
    h->mirror_fresh = false;
    if (h->h_jctr->error) return fail(h, NFAGG_EDEVICE, "sub-flow join table kernel bailed out (code %u)", h->h_jctr->error);
    if (h->h_jctr->n_out != flows)
        return fail(h, NFAGG_EDEVICE, "evict wrote %llu records, expected %llu", (unsigned long long)h->h_jctr->n_out, (unsigned long long)flows);
    h->join.valid = false; h->join.dirty = false;                 // launch_evict_dedup left J's counters reset; its slots expire with the tag
    if ((rc = subflow_join_table_next_epoch(h)) != NFAGG_OK) return rc;
    return finish_epoch(h, reason, flows);
}

static int evict_core(nfagg_handle* h, int reason, void* out, bool out_is_device, size_t cap, size_t* n_out) {
    if (!h || !n_out || reason < 0 || reason > 2) return fail(h, NFAGG_EINVAL, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = NFAGG_OK;
    if (h->exported) return fail(h, NFAGG_ESTATE, "partials were exported from / merged into this table: evict it with nfagg_evict_owned_device");
    if (h->tv.subflow) return subflow_evict(h, reason, 1, 0, out, out_is_device, cap, n_out);
    if (!h->counters_exact && (rc = refresh_counters(h)) != NFAGG_OK) return rc;
    const uint64_t legit = h->live;
    const uint64_t claimed = h->live_ub;          // = the device's n_live (counters_exact)
    *n_out = (size_t)legit;
    if (legit > cap) return NFAGG_TRUNCATED;
    if (legit && !out) return fail(h, NFAGG_EINVAL, "null output buffer");
    if (reason == NFAGG_REASON_TIMEOUT && legit == 0 && claimed == 0) return NFAGG_OK;  // account.go:64-66
    void* d_out = out;
    if (!out_is_device) {
        size_t capb = (size_t)h->d_evict_cap;
        rc = ensure_bytes(h, &h->d_evict, &capb, (size_t)legit * kRecordBytes + 16);
        h->d_evict_cap = capb;
        if (rc != NFAGG_OK) return rc;
        d_out = h->d_evict;
    }
    HIP_TRY(h, hipMemsetAsync(&h->tv.ctr->n_out, 0, sizeof(unsigned long long), h->stream));
    // large table: visit the claimed slots in address order (see launch_sort_slots)
    TableView tv = h->tv;
    // ... unless every batch of the epoch went through the two-pass fold: its pass-2 workgroups claim slots partition by
    // partition (partition = top bits of the slot index), the live list is clustered already and the sort does not pay
    // (0.27 ms unsorted against 0.40 ms sorted per 1 M flows of a 64 GiB table; 0.49 ms before either).
    bool sort_slots = claimed >= (1u << 16) && h->slots >= (1ull << 24) && claimed < (1ull << 31) && h->epoch_unclustered;
    size_t temp_bytes = 0;
    if (sort_slots) {
        int bits = 0;
        while ((1ull << bits) < h->slots) bits++;
        h->sort_bits = bits;
        hipError_t es = launch_sort_slots(h->tv.live_list, nullptr, claimed, bits, nullptr, &temp_bytes, h->stream);
        if (es != hipSuccess) return fail(h, NFAGG_EDEVICE, "sort size query failed: %s", hipGetErrorString(es));
        if ((rc = ensure_bytes(h, &h->d_sort[0], &h->d_sort_cap[0], claimed * sizeof(uint32_t))) != NFAGG_OK) return rc;
        if ((rc = ensure_bytes(h, &h->d_sort[1], &h->d_sort_cap[1], temp_bytes + 16)) != NFAGG_OK) return rc;
    }
    EventPair ep{};
    if (h->cfg.profile) prof_begin(h, ep, 1);
    if (sort_slots) {
        hipError_t es = launch_sort_slots(h->tv.live_list, (uint32_t*)h->d_sort[0], claimed, h->sort_bits, h->d_sort[1], &temp_bytes, h->stream);
        if (es != hipSuccess) return fail(h, NFAGG_EDEVICE, "slot sort failed: %s", hipGetErrorString(es));
        tv.live_list = (uint32_t*)h->d_sort[0];
    }
    hipError_t e = launch_evict(tv, claimed, h->must_evict ? h->split_seq - h->seq_origin : ~0ull, d_out, h->stream);
    if (h->cfg.profile) prof_end(h, ep);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "evict launch failed: %s", hipGetErrorString(e));
    // counters and (host variant) the records come back in stream order behind the kernel: one wait for both
    HIP_TRY(h, hipMemcpyAsync(h->h_ctr, h->tv.ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!out_is_device && legit && (rc = d2h_copy(h, out, h->d_evict, (size_t)legit * kRecordBytes)) != NFAGG_OK) return rc;
    h->mirror_fresh = true;
    if (h->h_ctr->error)
        return fail(h, NFAGG_EDEVICE, "flow table kernel bailed out (code %u: 1 = probe overflow / table too small, 2 = claim spin limit, 5 = spill overflow list full, 6 = claimed slot without a record, 7 = partial of another shard merged)", h->h_ctr->error);
    if (h->h_ctr->n_out != legit)
        return fail(h, NFAGG_EDEVICE, "evict wrote %llu records, expected %llu", (unsigned long long)h->h_ctr->n_out, (unsigned long long)legit);
    h->stats.records_skipped = h->h_ctr->n_skipped;
    h->stats.records_bypassed = h->h_ctr->n_bypassed;
    if (h->h_ctr->max_probe > h->stats.max_probe) h->stats.max_probe = h->h_ctr->max_probe;
    return finish_epoch(h, reason, legit);
}

int nfagg_evict(nfagg_handle* h, int reason, void* out, size_t cap, size_t* n_out) {
    return evict_core(h, reason, out, false, cap, n_out);
}

int nfagg_evict_device(nfagg_handle* h, int reason, void* d_out, size_t cap, size_t* n_out) {
    if (d_out && ((uintptr_t)d_out & 15u)) return fail(h, NFAGG_EINVAL, "device output must be 16-byte aligned");
    return evict_core(h, reason, d_out, true, cap, n_out);
}

// ---------------------------------------------------------------- local fold across GPUs: partials (nfagg_combine.hip)
// The three steps of the local-fold tick as handle calls. The one-process group (nfagg_group.inc) and one-process-per-GPU
// ranks (bench.py --gpus N: exchange with an RCCL all-to-all) run the same code.
static int partials_scratch(nfagg_handle* h) {
    int rc = ensure_bytes(h, &h->d_exp, &h->d_exp_cap, 136 * sizeof(unsigned long long));
    if (rc != NFAGG_OK) return rc;
    if (!h->h_exp) HIP_TRY(h, hipHostMalloc((void**)&h->h_exp, 136 * sizeof(unsigned long long), hipHostMallocDefault));
    return NFAGG_OK;
}

// counts[o] (host, n_shards words) = flows of segment o; *n_out = their sum. NFAGG_TRUNCATED (nothing written, state unchanged)
// when cap is too small. The partials are complete in d_out when the call returns.
static int partials_export_core(nfagg_handle* h, uint32_t n_shards, uint32_t self_shard, void* d_out, size_t cap, uint64_t* counts,
                                size_t* n_out) {
    if (!h || !counts || !n_out) return fail(h, NFAGG_EINVAL, "null argument");
    if (n_shards == 0 || n_shards > 64) return fail(h, NFAGG_EINVAL, "n_shards must be in [1, 64]");
    if (h->cfg.mode != NFAGG_MODE_ACCOUNTER && !h->tv.subflow)
        return fail(h, NFAGG_EINVAL, "partials need NFAGG_MODE_ACCOUNTER, or NFAGG_MODE_KERNEL_DEDUP on a handle created with nfagg_config.local_fold (flow-keyed kernel-dedup slots do not merge across tables)");
    if (h->cfg.n_shards > 1) return fail(h, NFAGG_ESTATE, "a handle that filters its input by shard (n_shards > 1) holds no other shard's flows");
    if (d_out && ((uintptr_t)d_out & 63u)) return fail(h, NFAGG_EINVAL, "partials buffer must be 64-byte aligned");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = partials_scratch(h);
    if (rc != NFAGG_OK) return rc;
    if ((rc = refresh_counters(h)) != NFAGG_OK) return rc;
    const uint64_t claimed = h->h_ctr->n_live;
    const uint64_t seq_limit = h->must_evict ? h->split_seq - h->seq_origin : ~0ull;
    unsigned long long* d_counts = (unsigned long long*)h->d_exp;
    unsigned long long* d_cursor = d_counts + 64;
    hipError_t e = launch_export_count(h->tv, claimed, seq_limit, n_shards, self_shard, d_counts, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "partials count launch failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipMemcpyAsync(h->h_exp, d_counts, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    uint64_t total = 0;
    for (uint32_t o = 0; o < n_shards; o++) { counts[o] = h->h_exp[o]; h->h_exp[64 + o] = total; total += counts[o]; }
    *n_out = (size_t)total;
    if (total > cap) return NFAGG_TRUNCATED;
    if (total && !d_out) return fail(h, NFAGG_EINVAL, "null partials buffer");
    if (total) {
        HIP_TRY(h, hipMemcpyAsync(d_cursor, h->h_exp + 64, 64 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
        e = launch_export_scatter(h->tv, claimed, seq_limit, n_shards, self_shard, d_cursor, d_out, h->stream);
        if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "partials export launch failed: %s", hipGetErrorString(e));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if (self_shard < n_shards && n_shards > 1) h->exported = true;       // other shards' flows are on their way to their owners
    return NFAGG_OK;
}

// Asynchronous (the handle's stream); d_partials must stay valid until the handle synchronises.
static int partials_merge_core(nfagg_handle* h, uint32_t n_shards, uint32_t shard_id, const void* d_partials, size_t n) {
    if (!h || (n && !d_partials)) return fail(h, NFAGG_EINVAL, "null argument");
    if (n_shards == 0 || n_shards > 64 || shard_id >= n_shards) return fail(h, NFAGG_EINVAL, "bad shard (n_shards in [1, 64], shard_id < n_shards)");
    if (h->cfg.mode != NFAGG_MODE_ACCOUNTER && !h->tv.subflow) return fail(h, NFAGG_EINVAL, "partials need NFAGG_MODE_ACCOUNTER or nfagg_config.local_fold");
    if ((uintptr_t)d_partials & 15u) return fail(h, NFAGG_EINVAL, "partials buffer must be 16-byte aligned");
    if (n == 0) return NFAGG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    { const int rcj = subflow_join_discard(h); if (rcj != NFAGG_OK) return rcj; }
    TableView tv = h->tv;
    tv.n_shards = n_shards; tv.shard_id = shard_id;
    h->counters_exact = false; h->mirror_fresh = false;                  // the merge claims slots
    h->exported = true;                                                  // merged state: only the eviction may follow
    const hipError_t e = launch_merge_raw(tv, d_partials, n, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "partials merge launch failed: %s", hipGetErrorString(e));
    return NFAGG_OK;
}

// *owned = flows of this table that (n_shards, shard_id) owns — what evict_owned_core will write. Synchronises; reports
// claims refused by a merge.
static int owned_count_core(nfagg_handle* h, uint32_t n_shards, uint32_t shard_id, uint64_t* owned, uint64_t* claimed_out) {
    HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if (h->tv.subflow) {                                         // flows, not sub-flows: known after the join (kept for the eviction)
        if ((rc = subflow_join(h, n_shards, shard_id)) != NFAGG_OK) return rc;
        *owned = h->join.flows;
        if (claimed_out) *claimed_out = h->join.claimed;
        return NFAGG_OK;
    }
    rc = partials_scratch(h);
    if (rc != NFAGG_OK) return rc;
    if ((rc = refresh_counters(h)) != NFAGG_OK) return rc;
    if (h->h_ctr->aborted) return fail(h, NFAGG_EDEVICE, "table too small for the flows this shard owns (claims refused while merging)");
    const uint64_t claimed = h->h_ctr->n_live;
    TableView tv = h->tv;
    tv.n_shards = n_shards; tv.shard_id = shard_id;
    unsigned long long* d_owned = (unsigned long long*)h->d_exp + 128;
    const hipError_t e = launch_count_owned(tv, claimed, h->must_evict ? h->split_seq - h->seq_origin : ~0ull, d_owned, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "owned count launch failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipMemcpyAsync(h->h_exp + 128, d_owned, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *owned = h->h_exp[128];
    if (claimed_out) *claimed_out = claimed;
    return NFAGG_OK;
}

static int evict_owned_core(nfagg_handle* h, int reason, uint32_t n_shards, uint32_t shard_id, void* d_out, size_t cap, size_t* n_out) {
    if (!h || !n_out || reason < 0 || reason > 2) return fail(h, NFAGG_EINVAL, "bad argument");
    if (n_shards == 0 || n_shards > 64 || shard_id >= n_shards) return fail(h, NFAGG_EINVAL, "bad shard (n_shards in [1, 64], shard_id < n_shards)");
    if (h->cfg.mode != NFAGG_MODE_ACCOUNTER && !h->tv.subflow) return fail(h, NFAGG_EINVAL, "partials need NFAGG_MODE_ACCOUNTER or nfagg_config.local_fold");
    if (d_out && ((uintptr_t)d_out & 15u)) return fail(h, NFAGG_EINVAL, "device output must be 16-byte aligned");
    if (h->tv.subflow) { HIP_TRY(h, hipSetDevice(h->device)); return subflow_evict(h, reason, n_shards, shard_id, d_out, true, cap, n_out); }
    uint64_t owned = 0, claimed = 0;
    int rc = owned_count_core(h, n_shards, shard_id, &owned, &claimed);
    if (rc != NFAGG_OK) return rc;
    *n_out = (size_t)owned;
    if (owned > cap) return NFAGG_TRUNCATED;
    if (owned && !d_out) return fail(h, NFAGG_EINVAL, "null output buffer");
    if (reason == NFAGG_REASON_TIMEOUT && claimed == 0 && !h->exported) return NFAGG_OK;      // account.go:64-66
    HIP_TRY(h, hipMemsetAsync(&h->tv.ctr->n_out, 0, sizeof(unsigned long long), h->stream));
    TableView tv = h->tv;
    tv.n_shards = n_shards; tv.shard_id = shard_id;
    EventPair ep{};
    if (h->cfg.profile) prof_begin(h, ep, 1);
    const hipError_t e = launch_evict_filtered(tv, claimed, h->must_evict ? h->split_seq - h->seq_origin : ~0ull, d_out, h->stream);
    if (h->cfg.profile) prof_end(h, ep);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "evict launch failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipMemcpyAsync(h->h_ctr, h->tv.ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->h_ctr->n_out != owned)
        return fail(h, NFAGG_EDEVICE, "evict wrote %llu records, expected %llu", (unsigned long long)h->h_ctr->n_out, (unsigned long long)owned);
    return finish_epoch(h, reason, owned);
}

int nfagg_partials_export_device(nfagg_handle* h, uint32_t n_shards, uint32_t self_shard, void* d_out, size_t cap, uint64_t* counts, size_t* n_out) {
    return partials_export_core(h, n_shards, self_shard, d_out, cap, counts, n_out);
}

int nfagg_partials_merge_device(nfagg_handle* h, uint32_t n_shards, uint32_t shard_id, const void* d_partials, size_t n) {
    return partials_merge_core(h, n_shards, shard_id, d_partials, n);
}

int nfagg_evict_owned_device(nfagg_handle* h, int reason, uint32_t n_shards, uint32_t shard_id, void* d_out, size_t cap, size_t* n_out) {
    return evict_owned_core(h, reason, n_shards, shard_id, d_out, cap, n_out);
}

size_t nfagg_partial_bytes(const nfagg_handle* h) { return (h && h->tv.subflow) ? kPartialBytesDedup : kPartialBytes; }

int nfagg_set_sequence(nfagg_handle* h, uint64_t next_seq) {
    if (!h) return NFAGG_EINVAL;
    if (next_seq < h->epoch_seq) return fail(h, NFAGG_EINVAL, "sequence numbers do not go backwards inside an epoch (%llu < %llu)",
                                             (unsigned long long)next_seq, (unsigned long long)h->epoch_seq);
    h->epoch_seq = next_seq;
    h->ext_sequenced = true;                                    // other tables number their records in the same space
    return NFAGG_OK;
}

// ---- the sequence window of tables that share ONE numbering (local fold across GPUs) ---------------------------------
// Such a table cannot rebase its tags on its own (nfagg_rebase.hip): the flows are first brought together at their owners —
// every table exports ALL its flows as partials grouped by owner (nfagg_partials_export_device with NFAGG_SHARD_NONE),
// empties itself (window_clear_core: the epoch tag, nothing is written), takes in what it owns (partials_merge_core, its own
// segment included) and rebases the single copy it now holds (window_finish_core). No flow leaves the epoch, the sketches are
// not touched; len(entries) of a table becomes the number of flows it owns.
static int window_clear_core(nfagg_handle* h) {
    if (h->must_evict) return fail(h, NFAGG_ESTATE, "an eviction on full is pending");
    HIP_TRY(h, hipSetDevice(h->device));
    const hipError_t e = launch_reset_counters(h->tv, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "counter reset launch failed: %s", hipGetErrorString(e));
    h->live = h->live_ub = 0; h->counters_exact = true; h->mirror_fresh = false;
    h->epoch_unclustered = false; h->exported = false;
    return bump_epoch(h);
}

static int window_finish_core(nfagg_handle* h, uint64_t next_seq) {
    if (next_seq < h->epoch_seq || next_seq < rebase_keep()) return fail(h, NFAGG_EINVAL, "window restart: next sequence number %llu below what the handle has reached",
                                                                         (unsigned long long)next_seq);
    HIP_TRY(h, hipSetDevice(h->device));
    hipError_t e = launch_finalize(h->tv, nullptr, 0, 0, h->stream);          // the merges copied the identity dwords: everything is finalized
    if (e == hipSuccess) e = launch_rebase(h->tv, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "window restart launch failed: %s", hipGetErrorString(e));
    h->epoch_seq = next_seq; h->seq_origin = next_seq - rebase_keep();
    h->exported = false; h->counters_exact = false; h->mirror_fresh = false;
    h->stats.sequence_rebases++;
    const int rc = refresh_counters(h);                                        // len(entries) = the flows this table owns now; reports refused claims
    if (rc != NFAGG_OK) return rc;
    if (h->h_ctr->aborted) return fail(h, NFAGG_EDEVICE, "table too small for the flows this shard owns (claims refused while merging)");
    h->epoch_unclustered = true;
    return NFAGG_OK;
}

int nfagg_window_restart_device(nfagg_handle* h, uint32_t n_shards, uint32_t shard_id, const void* d_partials, size_t n, uint64_t next_seq) {
    if (!h) return NFAGG_EINVAL;
    if (h->tv.subflow) return fail(h, NFAGG_ESTATE, "kernel-dedup mode: the sequence window of a local-fold job does not move (the order between a flow's sub-flows would be lost): evict instead");
    int rc = window_clear_core(h);
    if (rc != NFAGG_OK) return rc;
    if ((rc = partials_merge_core(h, n_shards, shard_id, d_partials, n)) != NFAGG_OK) return rc;
    h->ext_sequenced = true;
    return window_finish_core(h, next_seq);
}

// ---------------------------------------------------------------- nfagg_account*: the record arm WITH its evictions on "full"
constexpr uint64_t kAccountFastMaxEntries = 32768;   // beyond that an epoch is long enough for the optimistic fold of nfagg_ingest

constexpr uint32_t kVariantAccountChain = 30;        // ingest_variant (tests): nfagg_account always takes the kernel chain

// May the device-resident epoch loop (the kernel chain, nfagg_epoch_chain.hip) take this batch?
static bool account_fast_ok(const nfagg_handle* h, size_t n, size_t out_cap) {
    // (a handle that filters its input by shard takes the host-driven loop: the chain counts the records it skips per window, and a
    // window that ends on "full" is looked at again from the split on)
    return h->cfg.mode == NFAGG_MODE_ACCOUNTER && h->cfg.n_shards <= 1 && h->cfg.max_entries <= kAccountFastMaxEntries && !h->must_evict && !h->exported &&
           h->cfg.max_entries + chain_window() + 16 <= h->tv.claim_limit && out_cap >= h->cfg.max_entries &&
           h->epoch_seq - h->seq_origin + n < kSeqWindow - chain_window() && (h->tv.epoch_bits >> 48) < 0xFFFFull;
}

// One pass of the kernel chain (nfagg_epoch_chain.hip) over d[0..n). *consumed / *n_ep / *n_out: records consumed, evictions
// performed, records written to d_out (epoch e ends at epoch_end[e] records); *stop as the control block reports it.
// Windows are enqueued kChainWindows[k] at a time, the control block is read back after each batch of launches.
// Windows per launch of the chain. A window takes up to chain_window() records and ends early where an epoch ends, so a call of n
// records needs about n / chain_window() + its evictions + 1 of them; the kernels of windows beyond the call's end find `stop` set
// and return at once — but a launch of nothing still costs ~2 us, and a shim that drains a 50-slot channel
// (pkg/agent/agent.go:408, pkg/config/config.go:134) calls with 1 ... a few thousand records: round 5 replayed 24 windows = 96
// launches for every call, 230 us for ONE record (profiles/r06_account_small_calls.txt). Three graphs, the shortest that covers
// what is left is replayed.
constexpr int kChainWindows[3] = {2, 6, 24};
constexpr size_t kChainEndsInline = 32;                 // epoch ends that come back with the control block (more: a second copy)

static int account_chain_launch(nfagg_handle* h, const void* d, size_t n, void* d_out, size_t out_cap, uint64_t* epoch_end, size_t max_epochs,
                                size_t* consumed, size_t* n_ep, size_t* n_out, uint32_t* stop) {
    int rc;
    // len(entries), n_live: the host knows them after an eviction and after the last chain launch (counters_exact) — one round
    // trip less per call; otherwise asked for
    if (!h->counters_exact) {
        if ((rc = refresh_counters(h)) != NFAGG_OK) return rc;
        if (h->h_ctr->n_finalized != h->h_ctr->n_live) return fail(h, NFAGG_ESTATE, "account: unfinalized slots at the start of a batch");
    }
    const uint64_t n_live0 = h->live_ub;
    const uint32_t me = max_epochs > 0xFFFFu ? 0xFFFFu : (uint32_t)max_epochs;
    const size_t ctlb = (chain_ctl_bytes() + 63) & ~(size_t)63;
    if ((rc = ensure_bytes(h, &h->d_ep[0], &h->d_ep_cap[0], ctlb)) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_ep[1], &h->d_ep_cap[1], (size_t)(me + 1) * sizeof(uint64_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_ep[2], &h->d_ep_cap[2], (size_t)(h->cfg.max_entries + chain_window() + 64) * sizeof(uint32_t))) != NFAGG_OK) return rc;
    const size_t hb = ctlb + (size_t)(me + 1) * sizeof(uint64_t);
    if (h->h_ep_cap < hb) {
        if (h->h_ep) { hipHostFree(h->h_ep); h->h_ep = nullptr; h->h_ep_cap = 0; }
        HIP_TRY(h, hipHostMalloc(&h->h_ep, hb + 4096, hipHostMallocDefault));
        h->h_ep_cap = hb + 4096;
    }
    const uint64_t seq_start = h->epoch_seq - h->seq_origin;                      // window-relative, as the slots carry it
    chain_ctl_fill(h->h_ep, d, d_out, n, seq_start, h->live, n_live0, out_cap, h->tv.epoch_bits, h->cfg.max_entries, me);
    HIP_TRY(h, hipMemcpyAsync(h->d_ep[0], h->h_ep, chain_ctl_bytes(), hipMemcpyHostToDevice, h->stream));
    // kChainWindows[k] windows = 4 x that many launches whose arguments are the same in every call: captured into graphs once, one
    // hipGraphLaunch per batch afterwards (eager launches cost the host ~8 us each here: the chain was host-bound)
    if (!h->ep_graph_off && (!h->ep_graph[0] || h->ep_graph_key[0] != h->d_ep[0] || h->ep_graph_key[1] != h->d_ep[1] || h->ep_graph_key[2] != h->d_ep[2])) {
        hipError_t ec = hipSuccess;
        for (int k = 0; k < 3; k++) {
            if (h->ep_graph[k]) { hipGraphExecDestroy(h->ep_graph[k]); h->ep_graph[k] = nullptr; }
            hipGraph_t g = nullptr;
            if (ec == hipSuccess) ec = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
            else continue;
            if (ec == hipSuccess) {
                for (int w = 0; w < kChainWindows[k] && ec == hipSuccess; w++)
                    ec = launch_epoch_chain_window(h->tv, h->sk, h->d_ep[0], (uint32_t*)h->d_ep[2], (uint64_t*)h->d_ep[1], h->stream);
                const hipError_t ee = hipStreamEndCapture(h->stream, &g);
                if (ec == hipSuccess) ec = ee;
                if (ec == hipSuccess) ec = hipGraphInstantiate(&h->ep_graph[k], g, nullptr, nullptr, 0);
                if (g) hipGraphDestroy(g);
            }
        }
        if (ec != hipSuccess) {
            // no graph (a capture that something in the process invalidated, an instantiation that failed): this handle launches
            // its windows eagerly from now on — slower (~8 us of host time per launch), never wrong
            (void)hipGetLastError();
            for (int k = 0; k < 3; k++) if (h->ep_graph[k]) { hipGraphExecDestroy(h->ep_graph[k]); h->ep_graph[k] = nullptr; }
            h->ep_graph_off = true;
        } else {
            h->ep_graph_key[0] = h->d_ep[0]; h->ep_graph_key[1] = h->d_ep[1]; h->ep_graph_key[2] = h->d_ep[2];
        }
    }
    EventPair ep{};
    if (h->cfg.profile) prof_begin(h, ep, 0);
    uint64_t v[9];
    uint64_t left = n, ends_inline = 0;
    h->counters_exact = false; h->mirror_fresh = false;
    for (;;) {
        // windows this launch should hold: what is left in whole windows, one for every eviction an epoch of max_entries flows can
        // end in (each ends its window early), one to spare
        const uint64_t want = (left + chain_window() - 1) / chain_window() + left / (h->cfg.max_entries ? h->cfg.max_entries : 1) + 1;
        int k = 0;
        while (k < 2 && (uint64_t)kChainWindows[k] < want) k++;
        if (h->ep_graph[k]) HIP_TRY(h, hipGraphLaunch(h->ep_graph[k], h->stream));
        else {
            for (int w = 0; w < kChainWindows[k]; w++) {
                const hipError_t el = launch_epoch_chain_window(h->tv, h->sk, h->d_ep[0], (uint32_t*)h->d_ep[2], (uint64_t*)h->d_ep[1], h->stream);
                if (el != hipSuccess) return fail(h, NFAGG_EDEVICE, "epoch chain launch failed: %s", hipGetErrorString(el));
            }
        }
        // the control block, the first epoch ends and the counters behind ONE wait (round 5: three)
        ends_inline = me < kChainEndsInline ? me : kChainEndsInline;
        HIP_TRY(h, hipMemcpyAsync(h->h_ep, h->d_ep[0], chain_ctl_bytes(), hipMemcpyDeviceToHost, h->stream));
        if (ends_inline) HIP_TRY(h, hipMemcpyAsync((char*)h->h_ep + ctlb, h->d_ep[1], ends_inline * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        if ((rc = refresh_counters_enqueue(h)) != NFAGG_OK) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        chain_ctl_read(h->h_ep, v);
        if (v[6] != 0) break;
        left = n - v[0];
    }
    if (h->cfg.profile) prof_end(h, ep);
    const uint64_t pos = v[0], seq = v[1], live = v[2], out_pos = v[3], n_epochs = v[5];
    if (n_epochs > ends_inline) {
        HIP_TRY(h, hipMemcpyAsync((char*)h->h_ep + ctlb + ends_inline * sizeof(uint64_t), (const uint64_t*)h->d_ep[1] + ends_inline,
                                  (size_t)(n_epochs - ends_inline) * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if ((rc = refresh_counters_taken(h)) != NFAGG_OK) return rc;
    if (h->h_ctr->n_live != live)
        return fail(h, NFAGG_EDEVICE, "epoch chain: %llu claimed slots, len(entries) %llu", (unsigned long long)h->h_ctr->n_live, (unsigned long long)live);
    for (uint64_t k = 0; k < n_epochs && k < max_epochs; k++) epoch_end[k] = ((const uint64_t*)((const char*)h->h_ep + ctlb))[k];
    h->tv.epoch_bits = v[4];
    // identity dwords of the slots the epoch in progress claimed in this call, from the batch
    const bool began_here = v[8] != 0;
    const uint64_t first_rec = began_here ? v[7] : 0, seq_at_first = began_here ? 0 : seq_start;
    const hipError_t e = launch_finalize(h->tv, (const char*)d + first_rec * kRecordBytes, pos - first_rec, seq_at_first, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "finalize launch failed: %s", hipGetErrorString(e));
    if (n_epochs) h->seq_origin = 0;                            // an eviction inside the call restarted the epoch: a fresh window
    h->epoch_seq = h->seq_origin + seq; h->live = h->live_ub = live;
    h->counters_exact = true;
    h->mirror_fresh = false;                                    // (k_finalize moves n_finalized behind the mirror's back)
    h->epoch_unclustered = true;
    h->stats.records_ingested += pos;
    h->stats.evictions[NFAGG_REASON_FULL] += n_epochs;
    h->stats.evicted_flows[NFAGG_REASON_FULL] += out_pos;
    *consumed = (size_t)pos; *n_ep = (size_t)n_epochs; *n_out = (size_t)out_pos; *stop = (uint32_t)v[6];
    return NFAGG_OK;
}

#ifdef NFAGG_DIAG
#include <chrono>
static double diag_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define NF_DIAG_T(var) const double var = diag_now_ms()
#define NF_DIAG_PRINT(...) fprintf(stderr, __VA_ARGS__)
#else
#define NF_DIAG_T(var)
#define NF_DIAG_PRINT(...)
#endif

#include "nfagg_account_par.inc"

// d_out: DEVICE. epoch_end: HOST. Evictions append to d_out; *n_epochs of them, the e-th ends at record epoch_end[e].
static int account_device_core(nfagg_handle* h, const void* d_records, size_t n, void* d_out, size_t out_cap, uint64_t* epoch_end,
                               size_t max_epochs, size_t* n_epochs_out, size_t* consumed_out) {
    size_t consumed = 0, n_ep = 0, out_pos = 0;
    int rc = NFAGG_OK;
    const char* base = static_cast<const char*>(d_records);
    char* obase = static_cast<char*>(d_out);
    auto evict_pending = [&]() -> int {                         // the map is full and the record that found it so is next (account.go:85-94)
        if (n_ep >= max_epochs) return NFAGG_TRUNCATED;
        size_t got = 0;
        const int r = evict_core(h, NFAGG_REASON_FULL, obase + out_pos * kRecordBytes, true, out_cap - out_pos, &got);
        if (r != NFAGG_OK) return r;                            // NFAGG_TRUNCATED: nothing evicted, the caller drains `out` and calls again
        out_pos += got;
        epoch_end[n_ep++] = out_pos;
        return NFAGG_OK;
    };
    if (h->must_evict && n) rc = evict_pending();
    bool par_declined = false;
    while (rc == NFAGG_OK && consumed < n) {
        if (!par_declined && account_par_ok(h, n - consumed, out_cap - out_pos) && n_ep < max_epochs) {
            // the epochs of the call found first, then folded all at once (nfagg_account_par.inc)
            size_t c = 0, e = 0, o = 0; uint32_t stop = 0;
            rc = account_par_launch(h, base + consumed * kRecordBytes, n - consumed, obase + out_pos * kRecordBytes, out_cap - out_pos,
                                    epoch_end + n_ep, max_epochs - n_ep, &c, &e, &o, &stop);
            for (size_t k = 0; k < e; k++) epoch_end[n_ep + k] += out_pos;
            consumed += c; n_ep += e; out_pos += o;
            if (rc == kParDeclined) { rc = NFAGG_OK; par_declined = true; h->stats.account_declined++; continue; }
            if (rc == NFAGG_OK) h->stats.account_epochs_first++;
            if (rc != NFAGG_OK) break;
            if (stop == 2) { rc = NFAGG_TRUNCATED; break; }
            continue;
        }
        if (account_fast_ok(h, n - consumed, out_cap - out_pos) && n_ep < max_epochs) {
            size_t c = 0, e = 0, o = 0; uint32_t stop = 0;
            h->stats.account_chain++;
            rc = account_chain_launch(
                h, base + consumed * kRecordBytes, n - consumed, obase + out_pos * kRecordBytes, out_cap - out_pos,
                epoch_end + n_ep, max_epochs - n_ep, &c, &e, &o, &stop);
            if (rc != NFAGG_OK) break;
            for (size_t k = 0; k < e; k++) epoch_end[n_ep + k] += out_pos;
            consumed += c; n_ep += e; out_pos += o;
            if (stop == 2) { rc = NFAGG_TRUNCATED; break; }     // no room for another eviction
            if (stop != 3) continue;                            // 3: the epoch tags wrap at the next eviction — that epoch goes through the host path below
        }
        size_t c = 0;
        rc = ingest_device_core(h, base + consumed * kRecordBytes, n - consumed, &c);
        consumed += c;
        if (rc == NFAGG_FULL) rc = evict_pending();
    }
    if (n_epochs_out) *n_epochs_out = n_ep;
    if (consumed_out) *consumed_out = consumed;
    return rc;
}

int nfagg_account_device(nfagg_handle* h, const void* d_records, size_t n, void* d_out, size_t out_cap, uint64_t* epoch_end,
                         size_t max_epochs, size_t* n_epochs, size_t* consumed) {
    if (!h || (!d_records && n) || !epoch_end || !n_epochs || !consumed || (!d_out && out_cap)) return fail(h, NFAGG_EINVAL, "null argument");
    if ((((uintptr_t)d_records | (uintptr_t)d_out) & 15u) != 0) return fail(h, NFAGG_EINVAL, "device buffers must be 16-byte aligned");
    HIP_TRY(h, hipSetDevice(h->device));
    const int rc = account_device_core(h, d_records, n, d_out, out_cap, epoch_end, max_epochs, n_epochs, consumed);
    // synchronous, as the header says: the last launches of the call (k_finalize copies the new flows' identity dwords out of
    // d_records; a fall-through to the plain fold is asynchronous altogether) have read the caller's buffer when it returns
    if (rc >= 0) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return rc;
}

int nfagg_account(nfagg_handle* h, const void* records, size_t n, void* out, size_t out_cap, uint64_t* epoch_end, size_t max_epochs,
                  size_t* n_epochs_out, size_t* consumed_out) {
    if (!h || (!records && n) || !epoch_end || !n_epochs_out || !consumed_out || (!out && out_cap)) return fail(h, NFAGG_EINVAL, "null argument");
    if (h->stage_acquired) return fail(h, NFAGG_ESTATE, "a staging buffer is acquired; commit it first");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = staging_alloc(h);
    if (rc != NFAGG_OK) return rc;
    const char* src = static_cast<const char*>(records);
    const size_t cap = (size_t)h->cfg.staging_records;
    size_t consumed = 0, n_ep = 0, out_pos = 0;
    // Pinned ring, double buffered as in nfagg_ingest: chunk k+1 is copied into its pinned buffer and sent up on the copy
    // stream while the device works on chunk k. A chunk that stops early (no room for another eviction) ends the call.
    // The first chunks are short (a quarter, then half a staging buffer): nothing overlaps the first upload, so it should not be a
    // whole buffer's (2.6 ms of a 20 ms call for 8 M records over a 57 GB/s link).
    size_t staged_lo[2] = {0, 0}, staged_n[2] = {0, 0};
    const bool pinned_src = host_is_pinned(records);            // page-locked caller buffer: sent up as it is, no host copy
    const bool pinned_out = out_cap != 0 && host_is_pinned(out); // page-locked output: the evictions come down by DMA, asynchronously
    size_t n_staged = 0;
    // A call of about one staging buffer (the shim's batch: up to 1 Mi records) goes in up to four equal pieces, none shorter than
    // the epochs-found-first path takes: the first upload is a quarter of the call, the others overlap the device's work
    // (4.6 -> ~3.4 ms per 1 Mi-record call, profiles/r06_account_small_calls.txt).
    const size_t par_min = (size_t)account_par_min_records(h->cfg.max_entries);
    size_t piece = n;
    {
        size_t k = 4;
        while (k > 1 && n / k < (par_min > 131072 ? par_min : (size_t)131072)) k--;
        piece = ((n + k - 1) / k + 63) & ~(size_t)63;
    }
    auto stage = [&](int b, size_t lo) -> int {
        size_t want = cap;
        if (n > cap + cap / 2) { if (n_staged < 2) want = cap >> (2 - n_staged); }  // a long call ramps up: cap / 4, cap / 2, cap, cap, ...
        else want = piece;                                                          // a call of about one staging buffer: in up to four equal pieces
        // (never so short that the chunk leaves the epochs-found-first path, nor longer than a staging buffer)
        if (want < par_min) want = par_min;
        if (want > cap) want = cap;
        n_staged++;
        const size_t m = (n - lo) < want ? (n - lo) : want;
        HIP_TRY(h, hipEventSynchronize(h->stage_free[b]));
        const void* up = src + lo * kRecordBytes;
        if (!pinned_src) { staged_copy(h->pinned[b], up, m * kRecordBytes, h->cfg.copy_threads); up = h->pinned[b]; }
        HIP_TRY(h, hipMemcpyAsync(h->d_stage[b], up, m * kRecordBytes, hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(h, hipEventRecord(h->stage_up[b], h->copy_stream));
        staged_lo[b] = lo; staged_n[b] = m;
        return NFAGG_OK;
    };
    int b = h->stage_next;
    if (n && (rc = stage(b, 0)) != NFAGG_OK) return rc;
    // Three things overlap per chunk k: this thread drives the device over chunk k (synchronous); the evictions of chunk k-1
    // come down into the caller's buffer; chunk k+1 goes up. Two device buffers take the evictions in turn; a chunk of m records
    // delivers at most m + max_entries flows.
    //   page-locked output: an asynchronous copy on a stream of its own, waited for only when its device buffer comes round
    //     again (a blocking hipMemcpy from a helper thread was served only after the upload in flight: 3 ms of every second
    //     chunk, 27.6 ms for the 8 M-record call where the link needs 20.2 — profiles/r05_account_host_pipeline.txt);
    //   pageable output: a helper thread (pinned bounce buffers + host copies, d2h_copy).
    if (pinned_out) {
        if (!h->d2h_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->d2h_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++)
            if (!h->bounce_ev[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->bounce_ev[k], hipEventDisableTiming));
    }
    bool down_pending[2] = {false, false};                      // pinned_out: a download from eviction buffer [k] is in flight
    struct { bool has = false; const void* d = nullptr; size_t got = 0, at = 0; } prev;
    auto bring_down = [&]() -> int {
        int rd = NFAGG_OK;
        if (prev.has && prev.got) rd = d2h_copy(h, (char*)out + prev.at * kRecordBytes, prev.d, prev.got * kRecordBytes);
        prev.has = false;
        return rd;
    };
    int ebuf = 0;
    while (rc == NFAGG_OK && consumed < n) {
        const size_t lo = staged_lo[b], m = staged_n[b];
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->stage_up[b], 0));
        const bool more = lo + m < n;
        size_t room = out_cap - out_pos;
        // a chunk of m records delivers at most max_entries + m flows, and the device loop wants room for one more whole eviction
        // (max_entries) whenever an epoch ends: with less it would report "no room" although the caller's buffer has some
        if (room > m + 2 * (size_t)h->cfg.max_entries) room = m + 2 * (size_t)h->cfg.max_entries;
        void** dbuf = ebuf ? &h->d_ep_out : &h->d_evict;
        size_t capb = ebuf ? h->d_ep_out_cap : (size_t)h->d_evict_cap;
        if (down_pending[ebuf]) { HIP_TRY(h, hipEventSynchronize(h->bounce_ev[ebuf])); down_pending[ebuf] = false; }   // its last download has left
        rc = ensure_bytes(h, dbuf, &capb, room * kRecordBytes + 16);
        if (ebuf) h->d_ep_out_cap = capb; else h->d_evict_cap = capb;
        if (rc != NFAGG_OK) break;
        size_t c = 0, e = 0;
        int rc_helper = NFAGG_OK, rc_helper2 = NFAGG_OK;
        std::thread helper, helper2;                            // one brings chunk k-1's evictions down (pageable output), one sends chunk k+1 up
        const bool down = !pinned_out && prev.has && prev.got != 0;
        // (a thread that cannot be created must not throw through the C boundary: its job is done here and now instead)
        auto spawn = [](std::thread& t, auto fn) { try { t = std::thread(fn); } catch (...) { fn(); } };
        if (down) spawn(helper, [&] { (void)hipSetDevice(h->device); rc_helper = bring_down(); });
        else prev.has = false;
        if (more) spawn(helper2, [&] { (void)hipSetDevice(h->device); rc_helper2 = stage(b ^ 1, lo + m); });
        NF_DIAG_T(t_a);
        rc = account_device_core(h, h->d_stage[b], m, *dbuf, room, epoch_end + n_ep, max_epochs - n_ep, &e, &c);
        NF_DIAG_T(t_b);
        if (helper.joinable()) helper.join();
        NF_DIAG_T(t_c);
        if (helper2.joinable()) helper2.join();
        NF_DIAG_T(t_d);
        NF_DIAG_PRINT("[account] chunk at %zu: %zu records, core %.2f ms (%zu evictions), wait down %.2f, wait up %.2f\n", lo, m, t_b - t_a, e, t_c - t_b, t_d - t_c);
        if (rc_helper == NFAGG_OK) rc_helper = rc_helper2;
        if (rc == NFAGG_OK && rc_helper != NFAGG_OK) rc = rc_helper;
        HIP_TRY(h, hipEventRecord(h->stage_free[b], h->stream));
        const size_t got = e ? (size_t)epoch_end[n_ep + e - 1] : 0;
        if (pinned_out) {
            if (got && rc >= 0) {
                // account_device_core has synchronised the fold stream for everything it wrote into *dbuf
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                HIP_TRY(h, hipMemcpyAsync((char*)out + out_pos * kRecordBytes, *dbuf, got * kRecordBytes, hipMemcpyDeviceToHost, h->d2h_stream));
                HIP_TRY(h, hipEventRecord(h->bounce_ev[ebuf], h->d2h_stream));
                down_pending[ebuf] = true;
            }
        } else {
            prev.has = true; prev.d = *dbuf; prev.got = got; prev.at = out_pos;
        }
        for (size_t k = 0; k < e; k++) epoch_end[n_ep + k] += out_pos;
        n_ep += e; out_pos += got; consumed += c;
        h->stage_next = b ^ 1;
        ebuf ^= 1;
        if (rc != NFAGG_OK || c < m) break;                     // stopped inside the chunk: what was staged beyond it is staged again by the next call
        b ^= 1;
    }
    {
        int rc_down = bring_down();
        if (pinned_out && (down_pending[0] || down_pending[1]) && hipStreamSynchronize(h->d2h_stream) != hipSuccess)
            rc_down = fail(h, NFAGG_EDEVICE, "eviction download failed");
        if (rc == NFAGG_OK || rc == NFAGG_TRUNCATED) { if (rc_down != NFAGG_OK) rc = rc_down; }
    }
    if (pinned_src) (void)hipStreamSynchronize(h->copy_stream);       // a chunk staged ahead may still be on its way: the caller's buffer is its own again
    *n_epochs_out = n_ep; *consumed_out = consumed;
    return rc;
}

// Page-locked host memory for record / eviction buffers: what nfagg_ingest, nfagg_account and nfagg_evict are handed in such a
// buffer crosses PCIe by DMA straight from / into it (no copy through the library's staging ring).
int nfagg_host_alloc(size_t bytes, void** p) {
    if (!p || bytes == 0) return NFAGG_EINVAL;
    *p = nullptr;
    if (hipHostMalloc(p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return NFAGG_ENOMEM; }
    return NFAGG_OK;
}
void nfagg_host_free(void* p) { if (p) (void)hipHostFree(p); }

int nfagg_host_threads(unsigned threads, int numa_node) {
    if (threads > 64) return NFAGG_EINVAL;
    return (int)HostPool::get().configure(threads, numa_node);
}

int nfagg_host_info(nfagg_host_pool_info* out) {
    if (!out || out->struct_size != sizeof(nfagg_host_pool_info)) return NFAGG_EINVAL;
    HostPool& p = HostPool::get();
    out->workers = p.workers(); out->parts = p.best_parts(); out->numa_node = p.node(); out->bound = p.bound() ? 1u : 0u;
    out->calibrated_gbs = p.calibrated_gbs();
    return NFAGG_OK;
}

int nfagg_device_numa_node(int device) { return device_numa_node(device); }

// pkg/model/record.go:90-97
void nfagg_record_times(int64_t now_unix_ns, uint64_t mono_now_ns, const nfagg_flow_metrics* m,
                        int64_t* time_flow_start_unix_ns, int64_t* time_flow_end_unix_ns) {
    const int64_t start_delta = (int64_t)(mono_now_ns - m->start_mono_time_ts);
    const int64_t end_delta = (int64_t)(mono_now_ns - m->end_mono_time_ts);
    if (time_flow_start_unix_ns) *time_flow_start_unix_ns = (int64_t)((uint64_t)now_unix_ns - (uint64_t)start_delta);
    if (time_flow_end_unix_ns) *time_flow_end_unix_ns = (int64_t)((uint64_t)now_unix_ns - (uint64_t)end_delta);
}

// ---------------------------------------------------------------- rollups
static int rollup_core(nfagg_handle* h, int kind, const void* partials, size_t n_flows, size_t n_cpu,
                       nfagg_flow_metrics* base, void* folded) {
    if (!h || !partials || !base || !folded) return fail(h, NFAGG_EINVAL, "null argument");
    if (n_flows == 0) return NFAGG_OK;
    if (n_cpu == 0) return fail(h, NFAGG_EINVAL, "n_cpu must be >= 1");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t ssz = rollup_struct_size(kind);
    const size_t pb = n_flows * n_cpu * ssz, bb = n_flows * sizeof(nfagg_flow_metrics), fb = n_flows * ssz;
    int rc;
    if ((rc = ensure_bytes(h, &h->d_roll[0], &h->d_roll_cap[0], pb)) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_roll[1], &h->d_roll_cap[1], bb)) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_roll[2], &h->d_roll_cap[2], fb)) != NFAGG_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_roll[0], partials, pb, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_roll[1], base, bb, hipMemcpyHostToDevice, h->stream));
    hipError_t e = launch_rollup(kind, h->d_roll[0], n_flows, n_cpu, h->d_roll[1], h->d_roll[2], h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "rollup launch failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipMemcpyAsync(base, h->d_roll[1], bb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(folded, h->d_roll[2], fb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NFAGG_OK;
}

int nfagg_rollup_additional(nfagg_handle* h, const nfagg_additional_metrics* p, size_t nf, size_t nc,
                            nfagg_flow_metrics* base, nfagg_additional_metrics* folded) { return rollup_core(h, 0, p, nf, nc, base, folded); }
int nfagg_rollup_dns(nfagg_handle* h, const nfagg_dns_metrics* p, size_t nf, size_t nc,
                     nfagg_flow_metrics* base, nfagg_dns_metrics* folded) { return rollup_core(h, 1, p, nf, nc, base, folded); }
int nfagg_rollup_drops(nfagg_handle* h, const nfagg_pkt_drop_metrics* p, size_t nf, size_t nc,
                       nfagg_flow_metrics* base, nfagg_pkt_drop_metrics* folded) { return rollup_core(h, 2, p, nf, nc, base, folded); }
int nfagg_rollup_network_events(nfagg_handle* h, const nfagg_network_events_metrics* p, size_t nf, size_t nc,
                                nfagg_flow_metrics* base, nfagg_network_events_metrics* folded) { return rollup_core(h, 3, p, nf, nc, base, folded); }
int nfagg_rollup_xlat(nfagg_handle* h, const nfagg_xlat_metrics* p, size_t nf, size_t nc,
                      nfagg_flow_metrics* base, nfagg_xlat_metrics* folded) { return rollup_core(h, 4, p, nf, nc, base, folded); }
int nfagg_rollup_quic(nfagg_handle* h, const nfagg_quic_metrics* p, size_t nf, size_t nc,
                      nfagg_flow_metrics* base, nfagg_quic_metrics* folded) { return rollup_core(h, 5, p, nf, nc, base, folded); }

// ---------------------------------------------------------------- map merge (LookupAndDeleteMap's join)
static const int kWalk[7] = {-1, NFAGG_ROLLUP_DNS, NFAGG_ROLLUP_DROPS, NFAGG_ROLLUP_NETWORK_EVENTS, NFAGG_ROLLUP_XLAT,
                             NFAGG_ROLLUP_ADDITIONAL, NFAGG_ROLLUP_QUIC};   // tracer.go:1057-1110; position 0 = main map

static int map_merge_device_core(nfagg_handle* h, const nfagg_map_view* mm, const nfagg_map_view fm[6], size_t n_cpu,
                                 const nfagg_merged_flows* out, size_t cap, size_t* n_out, size_t* n_dup) {
    if (!h || !mm || !fm || !out || !n_out) return fail(h, NFAGG_EINVAL, "null argument");
    if (n_cpu == 0 || n_cpu > 0xFFFFu) return fail(h, NFAGG_EINVAL, "n_cpu must be in [1, 65535]");
    MergeIn in{};
    in.n_cpu = (uint32_t)n_cpu;
    uint64_t total = 0;
    uintptr_t align = 0;
    for (int q = 0; q < 7; q++) {
        const nfagg_map_view& v = q == 0 ? *mm : fm[kWalk[q]];
        if (v.n && (!v.ids || !v.values)) return fail(h, NFAGG_EINVAL, "map %d: null ids/values", q);
        in.ids[q] = (const uint8_t*)v.ids; in.vals[q] = (const uint8_t*)v.values;
        in.off[q] = (uint32_t)total;
        total += v.n;
        if (v.n) align |= (uintptr_t)v.ids | (uintptr_t)v.values;
    }
    if (total > (1ull << 30)) return fail(h, NFAGG_ERANGE, "map merge: more than 2^30 rows");
    in.off[7] = (uint32_t)total;
    *n_out = 0;
    if (n_dup) *n_dup = 0;
    if (total == 0) return NFAGG_OK;
    if (cap && (!out->records || !out->present)) return fail(h, NFAGG_EINVAL, "null records/present output");
    align |= (uintptr_t)out->records | (uintptr_t)out->additional | (uintptr_t)out->dns | (uintptr_t)out->drops |
             (uintptr_t)out->network_events | (uintptr_t)out->xlat | (uintptr_t)out->quic;
    if (align & 7u) return fail(h, NFAGG_EINVAL, "map merge: device arrays must be 8-byte aligned");
    HIP_TRY(h, hipSetDevice(h->device));
    uint32_t n_slots = 1024;
    while ((uint64_t)n_slots < 2 * total) n_slots <<= 1;
    const size_t blocks = (total + 1023) / 1024;
    int rc;
    if ((rc = ensure_bytes(h, &h->d_mm[0], &h->d_mm_cap[0], (size_t)n_slots * merge_slot_bytes())) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_mm[1], &h->d_mm_cap[1], total * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_mm[2], &h->d_mm_cap[2], total * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_mm[3], &h->d_mm_cap[3], blocks * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_mm[4], &h->d_mm_cap[4], (blocks + 1) * sizeof(uint64_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_mm[5], &h->d_mm_cap[5], 16)) != NFAGG_OK) return rc;
    HIP_TRY(h, hipMemsetAsync(h->d_mm[0], 0xFF, (size_t)n_slots * merge_slot_bytes(), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_mm[5], 0, 16, h->stream));
    hipError_t e = launch_merge_build(in, h->d_mm[0], n_slots, (uint32_t*)h->d_mm[1], (unsigned int*)h->d_mm[5],
                                      (uint32_t*)h->d_mm[2], (uint32_t*)h->d_mm[3], h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "map merge build launch failed: %s", hipGetErrorString(e));
    e = launch_scan_block_sums((const uint32_t*)h->d_mm[3], (uint32_t)blocks, (uint64_t*)h->d_mm[4], h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "map merge scan launch failed: %s", hipGetErrorString(e));
    uint64_t flows = 0; unsigned int dups = 0;
    HIP_TRY(h, hipMemcpyAsync(&flows, (uint64_t*)h->d_mm[4] + blocks, sizeof flows, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&dups, h->d_mm[5], sizeof dups, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *n_out = (size_t)flows;
    if (n_dup) *n_dup = dups;
    if (flows > cap) return NFAGG_TRUNCATED;
    MergeOut o{out->records, out->present, out->additional, out->dns, out->drops, out->network_events, out->xlat, out->quic};
    e = launch_merge_fold(in, o, h->d_mm[0], (const uint32_t*)h->d_mm[1], (const uint32_t*)h->d_mm[2], (const uint64_t*)h->d_mm[4], h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "map merge fold launch failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NFAGG_OK;
}

int nfagg_map_merge_device(nfagg_handle* h, const nfagg_map_view* d_main_map, const nfagg_map_view d_feature_maps[6],
                           size_t n_cpu, const nfagg_merged_flows* d_out, size_t cap, size_t* n_out, size_t* n_duplicate_keys) {
    return map_merge_device_core(h, d_main_map, d_feature_maps, n_cpu, d_out, cap, n_out, n_duplicate_keys);
}

int nfagg_map_merge(nfagg_handle* h, const nfagg_map_view* main_map, const nfagg_map_view feature_maps[6],
                    size_t n_cpu, const nfagg_merged_flows* out, size_t cap, size_t* n_out, size_t* n_duplicate_keys) {
    if (!h || !main_map || !feature_maps || !out || !n_out) return fail(h, NFAGG_EINVAL, "null argument");
    if (n_cpu == 0) return fail(h, NFAGG_EINVAL, "n_cpu must be >= 1");
    HIP_TRY(h, hipSetDevice(h->device));
    nfagg_map_view dm{}, df[6] = {};
    int rc;
    for (int q = 0; q < 7; q++) {
        const nfagg_map_view& v = q == 0 ? *main_map : feature_maps[q - 1];
        nfagg_map_view& d = q == 0 ? dm : df[q - 1];
        d.n = v.n;
        if (!v.n) continue;
        if (!v.ids || !v.values) return fail(h, NFAGG_EINVAL, "map %d: null ids/values", q);
        const size_t vb = q == 0 ? v.n * sizeof(nfagg_flow_metrics) : v.n * n_cpu * rollup_struct_size(q - 1);
        if ((rc = ensure_bytes(h, &h->d_mm[6 + q], &h->d_mm_cap[6 + q], v.n * sizeof(nfagg_flow_id))) != NFAGG_OK) return rc;
        if ((rc = ensure_bytes(h, &h->d_mm[13 + q], &h->d_mm_cap[13 + q], vb)) != NFAGG_OK) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->d_mm[6 + q], v.ids, v.n * sizeof(nfagg_flow_id), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_mm[13 + q], v.values, vb, hipMemcpyHostToDevice, h->stream));
        d.ids = (const nfagg_flow_id*)h->d_mm[6 + q]; d.values = h->d_mm[13 + q];
    }
    void* host_out[8] = {out->records, out->present, out->additional, out->dns, out->drops, out->network_events, out->xlat, out->quic};
    const size_t elem[8] = {sizeof(nfagg_flow_record), 1, sizeof(nfagg_additional_metrics), sizeof(nfagg_dns_metrics), sizeof(nfagg_pkt_drop_metrics),
                            sizeof(nfagg_network_events_metrics), sizeof(nfagg_xlat_metrics), sizeof(nfagg_quic_metrics)};
    void* dev_out[8] = {};
    for (int k = 0; k < 8; k++) {
        if (!host_out[k] || !cap) continue;
        if ((rc = ensure_bytes(h, &h->d_mm[20 + k], &h->d_mm_cap[20 + k], cap * elem[k] + 16)) != NFAGG_OK) return rc;
        dev_out[k] = h->d_mm[20 + k];
    }
    nfagg_merged_flows dout{(nfagg_flow_record*)dev_out[0], (uint8_t*)dev_out[1], (nfagg_additional_metrics*)dev_out[2], (nfagg_dns_metrics*)dev_out[3],
                            (nfagg_pkt_drop_metrics*)dev_out[4], (nfagg_network_events_metrics*)dev_out[5], (nfagg_xlat_metrics*)dev_out[6],
                            (nfagg_quic_metrics*)dev_out[7]};
    rc = map_merge_device_core(h, &dm, df, n_cpu, &dout, cap, n_out, n_duplicate_keys);
    if (rc != NFAGG_OK) return rc;
    for (int k = 0; k < 8; k++)
        if (dev_out[k] && *n_out) HIP_TRY(h, hipMemcpyAsync(host_out[k], dev_out[k], *n_out * elem[k], hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NFAGG_OK;
}

// ---------------------------------------------------------------- sketches
static int sketch_info(nfagg_handle* h, int which, void** p, size_t* bytes) {
    if (which == NFAGG_CM_SRC || which == NFAGG_CM_DST) {
        if (!(h->sk.flags & NFAGG_SKETCH_CM)) return fail(h, NFAGG_ESTATE, "Count-Min sketch not enabled");
        *p = h->sk.cm[which - NFAGG_CM_SRC];
        *bytes = ((size_t)h->sk.cm_depth << h->sk.cm_log2w) * sizeof(uint64_t);
        return NFAGG_OK;
    }
    if (which == NFAGG_HLL_SRC || which == NFAGG_HLL_DST) {
        if (!(h->sk.flags & NFAGG_SKETCH_HLL)) return fail(h, NFAGG_ESTATE, "HyperLogLog sketch not enabled");
        *p = h->sk.hll[which - NFAGG_HLL_SRC];
        *bytes = ((size_t)1 << h->sk.hll_p);                     // uint8_t registers
        return NFAGG_OK;
    }
    return fail(h, NFAGG_EINVAL, "unknown sketch id %d", which);
}

int nfagg_sketch_device_ptr(nfagg_handle* h, int which, void** d_ptr, size_t* bytes) {
    if (!h || !d_ptr || !bytes) return fail(h, NFAGG_EINVAL, "null argument");
    return sketch_info(h, which, d_ptr, bytes);
}

int nfagg_sketch_snapshot(nfagg_handle* h, int which, void* out, size_t out_bytes) {
    if (!h || !out) return fail(h, NFAGG_EINVAL, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    void* p; size_t bytes;
    int rc = sketch_info(h, which, &p, &bytes);
    if (rc != NFAGG_OK) return rc;
    if (which == NFAGG_CM_SRC || which == NFAGG_CM_DST) {
        if (out_bytes < bytes) return NFAGG_TRUNCATED;
        HIP_TRY(h, hipMemcpyAsync(out, p, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return NFAGG_OK;
    }
    if (out_bytes < bytes) return NFAGG_TRUNCATED;                   // the device registers ARE the snapshot layout: one byte each
    HIP_TRY(h, hipMemcpyAsync(out, p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NFAGG_OK;
}

int nfagg_sketch_reset(nfagg_handle* h) {
    if (!h) return NFAGG_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    for (int which = 0; which < 4; which++) {
        const bool on = which < 2 ? (h->sk.flags & NFAGG_SKETCH_CM) : (h->sk.flags & NFAGG_SKETCH_HLL);
        if (!on) continue;
        void* p; size_t bytes;
        int rc = sketch_info(h, which, &p, &bytes);
        if (rc != NFAGG_OK) return rc;
        HIP_TRY(h, hipMemsetAsync(p, 0, bytes, h->stream));
    }
    return NFAGG_OK;
}

// HyperLogLog estimate (Flajolet et al. 2007, 64-bit hash so no large-range
// correction) from the histogram of register values: sum_k hist[k] * 2^-k in
// ascending k. Our own spec; the scalar oracle loops over the registers instead.
double nfagg_hll_estimate_from_histogram(const uint32_t* hist, uint32_t p) {
    const double m = (double)(1ull << p);
    const double alpha = (p == 4) ? 0.673 : (p == 5) ? 0.697 : (p == 6) ? 0.709 : 0.7213 / (1.0 + 1.079 / m);
    double sum = 0.0;
    for (int k = 0; k <= 64; k++) sum += (double)hist[k] * __builtin_ldexp(1.0, -k);
    double e = alpha * m * m / sum;
    if (e <= 2.5 * m && hist[0] != 0) e = m * __builtin_log(m / (double)hist[0]);
    return e;
}

int nfagg_hll_estimate(nfagg_handle* h, int which, double* estimate) {
    if (!h || !estimate) return fail(h, NFAGG_EINVAL, "null argument");
    if (which != NFAGG_HLL_SRC && which != NFAGG_HLL_DST) return fail(h, NFAGG_EINVAL, "which must be an HLL sketch");
    HIP_TRY(h, hipSetDevice(h->device));
    void* p; size_t bytes;
    int rc = sketch_info(h, which, &p, &bytes);
    if (rc != NFAGG_OK) return rc;
    hipError_t e = launch_hll_histogram((const uint8_t*)p, h->sk.hll_p, h->d_hist, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "hll histogram launch failed: %s", hipGetErrorString(e));
    uint32_t hist[65];
    HIP_TRY(h, hipMemcpyAsync(hist, h->d_hist, sizeof hist, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *estimate = nfagg_hll_estimate_from_histogram(hist, h->sk.hll_p);
    return NFAGG_OK;
}

int nfagg_cm_query(nfagg_handle* h, int which, const uint8_t ip[16], uint64_t* estimate) {
    if (!h || !ip || !estimate) return fail(h, NFAGG_EINVAL, "null argument");
    if (which != NFAGG_CM_SRC && which != NFAGG_CM_DST) return fail(h, NFAGG_EINVAL, "which must be a CM sketch");
    HIP_TRY(h, hipSetDevice(h->device));
    void* p; size_t bytes;
    int rc = sketch_info(h, which, &p, &bytes);
    if (rc != NFAGG_OK) return rc;
    uint64_t lo, hi;
    memcpy(&lo, ip, 8); memcpy(&hi, ip + 8, 8);
    const uint64_t ha = ip_hash(lo, hi, 0), hb = ip_hash(lo, hi, 1) | 1ull;
    uint64_t best = ~0ull;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (uint32_t r = 0; r < h->sk.cm_depth; r++) {
        uint64_t v;
        const uint64_t at = ((uint64_t)r << h->sk.cm_log2w) + cm_index(ha, hb, r, h->sk.cm_log2w);
        HIP_TRY(h, hipMemcpy(&v, (const uint64_t*)p + at, sizeof v, hipMemcpyDeviceToHost));
        if (v < best) best = v;
    }
    *estimate = best;
    return NFAGG_OK;
}

// Heavy hitters. Device: estimate per record, radix sort by estimate (descending). Host: walk the sorted order, keep the
// first occurrence of every address, stop once k distinct addresses are known AND the estimate has dropped below the
// k-th one (ties at the boundary are resolved by address bytes, so every candidate with the boundary estimate must be seen).
static int cm_topk_core(nfagg_handle* h, int which, const void* d_records, size_t n, size_t k, nfagg_heavy_hitter* out, size_t* n_out) {
    if (!h || !n_out || (k && !out) || (n && !d_records)) return fail(h, NFAGG_EINVAL, "null argument");
    if (which != NFAGG_CM_SRC && which != NFAGG_CM_DST) return fail(h, NFAGG_EINVAL, "which must be a CM sketch");
    if (n >= (1ull << 31)) return fail(h, NFAGG_ERANGE, "heavy hitters: more than 2^31 candidate records");
    *n_out = 0;
    void* cm; size_t cm_bytes;
    int rc = sketch_info(h, which, &cm, &cm_bytes);
    if (rc != NFAGG_OK) return rc;
    if (n == 0 || k == 0) return NFAGG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const int side = which - NFAGG_CM_SRC;
    size_t temp_bytes = 0;
    hipError_t e = launch_cm_sort_desc(nullptr, nullptr, nullptr, nullptr, n, nullptr, &temp_bytes, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "sort size query failed: %s", hipGetErrorString(e));
    if ((rc = ensure_bytes(h, &h->d_hh[0], &h->d_hh_cap[0], n * sizeof(uint64_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_hh[1], &h->d_hh_cap[1], n * sizeof(uint64_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_hh[2], &h->d_hh_cap[2], n * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_hh[3], &h->d_hh_cap[3], n * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_hh[4], &h->d_hh_cap[4], temp_bytes + 16)) != NFAGG_OK) return rc;
    e = launch_cm_estimate((const uint64_t*)cm, h->sk.cm_depth, h->sk.cm_log2w, side, d_records, n, (uint64_t*)h->d_hh[0], (uint32_t*)h->d_hh[2], h->stream);
    if (e == hipSuccess) e = launch_cm_sort_desc((const uint64_t*)h->d_hh[0], (uint64_t*)h->d_hh[1], (const uint32_t*)h->d_hh[2], (uint32_t*)h->d_hh[3], n,
                                                 h->d_hh[4], &temp_bytes, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "heavy-hitter launch failed: %s", hipGetErrorString(e));
    struct Row { uint64_t lo, hi, est; };
    std::vector<Row> rows, best;                         // best: distinct addresses in order of appearance (estimate descending)
    std::set<std::pair<uint64_t, uint64_t>> group;       // addresses already taken at the current estimate
    uint64_t group_est = ~0ull;
    size_t seen = 0, m = k * 16 < 4096 ? 4096 : k * 16;
    for (;;) {
        if (m > n) m = n;
        if ((rc = ensure_bytes(h, &h->d_hh[5], &h->d_hh_cap[5], m * sizeof(Row))) != NFAGG_OK) return rc;
        e = launch_cm_gather(d_records, side, (const uint64_t*)h->d_hh[1], (const uint32_t*)h->d_hh[3], m, (uint64_t*)h->d_hh[5], h->stream);
        if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "heavy-hitter gather failed: %s", hipGetErrorString(e));
        rows.resize(m);
        HIP_TRY(h, hipMemcpyAsync(rows.data(), h->d_hh[5], m * sizeof(Row), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        bool done = false;
        for (; seen < m; seen++) {
            const Row& r = rows[seen];
            if (best.size() >= k && r.est < best[k - 1].est) { done = true; break; }   // below the boundary: nothing further can enter
            // an address always carries the same estimate, so a duplicate can only sit among the entries with THIS estimate
            if (r.est != group_est) { group.clear(); group_est = r.est; }
            if (group.insert(std::make_pair(r.lo, r.hi)).second) best.push_back(r);
        }
        if (done || m == n) break;
        m *= 4;
    }
    std::sort(best.begin(), best.end(), [](const Row& a, const Row& b) {
        if (a.est != b.est) return a.est > b.est;
        return memcmp(&a.lo, &b.lo, 16) < 0;             // lo,hi are adjacent: the 16 address bytes in order
    });
    const size_t cnt = best.size() < k ? best.size() : k;
    for (size_t q = 0; q < cnt; q++) { memcpy(out[q].ip, &best[q].lo, 16); out[q].estimate = best[q].est; }
    *n_out = cnt;
    return NFAGG_OK;
}

int nfagg_cm_topk_device(nfagg_handle* h, int which, const void* d_records, size_t n, size_t k, nfagg_heavy_hitter* out, size_t* n_out) {
    return cm_topk_core(h, which, d_records, n, k, out, n_out);
}

int nfagg_cm_topk(nfagg_handle* h, int which, const void* records, size_t n, size_t k, nfagg_heavy_hitter* out, size_t* n_out) {
    if (!h || (n && !records)) return fail(h, NFAGG_EINVAL, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_bytes(h, &h->d_hh[6], &h->d_hh_cap[6], n * kRecordBytes + 16);
    if (rc != NFAGG_OK) return rc;
    if (n) HIP_TRY(h, hipMemcpyAsync(h->d_hh[6], records, n * kRecordBytes, hipMemcpyHostToDevice, h->stream));
    return cm_topk_core(h, which, h->d_hh[6], n, k, out, n_out);
}

// ---------------------------------------------------------------- misc
uint64_t nfagg_key_hash(const nfagg_flow_id* id) {
    uint64_t w[5];
    memcpy(w, id, 40);
    w[4] &= 0x00ffffffffffffffull;   // byte 39
    return key_hash(w);
}

void nfagg_shard_ids(const void* records, size_t n, uint32_t n_shards, uint32_t* out_shard) {
    const char* p = static_cast<const char*>(records);
    for (size_t i = 0; i < n; i++)
        out_shard[i] = nfagg_shard_of(reinterpret_cast<const nfagg_flow_id*>(p + i * kRecordBytes), n_shards);
}

uint32_t nfagg_shard_of(const nfagg_flow_id* id, uint32_t n_shards) { return shard_of_hash(nfagg_key_hash(id), n_shards); }

uint64_t nfagg_ip_hash(const uint8_t ip[16], uint32_t seed_index) {
    uint64_t lo, hi;
    memcpy(&lo, ip, 8); memcpy(&hi, ip + 8, 8);
    return ip_hash(lo, hi, seed_index);
}

int nfagg_sync(nfagg_handle* h) {
    if (!h) return NFAGG_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NFAGG_OK;
}

void* nfagg_stream(nfagg_handle* h) { return h ? (void*)h->stream : nullptr; }

int nfagg_debug_skip_sequence(nfagg_handle* h, uint64_t records) {
    if (!h) return NFAGG_EINVAL;
    h->epoch_seq += records;                                    // any amount: the window follows (nfagg_rebase.hip)
    return NFAGG_OK;
}

int nfagg_stats_get(nfagg_handle* h, nfagg_stats* out) {
    if (!h || !out) return fail(h, NFAGG_EINVAL, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = refresh_counters(h);
    if (rc != NFAGG_OK) return rc;
    prof_resolve(h);
    h->stats.entries = h->live;
    h->stats.epoch_seq = h->epoch_seq;
    *out = h->stats;
    return NFAGG_OK;
}

// ---- record -> protobuf (nfagg_pb.hip)
// feat (optional): DEVICE pointers
static int encode_pb_device_core(nfagg_handle* h, const void* d_records, size_t n, const nfagg_pb_features* feat, const nfagg_pb_options* opt,
                                 void* d_out, size_t out_cap, uint64_t* d_frame_offsets, uint32_t* d_body_len,
                                 void* d_kafka_keys, size_t* out_bytes) {
    if (!h || !opt || !out_bytes || !d_frame_offsets || (n && (!d_records || !d_body_len))) return fail(h, NFAGG_EINVAL, "null argument");
    if (opt->struct_size != sizeof(nfagg_pb_options)) return fail(h, NFAGG_EINVAL, "nfagg_pb_options.struct_size mismatch");
    if (opt->unknown_len > 16 || (opt->n_names && !opt->names)) return fail(h, NFAGG_EINVAL, "bad namer table");
    if ((((uintptr_t)d_records | (uintptr_t)d_out | (uintptr_t)d_kafka_keys) & 15u) != 0) return fail(h, NFAGG_EINVAL, "device buffers must be 16-byte aligned");
    for (uint32_t k = 0; k < opt->n_names; k++)
        if (opt->names[k].name_len > 16 || opt->names[k].udn_len > 63) return fail(h, NFAGG_EINVAL, "namer row %u: name/udn too long", k);
    PbFeat F{};
    if (feat) {
        if (feat->struct_size != sizeof(nfagg_pb_features)) return fail(h, NFAGG_EINVAL, "nfagg_pb_features.struct_size mismatch");
        if ((((uintptr_t)feat->additional | (uintptr_t)feat->dns | (uintptr_t)feat->drops | (uintptr_t)feat->xlat | (uintptr_t)feat->quic) & 7u) != 0)
            return fail(h, NFAGG_EINVAL, "feature arrays must be 8-byte aligned");
        F.present = feat->present;
        F.additional = (const uint8_t*)feat->additional; F.dns = (const uint8_t*)feat->dns; F.drops = (const uint8_t*)feat->drops;
        F.xlat = (const uint8_t*)feat->xlat; F.quic = (const uint8_t*)feat->quic;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    *out_bytes = 0;
    if (n == 0) { HIP_TRY(h, hipMemsetAsync(d_frame_offsets, 0, sizeof(uint64_t), h->stream)); HIP_TRY(h, hipStreamSynchronize(h->stream)); return NFAGG_OK; }
    const size_t blocks = (n + 1023) / 1024;
    int rc;
    if ((rc = ensure_bytes(h, &h->d_pb[0], &h->d_pb_cap[0], n * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_pb[1], &h->d_pb_cap[1], blocks * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_pb[2], &h->d_pb_cap[2], (blocks + 1) * sizeof(uint64_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_pb[3], &h->d_pb_cap[3], (size_t)(opt->n_names + 1) * sizeof(nfagg_intf_name))) != NFAGG_OK) return rc;
    if (opt->n_names) {   // the kernels binary-search the table: stable sort by if_index keeps the scan-in-table-order answer
        h->pb_names.assign(opt->names, opt->names + opt->n_names);
        std::stable_sort(h->pb_names.begin(), h->pb_names.end(), [](const nfagg_intf_name& a, const nfagg_intf_name& b) { return a.if_index < b.if_index; });
        HIP_TRY(h, hipMemcpyAsync(h->d_pb[3], h->pb_names.data(), opt->n_names * sizeof(nfagg_intf_name), hipMemcpyHostToDevice, h->stream));
    }
    PbParams P{};
    P.now_sec = opt->now_unix_ns / 1000000000ll; P.now_nsec = opt->now_unix_ns % 1000000000ll;
    if (P.now_nsec < 0) { P.now_nsec += 1000000000ll; P.now_sec -= 1; }
    P.mono_now = opt->mono_now_ns;
    memcpy(P.agent_ip_w, opt->agent_ip, 16);
    static const uint8_t v4pre[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xff, 0xff};
    P.agent_is_v4 = memcmp(opt->agent_ip, v4pre, 12) == 0;     // net.IP.To4() != nil (proto.go:255-261)
    P.names = (const nfagg_intf_name*)h->d_pb[3]; P.n_names = opt->n_names;
    P.unknown_len = opt->unknown_len; memcpy(P.unknown, opt->unknown_name, 16);
    hipError_t e = launch_pb_size(d_records, n, P, F, d_body_len, (uint32_t*)h->d_pb[0], (uint32_t*)h->d_pb[1], (uint64_t*)h->d_pb[2], h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "protobuf size launch failed: %s", hipGetErrorString(e));
    uint64_t total = 0;
    HIP_TRY(h, hipMemcpyAsync(&total, (uint64_t*)h->d_pb[2] + blocks, sizeof total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *out_bytes = (size_t)total;
    if (total > out_cap || !d_out) return NFAGG_TRUNCATED;
    e = launch_pb_write(d_records, n, P, F, d_body_len, (const uint32_t*)h->d_pb[0], (const uint64_t*)h->d_pb[2], d_out, d_frame_offsets, d_kafka_keys, total, h->stream);
    if (e != hipSuccess) return fail(h, NFAGG_EDEVICE, "protobuf encode launch failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NFAGG_OK;
}

// feat (optional): HOST pointers
static int encode_pb_host_core(nfagg_handle* h, const void* records, size_t n, const nfagg_pb_features* feat, const nfagg_pb_options* opt,
                               void* out, size_t out_cap, uint64_t* frame_offsets, uint32_t* body_len,
                               void* kafka_keys, size_t* out_bytes) {
    if (!h || !opt || !out_bytes || !frame_offsets || (n && (!records || !body_len))) return fail(h, NFAGG_EINVAL, "null argument");
    if (feat && feat->struct_size != sizeof(nfagg_pb_features)) return fail(h, NFAGG_EINVAL, "nfagg_pb_features.struct_size mismatch");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_bytes(h, &h->d_pb[4], &h->d_pb_cap[4], n * kRecordBytes + 16)) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_pb[5], &h->d_pb_cap[5], out_cap + 32)) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_pb[6], &h->d_pb_cap[6], (n + 1) * sizeof(uint64_t))) != NFAGG_OK) return rc;
    if ((rc = ensure_bytes(h, &h->d_pb[7], &h->d_pb_cap[7], (n + 1) * sizeof(uint32_t))) != NFAGG_OK) return rc;
    if (kafka_keys && (rc = ensure_bytes(h, &h->d_pb[8], &h->d_pb_cap[8], n * 32 + 32)) != NFAGG_OK) return rc;
    if (n) HIP_TRY(h, hipMemcpyAsync(h->d_pb[4], records, n * kRecordBytes, hipMemcpyHostToDevice, h->stream));
    nfagg_pb_features dfeat{};
    if (feat && n) {
        dfeat.struct_size = sizeof dfeat;
        const void* src[6] = {feat->present, feat->additional, feat->dns, feat->drops, feat->xlat, feat->quic};
        const size_t elem[6] = {1, sizeof(nfagg_additional_metrics), sizeof(nfagg_dns_metrics), sizeof(nfagg_pkt_drop_metrics),
                                sizeof(nfagg_xlat_metrics), sizeof(nfagg_quic_metrics)};
        void* dst[6] = {};
        for (int k = 0; k < 6; k++) {
            if (!src[k]) continue;
            if ((rc = ensure_bytes(h, &h->d_pb[9 + k], &h->d_pb_cap[9 + k], n * elem[k] + 16)) != NFAGG_OK) return rc;
            HIP_TRY(h, hipMemcpyAsync(h->d_pb[9 + k], src[k], n * elem[k], hipMemcpyHostToDevice, h->stream));
            dst[k] = h->d_pb[9 + k];
        }
        dfeat.present = (const uint8_t*)dst[0]; dfeat.additional = (const nfagg_additional_metrics*)dst[1];
        dfeat.dns = (const nfagg_dns_metrics*)dst[2]; dfeat.drops = (const nfagg_pkt_drop_metrics*)dst[3];
        dfeat.xlat = (const nfagg_xlat_metrics*)dst[4]; dfeat.quic = (const nfagg_quic_metrics*)dst[5];
    }
    rc = encode_pb_device_core(h, h->d_pb[4], n, (feat && n) ? &dfeat : nullptr, opt, out ? h->d_pb[5] : nullptr, out_cap,
                               (uint64_t*)h->d_pb[6], (uint32_t*)h->d_pb[7], kafka_keys ? h->d_pb[8] : nullptr, out_bytes);
    if (rc != NFAGG_OK) return rc;
    if (*out_bytes) HIP_TRY(h, hipMemcpyAsync(out, h->d_pb[5], *out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(frame_offsets, h->d_pb[6], (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    if (n) HIP_TRY(h, hipMemcpyAsync(body_len, h->d_pb[7], n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    if (n && kafka_keys) HIP_TRY(h, hipMemcpyAsync(kafka_keys, h->d_pb[8], n * 32, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return NFAGG_OK;
}

int nfagg_encode_pb_device(nfagg_handle* h, const void* d_records, size_t n, const nfagg_pb_options* opt,
                           void* d_out, size_t out_cap, uint64_t* d_frame_offsets, uint32_t* d_body_len,
                           void* d_kafka_keys, size_t* out_bytes) {
    return encode_pb_device_core(h, d_records, n, nullptr, opt, d_out, out_cap, d_frame_offsets, d_body_len, d_kafka_keys, out_bytes);
}

int nfagg_encode_pb(nfagg_handle* h, const void* records, size_t n, const nfagg_pb_options* opt,
                    void* out, size_t out_cap, uint64_t* frame_offsets, uint32_t* body_len,
                    void* kafka_keys, size_t* out_bytes) {
    return encode_pb_host_core(h, records, n, nullptr, opt, out, out_cap, frame_offsets, body_len, kafka_keys, out_bytes);
}

int nfagg_encode_pb_content_device(nfagg_handle* h, const void* d_records, size_t n, const nfagg_pb_features* d_features,
                                   const nfagg_pb_options* opt, void* d_out, size_t out_cap, uint64_t* d_frame_offsets,
                                   uint32_t* d_body_len, void* d_kafka_keys, size_t* out_bytes) {
    if (!d_features) return fail(h, NFAGG_EINVAL, "null features (use nfagg_encode_pb_device)");
    return encode_pb_device_core(h, d_records, n, d_features, opt, d_out, out_cap, d_frame_offsets, d_body_len, d_kafka_keys, out_bytes);
}

int nfagg_encode_pb_content(nfagg_handle* h, const void* records, size_t n, const nfagg_pb_features* features,
                            const nfagg_pb_options* opt, void* out, size_t out_cap, uint64_t* frame_offsets,
                            uint32_t* body_len, void* kafka_keys, size_t* out_bytes) {
    if (!features) return fail(h, NFAGG_EINVAL, "null features (use nfagg_encode_pb)");
    return encode_pb_host_core(h, records, n, features, opt, out, out_cap, frame_offsets, body_len, kafka_keys, out_bytes);
}

#ifdef NFAGG_DIAG
// libnfagg_diag.so only: what the last epochs-found-first launch of nfagg_account saw — its control words and its cuts (host copies)
int nfagg_debug_last_cuts(nfagg_handle* h, uint32_t ctl[8], uint32_t* cuts, size_t cap) {
    if (!h || !ctl || !h->h_par) return NFAGG_EINVAL;
    for (int k = 0; k < 8; k++) ctl[k] = h->h_par[k];
    for (size_t k = 0; k < cap && k < 65535; k++) cuts[k] = h->h_par[512 + k];
    return NFAGG_OK;
}
int nfagg_debug_last_analysis(nfagg_handle* h, uint64_t* keys_sorted, int32_t* prev, uint32_t* pos, size_t n) {
    if (!h || !h->d_par[1]) return NFAGG_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    if (keys_sorted) HIP_TRY(h, hipMemcpy(keys_sorted, h->d_par[1], n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (prev) HIP_TRY(h, hipMemcpy(prev, h->d_par[2], n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (pos) HIP_TRY(h, hipMemcpy(pos, h->d_par[3], n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return NFAGG_OK;
}
// libnfagg_diag.so only: the dense-identity timing experiment of k_finalize + k_evict (nfagg_kernels.hip g_diag_dense). on = 1: a
// buffer of one 64-byte unit per slot is allocated and every table of the PROCESS uses it (one table at a time); 0: back to the
// cold half lines. Evicted MACs are wrong while it is on.
int nfagg_debug_dense_identity(nfagg_handle* h, int on) {
    static void* buf = nullptr;
    if (!h) return NFAGG_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    if (buf) { (void)hipFree(buf); buf = nullptr; }
    if (on) HIP_TRY(h, hipMalloc(&buf, (size_t)h->slots * 64));
    HIP_TRY(h, diag_set_dense(buf));
    return NFAGG_OK;
}
// libnfagg_diag.so only (not part of the drop-in ABI): per-phase wave-cycle sums of the phase-timing builds (variants 6/8/9).
int nfagg_debug_phase_cycles(nfagg_handle* h, uint64_t out[8]) {
    if (!h || !out) return NFAGG_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = refresh_counters(h);
    if (rc != NFAGG_OK) return rc;
    for (int k = 0; k < 8; k++) out[k] = h->h_ctr->phase[k];
    return NFAGG_OK;
}
#endif

int nfagg_stats_reset_profile(nfagg_handle* h) {
    if (!h) return NFAGG_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    prof_resolve(h);
    h->stats.ingest_launches = h->stats.evict_launches = h->stats.sketch_launches = 0;
    h->stats.ingest_kernel_ms = h->stats.evict_kernel_ms = h->stats.sketch_kernel_ms = 0.0;
    return NFAGG_OK;
}

}  // extern "C"

#include "nfagg_group.inc"
