// nfagg_dedup_cached.hip — kernel-dedup mode, ONE streaming pass over the batch (default for NFAGG_MODE_KERNEL_DEDUP; the direct
// kernels of nfagg_dedup.hip serve small batches).
//
// The merge of bpf/flows.c:98-143 needs the flow's first interface F before a record can be counted, and F is the interface of
// the record with the smallest sequence number anywhere in the batch. Rounds 1-2 therefore streamed the batch twice (a claim pass
// resolving F for every flow, then a fold pass): 2 x 14.4 GB per 100 M records. Here nothing that depends on F happens while the
// batch is streamed:
//   k_dedup_stream  streams the batch once. A workgroup folds the SUB-FLOWS (flow key, if_index_first_seen) it has LDS cache
//                   entries for — an entry is a DedupPartial (nfagg_dedup.h: sums, ORs, "last value" tags, the two earliest
//                   directions) plus the smallest sequence number of its records, everything kept for both roles (counted / side
//                   interface). A record whose sub-flow gets no entry is spilled by index to the queue of its FLOW's partition
//                   (nfagg_spill.h). At the end the entries themselves are exported (144 bytes each, the size of a record) and
//                   queued the same way. The table is not touched.
//   k_dedup_parts   one workgroup per partition, owner of the partition's flows: folds the queued records and exported entries
//                   in its own sub-flow cache, then flushes in two phases with a barrier between them — phase A claims the slot and
//                   resolves first record and earliest interfaces (dedup_claim) for EVERY sub-flow of the partition, phase B
//                   merges the partials (dedup_merge), F now being final. What finds no cache entry is claimed at once and
//                   retried in further rounds (in place, as in nfagg_ingest_part.hip).
//   k_dedup_overflow  the (normally empty) overflow list: claims before k_dedup_parts, folds after it.
// Merging DedupPartials is associative and commutative (sums, ORs, maxima over sequence-tagged words, top-2 over sequence-tagged
// words), so the result is that of the direct passes: bit-exact vs the oracle's sequential fold.
#include <atomic>
#include "nfagg_dedup.h"
#include "nfagg_spill.h"

namespace nfagg {
namespace dcache {

constexpr int kBlock = 1024;
constexpr int kProbe = 8;
constexpr uint32_t kPad = 0xffffffffu;
constexpr uint32_t kXpFlag = 0x80000000u;       // queue item: index of an exported entry, not of a record (batches hold < 2^31 records)
// Batches of up to 2^28 - 1 records (always, in practice): three more bits of the flow's key hash ride above the index, its
// SUB-PARTITION — the partition pass sorts what its cache could not take by them (as nfagg_ingest_part.hip does).
constexpr int kSubBits = 3, kSubs = 1 << kSubBits;
constexpr uint32_t kIdxBits = 31 - kSubBits, kIdxMask = (1u << kIdxBits) - 1u, kIdxMaskUntagged = ~kXpFlag;
NF_DEV uint32_t sub_tag(uint64_t h, const SpillView& q, bool tag_on) {
    const uint32_t sh = q.part_shift >= (uint32_t)kSubBits ? q.part_shift - kSubBits : 0;
    return tag_on ? ((uint32_t)(h >> sh) & (uint32_t)(kSubs - 1)) << kIdxBits : 0u;
}

NF_DEV uint4 u4(uint64_t a, uint64_t b) { return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)); }

NF_DEV uint64_t subflow_hash(uint64_t h, uint32_t ifx) {
    uint64_t z = (h ^ ((uint64_t)ifx * 0xD6E8FEB86659FD93ull)) * kMul;
    return (z ^ (z >> 32)) | 1ull;
}

template <int K>
struct FoldCache {
    uint64_t h64[K];
    uint64_t key[5][K];
    uint64_t bytes[K];
    uint64_t endl_lo[K], endl_hi[K];
    uint64_t dscp_tag[K], samp_tag[K];
    uint64_t ssl_first[K];
    uint64_t cs_tag[K], ks_tag[K];
    uint64_t dir[2][K];
    uint32_t ifx[K];
    uint32_t packets[K];
    uint32_t flags[K];
    uint32_t ssl_max[K], ssl_minv[K];
    uint32_t min_seq[K];
};

template <int K>
NF_DEV void cache_init(FoldCache<K>& L, int tid) {
    for (int e = tid; e < K; e += kBlock) {
        L.h64[e] = 0; L.bytes[e] = 0; L.endl_lo[e] = 0; L.endl_hi[e] = 0; L.dscp_tag[e] = 0; L.samp_tag[e] = 0;
        L.ssl_first[e] = 0; L.cs_tag[e] = 0; L.ks_tag[e] = 0; L.dir[0][e] = 0; L.dir[1][e] = 0;
        L.packets[e] = 0; L.flags[e] = 0; L.ssl_max[e] = 0; L.ssl_minv[e] = 0; L.min_seq[e] = 0xffffffffu;
    }
}

// find or claim the entry of sub-flow hash hs; the creator writes key and interface. -1 = window full, or (DOORBITS > 0) the
// sub-flow is seen for the first time: entries are never evicted, so a sub-flow is admitted on its second appearance (admission
// filter `door`, as in nfagg_ingest_part.hip) — one-off sub-flows of the cold tail do not take the entries of the hot ones.
// Exactly one of the lanes that meet a new sub-flow in the same tile is turned away (the atomic's return value decides).
template <int K, int DOORBITS, typename Cache>
NF_DEV int claim(Cache& L, uint32_t* door, uint64_t hs, const uint64_t w[5], uint32_t ifx, uint32_t* fill = nullptr) {
    uint32_t e = (uint32_t)(hs >> 40) & (K - 1);
#pragma unroll 1
    for (int p = 0; p < kProbe; p++) {
        uint64_t cur = L.h64[e];
        if (cur == 0) {
            if (DOORBITS > 0) {
                const uint32_t b = (uint32_t)(hs >> 14) & (uint32_t)(DOORBITS - 1), m = 1u << (b & 31);
                if (!(door[b >> 5] & m) && !(atomicOr(&door[b >> 5], m) & m)) return -1;
            }
            cur = atomicCAS((unsigned long long*)&L.h64[e], 0ull, (unsigned long long)hs);
            if (cur == 0) {
#pragma unroll
                for (int k = 0; k < 5; k++) L.key[k][e] = w[k];
                L.ifx[e] = ifx;
                if (fill) atomicAdd(fill, 1u);                   // entries in use (the partition pass sizes its rounds by it)
                return (int)e;
            }
        }
        if (cur == hs) return (int)e;
        e = (e + 1) & (K - 1);
    }
    return -1;
}

template <typename Cache>
NF_DEV bool same_subflow(const Cache& L, int ent, const uint64_t w[5], uint32_t ifx) {
    bool same = L.ifx[ent] == ifx;
#pragma unroll
    for (int k = 0; k < 5; k++) same &= (L.key[k][ent] == w[k]);
    return same;
}

// LDS flavour of topk_insert<2, 0xff> (nfagg_dedup.h): two earliest distinct directions
template <typename Cache>
NF_DEV void lds_dir_insert(Cache& L, int ent, uint64_t v) {
    for (int trip = 0; trip < 64; trip++) {
        const uint64_t c0 = L.dir[0][ent], c1 = L.dir[1][ent];
        const bool m0 = c0 != 0 && ((c0 ^ v) & 0xffull) == 0, m1 = c1 != 0 && ((c1 ^ v) & 0xffull) == 0;
        int pos; uint64_t cur;
        if (m0) { pos = 0; cur = c0; } else if (m1) { pos = 1; cur = c1; } else if (c1 < c0) { pos = 1; cur = c1; } else { pos = 0; cur = c0; }
        if (cur >= v) return;
        if (atomicCAS((unsigned long long*)&L.dir[pos][ent], (unsigned long long)cur, (unsigned long long)v) == cur) return;
    }
}

// a partial (one record, or an exported entry) into the entry of its sub-flow; ms = smallest sequence number it stands for
template <int K>
NF_DEV void fold_into(FoldCache<K>& L, int e, const DedupPartial& p, uint32_t ms) {
    if (p.bytes) atomicAdd((unsigned long long*)&L.bytes[e], (unsigned long long)p.bytes);
    if (p.packets) atomicAdd(&L.packets[e], p.packets);
    if (p.flags & ~L.flags[e]) atomicOr(&L.flags[e], p.flags);
    if (p.endl_lo > L.endl_lo[e]) atomicMax((unsigned long long*)&L.endl_lo[e], (unsigned long long)p.endl_lo);
    if (p.endl_hi > L.endl_hi[e]) atomicMax((unsigned long long*)&L.endl_hi[e], (unsigned long long)p.endl_hi);
    if (p.dscp_tag > L.dscp_tag[e]) atomicMax((unsigned long long*)&L.dscp_tag[e], (unsigned long long)p.dscp_tag);
    if (p.samp_tag > L.samp_tag[e]) atomicMax((unsigned long long*)&L.samp_tag[e], (unsigned long long)p.samp_tag);
    if (p.ssl_first) {
        atomicMax((unsigned long long*)&L.ssl_first[e], (unsigned long long)p.ssl_first);
        atomicMax(&L.ssl_max[e], p.ssl_max);
        atomicMax(&L.ssl_minv[e], p.ssl_minv);
    }
    if (p.cs_tag) atomicMax((unsigned long long*)&L.cs_tag[e], (unsigned long long)p.cs_tag);
    if (p.ks_tag) atomicMax((unsigned long long*)&L.ks_tag[e], (unsigned long long)p.ks_tag);
    if (p.dir0) lds_dir_insert(L, e, p.dir0);
    if (p.dir1) lds_dir_insert(L, e, p.dir1);
    if (L.min_seq[e] > ms) atomicMin(&L.min_seq[e], ms);
}

template <int K>
NF_DEV void partial_of_entry(const FoldCache<K>& L, int e, DedupPartial& p) {
    p.bytes = L.bytes[e]; p.packets = L.packets[e]; p.flags = L.flags[e];
    p.endl_lo = L.endl_lo[e]; p.endl_hi = L.endl_hi[e]; p.dscp_tag = L.dscp_tag[e]; p.samp_tag = L.samp_tag[e];
    p.ssl_first = L.ssl_first[e]; p.ssl_max = L.ssl_max[e]; p.ssl_minv = L.ssl_minv[e];
    p.cs_tag = L.cs_tag[e]; p.ks_tag = L.ks_tag[e];
    p.dir0 = L.dir[0][e]; p.dir1 = L.dir[1][e];
    p.ifx = L.ifx[e];
}

// The streaming pass's entry: 108 bytes instead of 152, so that 1024 of them fit beside the spill staging (a 512-entry cache
// holds the two interfaces of 256 flows: too few for a Zipf stream). What it leaves out:
//   * the TLS words (ssl_first / ssl_max / ssl_minv / cs_tag / ks_tag): a record that carries TLS information (handshake
//     packets: a few per connection) is spilled to the partition pass, whose entries have them;
//   * the sequence tags of the "last value" fields (end, dscp, sampling: assigned by the LAST record, flows.c:108,110-111,128).
//     A workgroup meets its records in sequence order, tile after tile, so the last record of an entry so far is in the current
//     tile: the lanes of a tile agree on it with ONE atomic max (last_seq), and after the tile's second barrier the lane that
//     holds it stores the three values plainly. (Round 2's entry took four 64-bit tagged atomic maxima per record for these,
//     every one of them a winner — later records carry larger tags — and all lanes of the hot flow on the same address.)
struct StreamCache {
    static constexpr int K = 1024;
    uint64_t h64[K];
    uint64_t key[5][K];
    uint64_t bytes[K];
    uint64_t end[K];              // of the record last_seq names
    uint64_t dir[2][K];
    uint32_t ifx[K];
    uint32_t packets[K];
    uint32_t flags[K];
    uint32_t min_seq[K];
    uint32_t last_seq[K];         // largest sequence number folded, + 1 (0 = none)
    uint32_t samp[K], dscp[K];    // of the record last_seq names
};
static_assert(sizeof(StreamCache) == 108 * 1024, "stream cache entry");

NF_DEV void stream_init(StreamCache& L, int tid) {
    for (int e = tid; e < StreamCache::K; e += kBlock) {
        L.h64[e] = 0; L.bytes[e] = 0; L.dir[0][e] = 0; L.dir[1][e] = 0;
        L.packets[e] = 0; L.flags[e] = 0; L.min_seq[e] = 0xffffffffu; L.last_seq[e] = 0;
    }
}

// does the record carry TLS information (ssl_version, tls_cipher_suite, tls_key_share, tls_types: record dwords 33, 34)?
NF_DEV bool has_tls(const Rec& r) { return (r.d[33] | (r.d[34] & 0x00ffffffu)) != 0; }

// phase B of the streaming pass: what is a sum, an OR or a minimum — and the vote for "last record"
NF_DEV void stream_fold(StreamCache& L, int e, const Rec& r, uint32_t seq32) {
    if (r.bytes()) atomicAdd((unsigned long long*)&L.bytes[e], (unsigned long long)r.bytes());
    if (r.packets()) atomicAdd(&L.packets[e], r.packets());
    if (r.flags() & ~L.flags[e]) atomicOr(&L.flags[e], r.flags());
    if (L.last_seq[e] < seq32 + 1u) atomicMax(&L.last_seq[e], seq32 + 1u);
    lds_dir_insert(L, e, ((uint64_t)(~seq32) << 8) | (r.d[24] & 0xffu));
    if (L.min_seq[e] > seq32) atomicMin(&L.min_seq[e], seq32);
}

NF_DEV void stream_export(const StreamCache& L, int e, uint4* dst) {
    const uint64_t s1 = L.last_seq[e], end = L.end[e];
    dst[0] = u4(L.key[0][e], L.key[1][e]);
    dst[1] = u4(L.key[2][e], L.key[3][e]);
    dst[2] = u4(L.key[4][e], (uint64_t)L.ifx[e] | ((uint64_t)L.min_seq[e] << 32));
    dst[3] = u4(L.bytes[e], (uint64_t)L.packets[e] | ((uint64_t)L.flags[e] << 32));
    dst[4] = u4((s1 << 32) | (uint32_t)end, (s1 << 32) | (uint32_t)(end >> 32));
    dst[5] = u4((s1 << 8) | L.dscp[e], (s1 << 32) | L.samp[e]);
    dst[6] = u4(0, 0);
    dst[7] = u4(0, 0);
    dst[8] = u4(L.dir[0][e], L.dir[1][e]);
}

// ---- an exported entry: 36 dwords, the size of a record (one load path for both kinds of queue item)
//   0..9 key   10 if_index   11 min_seq   12,13 bytes   14 packets   15 flags | tls_types << 16   16..19 endl_lo, endl_hi
//   20..23 dscp_tag, samp_tag   24,25 ssl_first   26 ssl_max   27 ssl_minv   28..31 cs_tag, ks_tag   32..35 dir0, dir1
// One queue item in registers: its flow key, interface, smallest sequence number and partial. `raw` holds the 144 bytes the
// item names: a record of the batch (canonicalised here) or an exported entry.
struct Item {
    uint64_t w[5];
    uint64_t h;                 // flow key hash
    uint32_t ifx, ms;
    DedupPartial p;
};

NF_DEV void decode_item(uint32_t it, uint32_t idx_mask, Rec& raw, uint32_t seq_base32, Item& x) {
    if (it & kXpFlag) {
#pragma unroll
        for (int k = 0; k < 5; k++) x.w[k] = raw.q(k);
        x.ifx = raw.d[10]; x.ms = raw.d[11];
        DedupPartial& p = x.p;
        p.bytes = raw.q(6); p.packets = raw.d[14]; p.flags = raw.d[15];
        p.endl_lo = raw.q(8); p.endl_hi = raw.q(9); p.dscp_tag = raw.q(10); p.samp_tag = raw.q(11);
        p.ssl_first = raw.q(12); p.ssl_max = raw.d[26]; p.ssl_minv = raw.d[27];
        p.cs_tag = raw.q(14); p.ks_tag = raw.q(15); p.dir0 = raw.q(16); p.dir1 = raw.q(17);
        p.ifx = x.ifx;
    } else {
        raw.canonicalize();
        raw.key_words(x.w);
        x.ifx = raw.d[21];
        x.ms = seq_base32 + (it & idx_mask);
        dedup_partial_from_record(raw, x.ms, x.p);
    }
    x.h = key_hash(x.w);
}

NF_DEV void load_item(const SpillView& q, const void* recs, uint32_t it, uint32_t idx_mask, Rec& raw) {
    // padding loads record 0 (the batch is not empty): the pipeline requests unconditionally
    const void* base = (it != kPad && (it & kXpFlag)) ? (const void*)q.xp : recs;
    const uint64_t i = it == kPad ? 0 : (uint64_t)(it & idx_mask);
    load_record(base, i, raw);
}

// ---- straight on the table, for one item (cache misses, overflow list)
NF_DEV void claim_item(const TableView& t, const Item& x) {
    Hints hx;
    const uint64_t kx = sub_kx(t, x.ifx), hs = sub_hash(t, x.h, x.ifx);      // sub-flow tables (nfagg_dedup.h): keyed by (flow, interface)
    uint32_t idx = probe_home(t, x.w, hs, hx, kx);
    if (idx == kNoSlot) {
        idx = find_or_claim(t, x.w, hs, nullptr, nullptr, nullptr, kx);
        if (idx == kNoSlot) return;
        hx.id0 = 0;
    }
    dedup_claim(t, idx, hx.id0, x.ifx, x.ms);
}

// every claim of the flow is done (by this workgroup, or an earlier kernel): the slot's id0 is final — read it past the caches
NF_DEV void merge_at(const TableView& t, uint32_t idx, const DedupPartial& p, uint32_t ms, const void* recs, uint32_t seq_base32) {
    Hints hx;
    hx.flags = 0;
    hx.id0 = ald(&t.hot[idx].id0);
    if ((uint32_t)(hx.id0 >> 32) == ~ms) {
        // this sub-flow's earliest record is the flow's first record: fetch it again and store it whole
        Rec r;
        load_record(recs, (uint64_t)(ms - seq_base32), r);
        r.canonicalize();
        dedup_publish_first(t, idx, r, ms);
    }
    dedup_merge<true>(t, idx, hx, p);
}

NF_DEV void fold_item(const TableView& t, const SketchView& sk, const Item& x, const void* recs, uint32_t seq_base32) {
    if (sk.flags) sketch_add(sk, x.w, x.p.bytes);                // the item's records reach the sketches here, once (see parts_flush)
    Hints hx;
    const uint64_t kx = sub_kx(t, x.ifx), hs = sub_hash(t, x.h, x.ifx);
    uint32_t idx = probe_home(t, x.w, hs, hx, kx);
    if (idx == kNoSlot) {
        idx = find_or_claim(t, x.w, hs, nullptr, nullptr, nullptr, kx);          // claimed before: this only walks the probe sequence
        if (idx == kNoSlot) return;
    }
    merge_at(t, idx, x.p, x.ms, recs, seq_base32);
}

constexpr int kStreamEntries = StreamCache::K;  // the streaming pass shares the LDS with the spill staging (40 KiB) and the filter
constexpr int kPartEntries = 1024;
constexpr int kDoorBits = 32768;                // 4 KiB
constexpr int kStreamGrid = 256;
static_assert((uint64_t)kStreamGrid * kStreamEntries == kDedupXpEntries, "exported-entry area (nfagg_internal.h)");
static_assert(kStreamEntries <= kBlock && kPartEntries == kBlock, "one cache entry per lane in the export / flush");

// ---- the streaming pass ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_dedup_stream(TableView t, SpillView q, const void* __restrict__ recs, uint64_t n, uint64_t seq_base) {
    constexpr int K = kStreamEntries;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    StreamCache& L = *reinterpret_cast<StreamCache*>(lds_raw);
    spill::Stage& S = *reinterpret_cast<spill::Stage*>(lds_raw + sizeof(StreamCache));
    uint32_t* door = reinterpret_cast<uint32_t*>(lds_raw + sizeof(StreamCache) + sizeof(spill::Stage));
    const int tid = threadIdx.x;
    const uint32_t seq_base32 = (uint32_t)seq_base;
    stream_init(L, tid);
    for (int e = tid; e < kDoorBits / 32; e += kBlock) door[e] = 0;
    spill::Lane<kBlock> sp;
    sp.init(S, tid);
    __syncthreads();
    unsigned long long skipped = 0, spilled = 0;
    const uint64_t n_tiles = (n + kBlock - 1) / kBlock;
    const bool tag_on = n <= (uint64_t)kIdxMask;
    // software pipeline: the next tile's record is requested (unconditionally, on a clamped index) before this one is processed
    bool valid; uint64_t i; Rec r;
    {
        const uint64_t pos = (uint64_t)blockIdx.x * kBlock + tid;
        valid = pos < n; i = valid ? pos : 0;
        load_record(recs, i, r);
    }
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        bool valid_n; uint64_t i_n; Rec r_n;
        {
            const uint64_t pos = (tile + gridDim.x) * kBlock + tid;
            valid_n = pos < n; i_n = valid_n ? pos : 0;
            load_record(recs, i_n, r_n);
        }
        uint64_t w[5], h = 0;
        if (valid && !record_keys(t, r, w, h)) { valid = false; skipped++; }
        const uint32_t seq32 = seq_base32 + (uint32_t)i;
        const uint32_t ifx = valid ? r.d[21] : 0;
        int ent = -1;
        if (valid && !has_tls(r)) ent = claim<K, kDoorBits>(L, door, subflow_hash(h, ifx), w, ifx);
        __syncthreads();
        bool hit = false;
        if (valid && ent >= 0 && same_subflow(L, ent, w, ifx)) { hit = true; stream_fold(L, ent, r, seq32); }
        const bool miss = valid && !hit;
        sp.drain(S, q, tid);
        __syncthreads();
        // the last record of the entry so far is in this tile: its lane stores the "last value" fields
        if (hit && L.last_seq[ent] == seq32 + 1u) { L.end[ent] = r.end(); L.samp[ent] = r.sampling(); L.dscp[ent] = r.dscp(); }
        if (miss) spilled++;
        sp.append(S, q, miss, spill::part_of(h, q.part_shift), (uint32_t)i | sub_tag(h, q, tag_on));
        // the next tile's claims only write h64/key/ifx of NEW entries; everything else is ordered by its barrier
        r = r_n; valid = valid_n; i = i_n;
    }
    // ---- export: one more "tile" of the spill protocol, whose items are this workgroup's entries
    __syncthreads();
    bool used = false;
    uint32_t item = 0, part = 0;
    if (tid < K && L.h64[tid] != 0 && L.min_seq[tid] != 0xffffffffu) {
        used = true;
        const uint32_t at = (uint32_t)blockIdx.x * (uint32_t)K + (uint32_t)tid;
        stream_export(L, tid, q.xp + (uint64_t)at * 9);
        uint64_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = L.key[k][tid];
        const uint64_t h = key_hash(w);
        part = spill::part_of(h, q.part_shift);
        item = kXpFlag | at | sub_tag(h, q, tag_on);
    }
    sp.drain(S, q, tid);
    __syncthreads();
    sp.append(S, q, used, part, item);
    // one more (empty) tile: an entry that found its staging group full — the groups hold what the last tiles left in them — is
    // carried, and a carried item only gets in after a drain; finish() alone would send it to the overflow list, whose items
    // are claimed and merged one by one (~115 per workgroup: 30 k per call, 0.17 ms in the two overflow kernels)
    __syncthreads();
    sp.drain(S, q, tid);
    __syncthreads();
    sp.append(S, q, false, 0, 0);
    sp.finish(S, q, tid);
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
    if (spilled) aadd(&t.ctr->n_bypassed, spilled);
}

// ---- the partition pass ------------------------------------------------------------------------------------------------
struct PartsLds {
    uint32_t retry_cnt;             // items written back for a further round
    uint32_t sub_cnt[kSubs];        // ... of the first round, per sub-partition
    uint32_t sub_off[kSubs + 1];    // counting sort: start of every sub-partition's segment
    uint32_t sub_fill[kSubs];
    uint32_t fill;                  // cache entries in use since the last cache_init (sorted rounds)
    uint32_t new_cnt[4];            // grouped flush: [0] fresh slots claimed by it, [1..2] base of their live-list range, [3] deferred claims allowed
    uint32_t grp[kPartEntries];     // grouped flush: flow hash -> leader entry + 1 (0 = empty)
};
constexpr uint32_t kPackItems = 768;   // consecutive sub-partitions share a round while their items (>= sub-flows) surely fit a cache

// Fold the `count` items at `queue` into the cache. An item whose sub-flow gets no entry goes back to the front of the same
// region (the write position never passes the read position: items are read two tiles ahead) and is retried in the next round
// with a fresh cache; FIRST: its claim is made at once, so that after this round's flush every sub-flow of the partition has
// been claimed. COHERENT: the items were written by this workgroup (read past L1).
// ABL (libnfagg_diag.so only, ingest_variant 13..15: timing experiments, results are WRONG): bit 0 = no flush, bit 1 = no fold
// into the entry, bit 2 = no cache claim either (every item only gathered and decoded).
// retry_to: where misses go (the list itself for the in-place rounds; the front of the queue region when the items were sorted into
// its tail first: k_dedup_parts). COUNT_SUBS: keep the per-sub-partition counts of the misses (the unsorted first round).
template <bool FIRST, bool COHERENT, int ABL = 0, bool COUNT_SUBS = true>
NF_DEV void parts_round(const TableView& t, const SpillView& q, FoldCache<kPartEntries>& L, PartsLds& P, uint32_t* queue, uint32_t count,
                        const void* recs, uint32_t seq_base32, uint32_t idx_mask, uint32_t* retry_to = nullptr) {
    if (!retry_to) retry_to = queue;
    constexpr int K = kPartEntries;
    const int tid = threadIdx.x;
    auto qload = [&](uint32_t pos) -> uint32_t { return COHERENT ? ald(&queue[pos]) : queue[pos]; };
    const uint32_t n_tiles = (count + kBlock - 1) / kBlock;
    uint32_t it_cur = kPad, it_next = kPad;
    Rec raw;
    {
        const uint32_t pos = (uint32_t)tid;
        if (pos < count) it_cur = qload(pos);
        if (pos + kBlock < count) it_next = qload(pos + kBlock);
        load_item(q, recs, it_cur, idx_mask, raw);
    }
    for (uint32_t tile = 0; tile < n_tiles; tile++) {
        uint32_t it_nn = kPad;
        Rec raw_n;
        {
            const uint64_t p2 = (uint64_t)(tile + 2) * kBlock + tid;
            if (p2 < count) it_nn = qload((uint32_t)p2);
            load_item(q, recs, it_next, idx_mask, raw_n);
        }
        const bool valid = it_cur != kPad;
        Item x;
        x.h = 0; x.ifx = 0; x.ms = 0;
        int ent = -1;
        if (valid) {
            decode_item(it_cur, idx_mask, raw, seq_base32, x);
            if (!(ABL & 4)) ent = claim<K, 0>(L, nullptr, subflow_hash(x.h, x.ifx), x.w, x.ifx, &P.fill);
            if (ABL & 4) asm volatile("" :: "v"(x.h), "v"(x.p.bytes), "v"(x.p.dir0));
        }
        __syncthreads();
        if (valid && !(ABL & 4)) {
            if (ent >= 0 && same_subflow(L, ent, x.w, x.ifx)) {
                if (!(ABL & 2)) fold_into(L, ent, x.p, x.ms);
            } else {
                if (FIRST) {
                    claim_item(t, x);
                    if (COUNT_SUBS) atomicAdd(&P.sub_cnt[(it_cur & ~kXpFlag) >> kIdxBits], 1u);      // (untagged items: all in sub-partition 0..7 of the index bits; unused then)
                }
                retry_to[atomicAdd(&P.retry_cnt, 1u)] = it_cur;  // in place: lands below (tile + 1) * kBlock
            }
        }
        it_cur = it_next; it_next = it_nn;
        raw = raw_n;
    }
    __syncthreads();
}

// Phase A: slot + first record + earliest interfaces for every entry. Barrier. Phase B: the merges (F is final: every sub-flow
// of every flow of this partition has been claimed — by this flush, by parts_round<FIRST>, by the overflow kernel before this
// launch, or in an earlier call).
// The sketches (every record's bytes, whatever the dedup merge does with them) are fed HERE, one contribution per
// cache entry: an entry's `bytes` is the plain sum over all the records it stands for — the counted / side decision is taken at
// the merge (dedup_merge), not in the partial — and every record of the batch ends in exactly one flush of one entry (through the
// streaming pass's exported entry or as a queued record) or in fold_item. Count-Min is linear, HyperLogLog idempotent. Round 5 ran
// k_sketch_update over the batch as a second pass: 8.1 ms of the 14.3 ms configs[4] step (profiles/r05_bench_world1_nccl_configs4_100m.json).
NF_DEV void parts_flush(const TableView& t, const SketchView& sk, FoldCache<kPartEntries>& L, const void* recs, uint32_t seq_base32) {
    const int e = threadIdx.x;
    const bool used = L.h64[e] != 0 && L.min_seq[e] != 0xffffffffu;
    uint32_t idx = kNoSlot;
    if (used) {
        uint64_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = L.key[k][e];
        const uint64_t kx = sub_kx(t, L.ifx[e]), h = sub_hash(t, key_hash(w), L.ifx[e]);
        Hints hx;
        bool fresh = false;
        idx = probe_home(t, w, h, hx, kx);
        if (idx == kNoSlot) {
            // a lane that claims the slot plants its own first-record tag and candidate interface with the slot's first values
            Partial first{};
            first.first_inv = ~L.min_seq[e]; first.ident0 = L.ifx[e];
            idx = find_or_claim(t, w, h, &fresh, &hx.home_tag, &first, kx);
            hx.id0 = 0;
        }
        if (idx != kNoSlot && !fresh) dedup_claim(t, idx, hx.id0, L.ifx[e], L.min_seq[e]);
    }
    drain_stores();                                              // this lane's claims have reached the memory side
    __syncthreads();
    if (idx != kNoSlot) {
        DedupPartial p;
        partial_of_entry(L, e, p);
        merge_at(t, idx, p, L.min_seq[e], recs, seq_base32);
    }
    if (used && sk.flags) {
        uint64_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = L.key[k][e];
        sketch_add(sk, w, L.bytes[e]);
    }
    __syncthreads();
}

// ---- the FIRST flush of a partition, grouped by flow (round 4) -----------------------------------------------------------------
// The partition's workgroup owns its flows while it runs, and after the first round nearly every flow of a one-call epoch is NEW
// to the table. The flush above pays for such a flow, per sub-flow: a probe, (one of them) a claim that zeroes 512 bytes with 26
// write-through stores, then ~10 atomics on the words just zeroed. Here the sub-flow entries of a flow are first brought together
// in LDS (a hash of the flow key elects one entry the LEADER, the others hang on its list); the leader takes the slot
// (find_or_claim<DEFER>: tag = claimed, nothing written) and, when the slot is fresh, works out the flow's whole state from its
// entries — F = the interface of the entry with the smallest sequence number, the seven earliest interfaces, the counted entry's
// sums and "last value" tags, the side entries' directions, end and flags (exactly what dedup_claim + dedup_merge would leave) —
// and writes hot line, cold half line and aux lines ONCE with plain 16-byte stores (key and tag coherent, as
// merge_partial_exclusive does). The slots claimed this way enter the live list as one range per workgroup. A flow whose slot
// exists already (an earlier call, the overflow kernel, an item of the round that found no cache entry and was claimed at once),
// a flow with more than kGroupMax entries, and every flow on tables too small for deferred claims (TableView.defer_claims) take
// the two-phase path above, lane by lane. Exactness: the state is a pure function of the flow's sub-flow partials (the slot was
// empty), and it is the function the atomic path computes — the join of nfagg_dedup_join.hip is the same computation on slots.
constexpr int kGroupMax = 8;
struct FlushLink { uint32_t lead_nxt; uint32_t slot; };   // overlays FoldCache.h64[e] during the flush: leader entry | next entry << 16; the leader's slot | fresh << 31
static_assert(sizeof(FlushLink) == sizeof(uint64_t), "FlushLink overlays h64");

// The leader of a fresh flow: everything the slot will hold, from the entries on its list.
NF_DEV void flush_fresh_flow(const TableView& t, const FoldCache<kPartEntries>& L, const FlushLink* K, int lead, uint32_t idx, const uint64_t w[5],
                             const void* recs, uint32_t seq_base32) {
    // the entry with the flow's earliest record: its interface is F (flows.c: the first record is stored whole, its if_index is if_index_first_seen)
    uint32_t first_seq = 0xffffffffu; int fe = lead;
    for (uint32_t x = (uint32_t)lead; x != 0xffffu; x = K[x].lead_nxt >> 16) {
        const uint32_t sq = L.min_seq[x];
        if (sq < first_seq) { first_seq = sq; fe = (int)x; }
    }
    const uint32_t F = L.ifx[fe];
    // end: the LAST record that reaches either branch; flags: every record that does (side records: the 16 flag bits only)
    uint64_t endl_lo = 0, endl_hi = 0;
    uint32_t flags = 0;
    for (uint32_t x = (uint32_t)lead; x != 0xffffu; x = K[x].lead_nxt >> 16) {
        const uint32_t ifx = L.ifx[x];
        if (ifx != F && ifx == 0) continue;                                  // flows.c:126: ignored
        const uint64_t lo = L.endl_lo[x], hi = L.endl_hi[x];
        endl_lo = lo > endl_lo ? lo : endl_lo; endl_hi = hi > endl_hi ? hi : endl_hi;
        flags |= ifx == F ? L.flags[x] : (L.flags[x] & 0xffffu);
    }
    // the seven interfaces that appear earliest (non-zero ones; F among them when it is non-zero); side ones with their directions
    uint64_t cand[kCand], d0[kCand], d1[kCand];
    long long prev = -1;
#pragma unroll
    for (int k = 0; k < kCand; k++) {
        uint32_t best = 0xffffffffu; int bx = -1;
        for (uint32_t x = (uint32_t)lead; x != 0xffffu; x = K[x].lead_nxt >> 16) {
            const uint32_t sq = L.min_seq[x];
            if (L.ifx[x] != 0 && (long long)sq > prev && sq < best) { best = sq; bx = (int)x; }
        }
        cand[k] = 0; d0[k] = 0; d1[k] = 0;
        if (bx >= 0) {
            cand[k] = tagged(~best, L.ifx[bx]);
            if (L.ifx[bx] != F) { d0[k] = L.dir[0][bx]; d1[k] = L.dir[1][bx]; }
            prev = (long long)best;
        } else prev = 0x100000000ll;                                         // none left
    }
    // the first record, whole (account.go:95): start, eth_protocol, MACs here; its identity dwords too (k_finalize writes them again)
    Rec r;
    load_record(recs, (uint64_t)(first_seq - seq_base32), r);
    r.canonicalize();
    const uint32_t inv = ~first_seq;
    uint4* HL = reinterpret_cast<uint4*>(&t.hot[idx]);
    uint4* CL = reinterpret_cast<uint4*>(&t.cold[idx]);
    uint4* AL = reinterpret_cast<uint4*>(&t.aux[idx]);
    const uint64_t bytes = L.bytes[fe], pf = (uint64_t)L.packets[fe] | ((uint64_t)flags << 32), id0 = tagged(inv, F);
    const uint64_t smac_lo = tagged(inv, (uint32_t)r.smac()), dmac_lo = tagged(inv, (uint32_t)r.dmac());
    HL[3] = u4(bytes, 0);
    HL[4] = u4(r.start(), pf);
    HL[5] = u4((uint64_t)r.eth(), L.dscp_tag[fe]);
    HL[6] = u4(L.samp_tag[fe], id0);
    HL[7] = u4(smac_lo, dmac_lo);
    CL[0] = u4(tagged(inv, (uint32_t)(r.smac() >> 32)), tagged(inv, (uint32_t)(r.dmac() >> 32)));
    CL[1] = make_uint4(r.d[22], r.d[24], r.d[25], r.d[26]);                 // (pad2 cleared by canonicalize)
    CL[2] = make_uint4(r.d[27], r.d[28], r.d[29], r.d[30]);
    CL[3] = make_uint4(r.d[31], r.d[32], r.d[33], r.d[34]);
    // aux words: cand[0..6], endl_lo, endl_hi, ssl_first, ssl_max | ssl_minv << 32, cs_tag, ks_tag, dir[k][0..1] (13 + 2k), pad
    AL[0] = u4(cand[0], cand[1]); AL[1] = u4(cand[2], cand[3]); AL[2] = u4(cand[4], cand[5]);
    AL[3] = u4(cand[6], endl_lo);
    AL[4] = u4(endl_hi, L.ssl_first[fe]);
    AL[5] = u4((uint64_t)L.ssl_max[fe] | ((uint64_t)L.ssl_minv[fe] << 32), L.cs_tag[fe]);
    AL[6] = u4(L.ks_tag[fe], d0[0]);
    AL[7] = u4(d1[0], d0[1]); AL[8] = u4(d1[1], d0[2]); AL[9] = u4(d1[2], d0[3]); AL[10] = u4(d1[3], d0[4]);
    AL[11] = u4(d1[4], d0[5]); AL[12] = u4(d1[5], d0[6]); AL[13] = u4(d1[6], 0);
    AL[14] = u4(0, 0); AL[15] = u4(0, 0);
    // the key, coherently (other workgroups compare it while they probe past this slot); the tag follows after the drain
    SlotHot* H = &t.hot[idx];
#pragma unroll
    for (int k = 0; k < 5; k++) ast(&H->key[k], w[k]);
}

NF_DEV void parts_flush_grouped(const TableView& t, const SketchView& sk, FoldCache<kPartEntries>& L, PartsLds& P, const void* recs, uint32_t seq_base32) {
    static_assert(offsetof(SlotAux, endl_lo) == 56 && offsetof(SlotAux, ssl_first) == 72 && offsetof(SlotAux, cs_tag) == 88 &&
                  offsetof(SlotAux, dir) == 104, "flush_fresh_flow writes the aux lines as 16-byte units");
    const int e = threadIdx.x;
    const bool used = L.h64[e] != 0 && L.min_seq[e] != 0xffffffffu;
    const bool defer = P.new_cnt[3] != 0;                                  // written before the round (a barrier since)
    uint64_t w[5] = {0, 0, 0, 0, 0}, h = 0;
    if (used) {
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = L.key[k][e];
        h = key_hash(w);
    }
    __syncthreads();                                                         // every lane has looked at its h64: the links may overlay it
    FlushLink* K = reinterpret_cast<FlushLink*>(L.h64);
    K[e].lead_nxt = (uint32_t)e | 0xffff0000u;                               // its own leader, empty list
    K[e].slot = kNoSlot;
    P.grp[e] = 0;
    if (e == 0) P.new_cnt[0] = 0;
    __syncthreads();
    // ---- bring the sub-flows of a flow together: the entry that plants itself in the flow's grp slot leads
    int lead = e;
    if (used) {
        uint32_t g = (uint32_t)(h >> 20) & (uint32_t)(kPartEntries - 1);
        for (;;) {
            uint32_t cur = P.grp[g];
            if (cur == 0) { cur = atomicCAS(&P.grp[g], 0u, (uint32_t)e + 1u); if (cur == 0) break; }
            const int o = (int)cur - 1;
            bool same = true;
#pragma unroll
            for (int k = 0; k < 5; k++) same &= (L.key[k][o] == w[k]);
            if (same) { lead = o; break; }
            g = (g + 1) & (uint32_t)(kPartEntries - 1);
        }
        if (lead != e) {                                                     // hang on the leader's list (its word: leader | head << 16)
            uint32_t* lw = &K[lead].lead_nxt;
            uint32_t cur = *lw;
            for (;;) {
                K[e].lead_nxt = (uint32_t)lead | (cur & 0xffff0000u);        // my next = the old head
                const uint32_t old = atomicCAS(lw, cur, (cur & 0xffffu) | ((uint32_t)e << 16));
                if (old == cur) break;
                cur = old;
            }
        }
    }
    __syncthreads();
    // ---- leaders take the slot
    bool fresh = false;
    uint32_t my_k = 0;
    if (used && lead == e) {
        int len = 0;
        for (uint32_t x = (uint32_t)e; x != 0xffffu && len <= kGroupMax; x = K[x].lead_nxt >> 16) len++;
        Hints hx;
        uint32_t idx = probe_home(t, w, h, hx);
        if (idx == kNoSlot) {
            if (defer && len <= kGroupMax) idx = find_or_claim<true>(t, w, h, &fresh, &hx.home_tag);
            else idx = find_or_claim(t, w, h);
        }
        if (fresh) my_k = atomicAdd(&P.new_cnt[0], 1u);
        K[e].slot = idx == kNoSlot ? kNoSlot : (idx | (fresh ? 0x80000000u : 0u));
    }
    __syncthreads();
    // ---- fresh flows: the leader writes the whole slot. Every other entry: pass 1 of the two-phase path on the flow's slot
    const uint32_t sl = used ? K[lead].slot : kNoSlot;
    const uint32_t idx = sl == kNoSlot ? kNoSlot : (sl & 0x7fffffffu);
    const bool flow_fresh = sl != kNoSlot && (sl & 0x80000000u) != 0;
    if (fresh) flush_fresh_flow(t, L, K, e, idx, w, recs, seq_base32);
    else if (idx != kNoSlot && !flow_fresh) dedup_claim(t, idx, 0, L.ifx[e], L.min_seq[e]);
    drain_stores();                                                          // this lane's claims / slot stores have reached the memory side
    __syncthreads();
    const uint32_t cnt = P.new_cnt[0];
    if (e == 0 && cnt) {                                                     // the fresh slots: one range of the live list (pass2_flush's rule)
        const unsigned long long base = aadd(&t.ctr->n_live, (unsigned long long)cnt);
        P.new_cnt[1] = (uint32_t)base; P.new_cnt[2] = (uint32_t)(base >> 32);
        if (base + cnt > t.claim_limit) {
            const unsigned long long keep = base < t.claim_limit ? t.claim_limit - base : 0ull;
            aadd(&t.ctr->n_live, ~(unsigned long long)(cnt - keep) + 1ull);
            atomicExch(&t.ctr->aborted, 1u);
        }
    }
    if (fresh) ast(&t.hot[idx].tag, tag_ready(t, h));
    else if (idx != kNoSlot && !flow_fresh) {
        DedupPartial p;
        partial_of_entry(L, e, p);
        merge_at(t, idx, p, L.min_seq[e], recs, seq_base32);
    }
    __syncthreads();
    if (fresh) {
        const unsigned long long base = (unsigned long long)P.new_cnt[1] | ((unsigned long long)P.new_cnt[2] << 32);
        if (base + my_k < t.claim_limit) t.live_list[base + my_k] = idx;
        else ast(&t.hot[idx].tag, (uint64_t)0);
    }
    if (used && sk.flags) sketch_add(sk, w, L.bytes[e]);                     // one contribution per entry (parts_flush)
    __syncthreads();
}

constexpr int kMaxRounds = 16;

// Rounds over the items at list[0..m) (written by this workgroup) until none is left: every round takes what fits a fresh cache
// and writes the rest back to the front of the list. Every item's sub-flow has been claimed on the table (first round).
NF_DEV void parts_drain(const TableView& t, const SketchView& sk, const SpillView& q, FoldCache<kPartEntries>& L, PartsLds& P, uint32_t* list, uint32_t m,
                        const void* recs, uint32_t seq_base32, uint32_t idx_mask, int max_rounds) {
    const int tid = threadIdx.x;
    for (int round = 1; m != 0; round++) {
        if (round >= max_rounds) {
            // far more sub-flows than rounds x entries: what is left is merged item by item
            for (uint32_t k = tid; k < m; k += kBlock) {
                const uint32_t it = ald(&list[k]);
                Rec raw;
                load_item(q, recs, it, idx_mask, raw);
                Item x;
                decode_item(it, idx_mask, raw, seq_base32, x);
                fold_item(t, sk, x, recs, seq_base32);
            }
            break;
        }
        cache_init(L, tid);
        if (tid == 0) P.retry_cnt = 0;
        __syncthreads();
        parts_round<false, true>(t, q, L, P, list, m, recs, seq_base32, idx_mask);
        parts_flush(t, sk, L, recs, seq_base32);
        m = P.retry_cnt;
        drain_stores();
        __syncthreads();                                          // everybody has read retry_cnt; the retry list is written
    }
}


template <int ABL = 0>
__global__ __launch_bounds__(kBlock) void k_dedup_parts(TableView t, SketchView sk, SpillView q, const void* __restrict__ recs, uint64_t n, uint64_t seq_base,
                                                        int max_rounds) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FoldCache<kPartEntries>& L = *reinterpret_cast<FoldCache<kPartEntries>*>(lds_raw);
    PartsLds& P = *reinterpret_cast<PartsLds*>(lds_raw + sizeof(FoldCache<kPartEntries>));
    const int tid = threadIdx.x;
    const uint32_t seq_base32 = (uint32_t)seq_base;
    const uint32_t tail = q.qtail[blockIdx.x];                    // written by the streaming pass (previous kernel)
    const uint32_t count = tail < q.qcap ? tail : q.qcap;
    if (count == 0) return;                                       // uniform for the workgroup
    uint32_t* my_queue = q.queue + (uint64_t)blockIdx.x * q.qcap;
    const bool tag_on = n <= (uint64_t)kIdxMask;                  // the streaming pass put the sub-partition bits above the index
    const uint32_t idx_mask = tag_on ? kIdxMask : kIdxMaskUntagged;
    cache_init(L, tid);
    if (tid == 0) {
        P.retry_cnt = 0;
        // deferred claims of the grouped flush (as pass 2 of the accounter fold: a look at n_live a round early)
        P.new_cnt[3] = (t.defer_claims && ald(&t.ctr->n_live) + 512ull * kPartEntries <= t.claim_limit) ? 1u : 0u;
    }
    if (tid < kSubs) { P.sub_cnt[tid] = 0; P.sub_fill[tid] = 0; }
    if (tid == 0) P.fill = 0;
    __syncthreads();
    if (tid == 0) q.qtail[blockIdx.x] = 0;                        // every lane has read it: ready for the next batch
    // ---- Round 4: SORT FIRST. The items carry three more bits of their flow's hash (the sub-partition): a counting sort of the
    // queue by those bits — 4-byte items, no record is touched — puts every flow's items into one of eight runs, and the runs are
    // folded one after the other, several sharing a cache while it has room (P.fill: entries in use; the density seen so far
    // decides how many runs the next segment takes) and a flush + a fresh cache when it has not. Every flush sees COMPLETE flows
    // (all sub-flows of a flow are in one run), so it can be the grouped one, and what used to overflow the one cache of the first
    // round — claimed item by item on the table, gathered again in retry rounds: ~80 % of the items at 10 M flows per GPU — mostly
    // fits now. It costs the sparse case ~9 % (more, shorter segments; two flushes where one cache nearly held the partition: dedup
    // Zipf over 1 M flows 7.1 -> 7.8 ms per 100 M records), so the API asks for it (SpillView.sort_first) when the last epoch
    // delivered more flows than the partitions' caches hold together — both ways are exact. Partitions too small to bother,
    // untagged items (batches of 2^28 records or more) and regions without room for the sorted copy take the unsorted first
    // round below.
    const uint32_t sorted_at0 = (count + 3u) & ~3u;
    if (ABL == 0 && q.sort_first && tag_on && (count >= 4u * kBlock || q.sort_first == 2u) && (uint64_t)sorted_at0 + count <= q.qcap) {
        for (uint32_t k = tid; k < count; k += kBlock) {
            const uint32_t it = my_queue[k];
            if (it != kPad) atomicAdd(&P.sub_cnt[(it & ~kXpFlag) >> kIdxBits], 1u);
        }
        __syncthreads();
        if (tid == 0) { uint32_t o = 0; for (int s2 = 0; s2 < kSubs; s2++) { P.sub_off[s2] = o; o += P.sub_cnt[s2]; } P.sub_off[kSubs] = o; }
        __syncthreads();
        uint32_t* sorted = my_queue + sorted_at0;
        for (uint32_t k = tid; k < count; k += kBlock) {
            const uint32_t it = my_queue[k];
            if (it == kPad) continue;
            const uint32_t s2 = (it & ~kXpFlag) >> kIdxBits;
            sorted[P.sub_off[s2] + atomicAdd(&P.sub_fill[s2], 1u)] = it;
        }
        drain_stores();
        __syncthreads();
        const bool grouped = t.defer_claims && !t.subflow;                 // sub-flow tables: every entry its own slot
        constexpr uint32_t kFillLimit = 832;                               // entries a cache takes before its probe windows start to overflow
        int s0 = 0, runs_in_cache = 0;
        while (s0 < kSubs) {
            // how many runs the next segment takes: one when nothing is known yet, else what the density seen in this cache allows
            const uint32_t fill = P.fill;
            __syncthreads();                                               // every lane has read it before the next round's claims raise it
            int take = 1;
            if (runs_in_cache > 0) {
                const uint32_t per_run = fill / (uint32_t)runs_in_cache + 1u;
                const uint32_t room = fill < kFillLimit ? kFillLimit - fill : 0u;
                take = (int)(room / (per_run + per_run / 4u));
                if (take == 0) {                                           // no room for another run: flush, fresh cache
                    __syncthreads();
                    if (grouped) parts_flush_grouped(t, sk, L, P, recs, seq_base32); else parts_flush(t, sk, L, recs, seq_base32);
                    cache_init(L, tid);
                    if (tid == 0) {
                        P.fill = 0;
                        P.new_cnt[3] = (t.defer_claims && ald(&t.ctr->n_live) + 512ull * kPartEntries <= t.claim_limit) ? 1u : 0u;
                    }
                    __syncthreads();
                    runs_in_cache = 0;
                    take = (int)(kFillLimit / (per_run + per_run / 4u));   // the density of the last cache is the best guess
                    if (take < 1) take = 1;
                }
            }
            int s1 = s0 + take < kSubs ? s0 + take : kSubs;
            const uint32_t lo = P.sub_off[s0], hi = P.sub_off[s1];
            if (hi > lo) parts_round<true, true, 0, false>(t, q, L, P, sorted + lo, hi - lo, recs, seq_base32, idx_mask, my_queue);
            else __syncthreads();
            runs_in_cache += s1 - s0;
            s0 = s1;
        }
        if (grouped) parts_flush_grouped(t, sk, L, P, recs, seq_base32); else parts_flush(t, sk, L, recs, seq_base32);
        const uint32_t m2 = P.retry_cnt;                                   // sub-flows of a run that overflowed a whole cache: claimed already
        if (m2 == 0) return;
        drain_stores();
        __syncthreads();
        parts_drain(t, sk, q, L, P, my_queue, m2, recs, seq_base32, idx_mask, max_rounds);
        return;
    }
    parts_round<true, false, ABL>(t, q, L, P, my_queue, count, recs, seq_base32, idx_mask);
    if (ABL) {
        if (!(ABL & 1)) parts_flush(t, sk, L, recs, seq_base32);
        return;
    }
    if (t.defer_claims && !t.subflow) parts_flush_grouped(t, sk, L, P, recs, seq_base32); // sub-flow tables: every entry its own slot
    else parts_flush(t, sk, L, recs, seq_base32);
    const uint32_t m = P.retry_cnt;
    if (m == 0) return;
    drain_stores();
    __syncthreads();                                              // everybody has read retry_cnt; the retry list is written
    // More sub-flows in the partition than the cache has entries. Sort what is left by sub-partition into the free tail of this
    // workgroup's queue region (counting sort, counts kept during the first round) and give every run of sub-partitions that
    // surely fits a cache one round of its own; a run that does not fit after all is drained in further rounds.
    const uint32_t sorted_at = (count + 3u) & ~3u;
    if (!tag_on || (uint64_t)sorted_at + m > q.qcap) {
        parts_drain(t, sk, q, L, P, my_queue, m, recs, seq_base32, idx_mask, max_rounds);
        return;
    }
    if (tid == 0) { uint32_t o = 0; for (int s = 0; s < kSubs; s++) { P.sub_off[s] = o; o += P.sub_cnt[s]; } P.sub_off[kSubs] = o; }
    __syncthreads();
    uint32_t* sorted = my_queue + sorted_at;
    for (uint32_t k = tid; k < m; k += kBlock) {
        const uint32_t it = ald(&my_queue[k]);
        const uint32_t s = (it & ~kXpFlag) >> kIdxBits;
        sorted[P.sub_off[s] + atomicAdd(&P.sub_fill[s], 1u)] = it;
    }
    drain_stores();
    __syncthreads();
    for (int s = 0; s < kSubs;) {
        uint32_t c = P.sub_cnt[s];
        int e = s + 1;
        while (e < kSubs && c + P.sub_cnt[e] <= kPackItems) { c += P.sub_cnt[e]; e++; }
        if (c) parts_drain(t, sk, q, L, P, sorted + P.sub_off[s], c, recs, seq_base32, idx_mask, max_rounds);
        s = e;
    }
}

// the (normally empty) overflow list of the streaming pass: one item per lane, straight on the table
template <bool FOLD>
__global__ __launch_bounds__(256) void k_dedup_overflow(TableView t, SketchView sk, SpillView q, const void* __restrict__ recs, uint64_t n, uint64_t seq_base) {
    const uint32_t idx_mask = n <= (uint64_t)kIdxMask ? kIdxMask : kIdxMaskUntagged;
    uint32_t count = *q.ovf_tail;
    if (count > q.ovf_cap) count = q.ovf_cap;
    unsigned long long direct = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t it = q.ovf[k];
        if (it == kPad) continue;
        Rec raw;
        load_item(q, recs, it, idx_mask, raw);
        Item x;
        decode_item(it, idx_mask, raw, (uint32_t)seq_base, x);
        if (FOLD) { fold_item(t, sk, x, recs, (uint32_t)seq_base); direct++; }
        else claim_item(t, x);
    }
    if (direct) aadd(&t.ctr->phase[7], direct);                 // diagnostics (nfagg_debug_phase_cycles[7] of libnfagg_diag.so): items through the overflow list
}

}  // namespace dcache

// Four launches: stream, overflow claims, partitions, overflow folds (+ the reset of the overflow tail). The sketches of sk (flags = 0:
// none) are fed by the partition pass's flushes and the overflow folds.
// variant 12 (A/B, tests): no retry rounds in the partition pass.
hipError_t launch_ingest_dedup_cached(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, uint64_t seq_base, int variant, hipStream_t s) {
    using namespace dcache;
    if (n == 0) return hipSuccess;
    const SpillView& q = t.spill;
    if (!t.aux || !q.queue || !q.qtail || !q.ovf || !q.ovf_tail || !q.xp || q.qcap < 4 || (q.qcap & 3u) || n >= (uint64_t)kXpFlag) return hipErrorInvalidValue;
    const size_t lds1 = sizeof(StreamCache) + sizeof(spill::Stage) + kDoorBits / 8,
                 lds2 = sizeof(FoldCache<kPartEntries>) + sizeof(PartsLds);
    static_assert(sizeof(StreamCache) + sizeof(spill::Stage) + kDoorBits / 8 <= 160 * 1024, "LDS of one CU");
    static_assert(sizeof(FoldCache<kPartEntries>) + sizeof(PartsLds) <= 160 * 1024, "LDS of one CU");
    static std::atomic<bool> attr_set_dev[64];   // per device: a process may drive several GPUs, from several host threads
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    std::atomic<bool>& attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_stream), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_parts<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
#ifdef NFAGG_DIAG
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_parts<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_parts<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_parts<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
#endif
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    const uint64_t tiles = (n + kBlock - 1) / kBlock;
    const unsigned grid = (unsigned)(tiles < (uint64_t)kStreamGrid ? tiles : (uint64_t)kStreamGrid);
    hipError_t e;
#define NF_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); e = hipGetLastError(); if (e != hipSuccess) return e; } while (0)
    NF_LAUNCH(k_dedup_stream, dim3(grid), dim3(kBlock), lds1, s, t, q, d_records, n, seq_base);
    NF_LAUNCH((k_dedup_overflow<false>), dim3(32), dim3(256), 0, s, t, sk, q, d_records, n, seq_base);
#ifdef NFAGG_DIAG
    if (variant == 13) NF_LAUNCH(k_dedup_parts<1>, dim3(kSpillParts), dim3(kBlock), lds2, s, t, sk, q, d_records, n, seq_base, 1);
    else if (variant == 14) NF_LAUNCH(k_dedup_parts<3>, dim3(kSpillParts), dim3(kBlock), lds2, s, t, sk, q, d_records, n, seq_base, 1);
    else if (variant == 15) NF_LAUNCH(k_dedup_parts<7>, dim3(kSpillParts), dim3(kBlock), lds2, s, t, sk, q, d_records, n, seq_base, 1);
    else
#endif
    NF_LAUNCH(k_dedup_parts<0>, dim3(kSpillParts), dim3(kBlock), lds2, s, t, sk, q, d_records, n, seq_base, variant == 12 ? 1 : kMaxRounds);
    NF_LAUNCH((k_dedup_overflow<true>), dim3(32), dim3(256), 0, s, t, sk, q, d_records, n, seq_base);
#undef NF_LAUNCH
    return hipMemsetAsync(q.ovf_tail, 0, sizeof(uint32_t), s);
}

}  // namespace nfagg
