// nfagg_dedup_cached.hip — kernel-dedup mode with persistent LDS caches (default for
// NFAGG_MODE_KERNEL_DEDUP; the direct kernels of nfagg_dedup.hip serve small batches).
//
// The direct passes touch the table once per record: on a hot flow (BASELINE configs[4]:
// 90 % of the records are one flow seen on two interfaces) every record's atomics hit one
// slot and serialise. Here each workgroup keeps a cache keyed by the SUB-FLOW
// (flow key, if_index_first_seen) for its whole lifetime:
//   pass 1  k_dedup_claim_cached : an entry only tracks the smallest sequence number of
//           its sub-flow; the flush does the table work (claim, first record, earliest
//           interfaces) once per entry.
//   pass 2  k_dedup_fold_cached  : an entry is a DedupPartial (nfagg_dedup.h) — sums, ORs,
//           "last value" tags and the two earliest distinct directions, folded with LDS
//           atomics; whether the sub-flow is the counted one (if_index == F) or a side
//           interface is decided at the flush, when the slot's first record is known.
// Records whose sub-flow gets no entry are SPILLED to the queue of their flow's partition and folded by a second,
// per-partition launch of the same kernel (QUEUE == true), exactly as in accounter mode (nfagg_ingest_part.hip); only
// what finds no entry even there (probe window full) is merged record by record.
// Merging DedupPartials is associative and commutative (everything is a sum, an OR, a
// max over sequence-tagged words, or a top-2 over sequence-tagged words), so the result is
// the same as the direct passes': bit-exact vs the oracle's sequential fold.
#include "nfagg_dedup.h"
#include "nfagg_spill.h"

namespace nfagg {
namespace dcache {

constexpr int kBlock = 1024;
constexpr int kProbe = 8;

NF_DEV uint64_t subflow_hash(uint64_t h, uint32_t ifx) {
    uint64_t z = (h ^ ((uint64_t)ifx * 0xD6E8FEB86659FD93ull)) * kMul;
    return (z ^ (z >> 32)) | 1ull;
}

template <int K>
struct ClaimCache {
    uint64_t h64[K];
    uint64_t key[5][K];
    uint32_t ifx[K];
    uint32_t min_seq[K];
};

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

// find or claim the entry of sub-flow hash hs; the creator writes key and interface. -1 = window full, or the
// sub-flow is seen for the first time: entries are never evicted, so a sub-flow is admitted on its second
// appearance (admission filter `door`, DOORBITS bits of LDS, as in nfagg_ingest_part.hip) — one-off sub-flows of
// the cold tail do not take the entries of the hot ones. Exactly one of the lanes that meet a new sub-flow in
// the same tile is turned away (the atomic's return value decides).
template <typename Cache, int K, int DOORBITS>
NF_DEV int claim(Cache& L, uint32_t* door, uint64_t hs, const uint64_t w[5], uint32_t ifx) {
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
template <int K>
NF_DEV void lds_dir_insert(FoldCache<K>& L, int ent, uint64_t v) {
    for (int trip = 0; trip < 64; trip++) {
        const uint64_t c0 = L.dir[0][ent], c1 = L.dir[1][ent];
        const bool m0 = c0 != 0 && ((c0 ^ v) & 0xffull) == 0, m1 = c1 != 0 && ((c1 ^ v) & 0xffull) == 0;
        int pos; uint64_t cur;
        if (m0) { pos = 0; cur = c0; } else if (m1) { pos = 1; cur = c1; } else if (c1 < c0) { pos = 1; cur = c1; } else { pos = 0; cur = c0; }
        if (cur >= v) return;
        if (atomicCAS((unsigned long long*)&L.dir[pos][ent], (unsigned long long)cur, (unsigned long long)v) == cur) return;
    }
}

constexpr int kClaimEntries = 2048;
constexpr int kFoldEntries = 1024;
constexpr int kClaimDoorBits = 65536, kFoldDoorBits = 32768;   // 8 KiB / 4 KiB of LDS behind the caches

// Tile source of both passes. QUEUE == false: pass 1, tiles of consecutive records, grid-strided. QUEUE == true: pass 2,
// workgroup b walks the record indices pass 1 queued for partition b (0xffffffff = padding of a partial group).
template <bool QUEUE>
struct Tiles {
    uint64_t count, n_tiles, tile0, step;
    const uint32_t* queue;
    NF_DEV bool setup(const SpillView& q, uint64_t n) {
        if (QUEUE) {
            const uint32_t tail = q.qtail[blockIdx.x];            // written by pass 1 (previous kernel)
            count = tail < q.qcap ? tail : q.qcap;
            queue = q.queue + (uint64_t)blockIdx.x * q.qcap;
            tile0 = 0; step = 1;
        } else {
            count = n; queue = nullptr; tile0 = blockIdx.x; step = gridDim.x;
        }
        n_tiles = (count + kBlock - 1) / kBlock;
        return count != 0;
    }
    // record index of this lane in `tile` (valid == false: nothing to do)
    NF_DEV uint64_t index(uint64_t tile, int tid, bool& valid) const {
        const uint64_t pos = tile * kBlock + tid;
        valid = pos < count;
        if (!QUEUE) return valid ? pos : 0;
        const uint32_t qi = valid ? queue[pos] : 0xffffffffu;
        valid = qi != 0xffffffffu;
        return valid ? qi : 0;
    }
};

// ---- pass 1 of the dedup merge (first record + earliest interfaces), LDS-cached; misses are spilled (QUEUE == false)
// or, in the partition pass, claimed record by record (probe window full: rare)
template <bool QUEUE>
__global__ __launch_bounds__(kBlock) void k_dedup_claim_cached(TableView t, SpillView q, const void* __restrict__ recs, uint64_t n, uint64_t seq_base) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ClaimCache<kClaimEntries>& L = *reinterpret_cast<ClaimCache<kClaimEntries>*>(lds_raw);
    spill::Stage& S = *reinterpret_cast<spill::Stage*>(lds_raw + sizeof(ClaimCache<kClaimEntries>));                    // pass 1 only
    uint32_t* door = reinterpret_cast<uint32_t*>(lds_raw + sizeof(ClaimCache<kClaimEntries>) + sizeof(spill::Stage));   // pass 1 only
    const int tid = threadIdx.x;
    Tiles<QUEUE> T;
    if (!T.setup(q, n)) return;                                   // uniform for the workgroup
    for (int e = tid; e < kClaimEntries; e += kBlock) { L.h64[e] = 0; L.min_seq[e] = 0xffffffffu; }
    spill::Lane<kBlock> sp;
    if (!QUEUE) {
        for (int e = tid; e < kClaimDoorBits / 32; e += kBlock) door[e] = 0;
        sp.init(S, tid);
    }
    __syncthreads();
    if (QUEUE && tid == 0) q.qtail[blockIdx.x] = 0;               // every lane has read it: ready for the fold pass
    unsigned long long skipped = 0, spilled = 0;
    // software pipeline: the next tile's record is requested (unconditionally, on a clamped index) before this one is processed
    bool valid;
    uint64_t i = T.index(T.tile0, tid, valid);
    Rec r;
    load_record(recs, i, r);
    for (uint64_t tile = T.tile0; tile < T.n_tiles; tile += T.step) {
        bool valid_n = false;
        uint64_t i_n = 0;
        if (tile + T.step < T.n_tiles) i_n = T.index(tile + T.step, tid, valid_n);
        Rec r_n;
        load_record(recs, i_n, r_n);
        uint64_t w[5], h = 0;
        if (valid && !record_keys(t, r, w, h)) { valid = false; skipped++; }
        const uint32_t seq32 = (uint32_t)(seq_base + i);
        const uint32_t ifx = valid ? r.d[21] : 0;
        int ent0 = -1;
        if (valid) ent0 = QUEUE ? claim<ClaimCache<kClaimEntries>, kClaimEntries, 0>(L, nullptr, subflow_hash(h, ifx), w, ifx)
                                : claim<ClaimCache<kClaimEntries>, kClaimEntries, kClaimDoorBits>(L, door, subflow_hash(h, ifx), w, ifx);
        __syncthreads();
        bool miss = false;
        if (valid) {
            if (ent0 >= 0 && same_subflow(L, ent0, w, ifx)) {
                if (L.min_seq[ent0] > seq32) atomicMin(&L.min_seq[ent0], seq32);
            } else if (QUEUE) {
                dedup_claim_record(t, r, w, h, seq32);          // no entry even in the partition's cache
            } else {
                miss = true;
            }
        }
        if (!QUEUE) {
            sp.drain(S, q, tid);
            __syncthreads();
            if (miss) spilled++;
            sp.append(S, q, miss, spill::part_of(h, q.part_shift), (uint32_t)i);
        }
        // the next tile's claims only write h64/key/ifx of NEW entries; min_seq reads/updates are ordered by its barrier
        r = r_n; valid = valid_n; i = i_n;
    }
    if (!QUEUE) sp.finish(S, q, tid); else __syncthreads();
    for (int e = tid; e < kClaimEntries; e += kBlock) {
        if (L.h64[e] == 0 || L.min_seq[e] == 0xffffffffu) continue;
        uint64_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = L.key[k][e];
        const uint64_t h = key_hash(w);
        Hints x;
        uint32_t idx = probe_home(t, w, h, x);
        if (idx == kNoSlot) {
            idx = find_or_claim(t, w, h);
            if (idx == kNoSlot) continue;
            x.id0 = 0;
        }
        dedup_claim(t, idx, x.id0, L.ifx[e], L.min_seq[e]);
    }
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
    (void)spilled;                                                // the fold pass reports the bypass count (stats.records_bypassed)
}

// ---- pass 2 of the dedup merge (sums, tags, directions), LDS-cached with K entries; misses as above
template <bool QUEUE, int K>
__global__ __launch_bounds__(kBlock) void k_dedup_fold_cached(TableView t, SpillView q, const void* __restrict__ recs, uint64_t n, uint64_t seq_base) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FoldCache<K>& L = *reinterpret_cast<FoldCache<K>*>(lds_raw);
    spill::Stage& S = *reinterpret_cast<spill::Stage*>(lds_raw + sizeof(FoldCache<K>));                    // pass 1 only
    uint32_t* door = reinterpret_cast<uint32_t*>(lds_raw + sizeof(FoldCache<K>) + sizeof(spill::Stage));   // pass 1 only
    const int tid = threadIdx.x;
    Tiles<QUEUE> T;
    if (!T.setup(q, n)) return;
    for (int e = tid; e < K; e += kBlock) {
        L.h64[e] = 0; L.bytes[e] = 0; L.endl_lo[e] = 0; L.endl_hi[e] = 0; L.dscp_tag[e] = 0; L.samp_tag[e] = 0;
        L.ssl_first[e] = 0; L.cs_tag[e] = 0; L.ks_tag[e] = 0; L.dir[0][e] = 0; L.dir[1][e] = 0;
        L.packets[e] = 0; L.flags[e] = 0; L.ssl_max[e] = 0; L.ssl_minv[e] = 0; L.min_seq[e] = 0xffffffffu;
    }
    spill::Lane<kBlock> sp;
    if (!QUEUE) {
        for (int e = tid; e < kFoldDoorBits / 32; e += kBlock) door[e] = 0;
        sp.init(S, tid);
    }
    __syncthreads();
    if (QUEUE && tid == 0) q.qtail[blockIdx.x] = 0;
    unsigned long long spilled = 0;
    bool valid;
    uint64_t i = T.index(T.tile0, tid, valid);
    Rec r;
    load_record(recs, i, r);
    for (uint64_t tile = T.tile0; tile < T.n_tiles; tile += T.step) {
        bool valid_n = false;
        uint64_t i_n = 0;
        if (tile + T.step < T.n_tiles) i_n = T.index(tile + T.step, tid, valid_n);
        Rec r_n;
        load_record(recs, i_n, r_n);
        uint64_t w[5], h = 0;
        if (valid && !record_keys(t, r, w, h)) valid = false;
        const uint32_t seq32 = (uint32_t)(seq_base + i);
        const uint32_t ifx = valid ? r.d[21] : 0;
        int ent0 = -1;
        if (valid) ent0 = QUEUE ? claim<FoldCache<K>, K, 0>(L, nullptr, subflow_hash(h, ifx), w, ifx)
                                : claim<FoldCache<K>, K, kFoldDoorBits>(L, door, subflow_hash(h, ifx), w, ifx);
        __syncthreads();
        bool miss = false;
        if (valid) {
            if (ent0 >= 0 && same_subflow(L, ent0, w, ifx)) {
                const int e = ent0;
                DedupPartial p;
                dedup_partial_from_record(r, seq32, p);
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
                lds_dir_insert(L, e, p.dir0);
                if (L.min_seq[e] > seq32) atomicMin(&L.min_seq[e], seq32);
            } else if (QUEUE) {
                dedup_fold_record(t, r, w, h, seq32);            // no entry even in the partition's cache
            } else {
                miss = true;
            }
        }
        if (!QUEUE) {
            sp.drain(S, q, tid);
            __syncthreads();
            if (miss) spilled++;
            sp.append(S, q, miss, spill::part_of(h, q.part_shift), (uint32_t)i);
        }
        r = r_n; valid = valid_n; i = i_n;
    }
    if (!QUEUE) sp.finish(S, q, tid); else __syncthreads();
    for (int e = tid; e < K; e += kBlock) {
        if (L.h64[e] == 0 || L.min_seq[e] == 0xffffffffu) continue;
        uint64_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = L.key[k][e];
        const uint64_t h = key_hash(w);
        Hints x;
        uint32_t idx = probe_home(t, w, h, x);
        if (idx == kNoSlot) {
            idx = find_or_claim(t, w, h);                        // pass 1 claimed it
            if (idx == kNoSlot) continue;
            load_hints(&t.hot[idx], x);
        }
        const uint32_t ms = L.min_seq[e];
        if ((uint32_t)(x.id0 >> 32) == ~ms) {
            // this sub-flow's earliest record is the flow's first record: fetch it again and store it whole
            Rec r;
            load_record(recs, (uint64_t)(ms - (uint32_t)seq_base), r);
            r.canonicalize();
            dedup_publish_first(t, idx, r, ms);
        }
        DedupPartial p;
        p.bytes = L.bytes[e]; p.packets = L.packets[e]; p.flags = L.flags[e];
        p.endl_lo = L.endl_lo[e]; p.endl_hi = L.endl_hi[e]; p.dscp_tag = L.dscp_tag[e]; p.samp_tag = L.samp_tag[e];
        p.ssl_first = L.ssl_first[e]; p.ssl_max = L.ssl_max[e]; p.ssl_minv = L.ssl_minv[e];
        p.cs_tag = L.cs_tag[e]; p.ks_tag = L.ks_tag[e];
        p.dir0 = L.dir[0][e]; p.dir1 = L.dir[1][e];
        p.ifx = L.ifx[e];
        dedup_merge(t, idx, x, p);
    }
    if (spilled) aadd(&t.ctr->n_bypassed, spilled);
}

// the (normally empty) overflow list of a pass: one record per lane, merged directly
template <bool FOLD>
__global__ __launch_bounds__(256) void k_dedup_overflow(TableView t, SpillView q, const void* __restrict__ recs, uint64_t seq_base) {
    uint32_t count = *q.ovf_tail;
    if (count > q.ovf_cap) count = q.ovf_cap;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = q.ovf[k];
        if (i == 0xffffffffu) continue;
        Rec r; uint64_t w[5], h = 0;
        if (!record_prologue(t, recs, i, r, w, h)) continue;
        if (FOLD) dedup_fold_record(t, r, w, h, (uint32_t)(seq_base + i));
        else dedup_claim_record(t, r, w, h, (uint32_t)(seq_base + i));
    }
}

}  // namespace dcache

// Six launches: claim pass 1 / its partitions / its overflow, then the same for the fold — the fold needs every flow's first
// record (F) resolved, i.e. the whole claim pass finished.
hipError_t launch_ingest_dedup_cached(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base, hipStream_t s) {
    using namespace dcache;
    if (n == 0) return hipSuccess;
    const SpillView& q = t.spill;
    if (!t.aux || !q.queue || !q.qtail || !q.ovf || !q.ovf_tail || q.qcap < 4 || (q.qcap & 3u)) return hipErrorInvalidValue;
    constexpr int kFoldEntries1 = 512;      // pass 1 shares the LDS with the spill staging
    const size_t lds_c1 = sizeof(ClaimCache<kClaimEntries>) + sizeof(spill::Stage) + kClaimDoorBits / 8, lds_c2 = sizeof(ClaimCache<kClaimEntries>),
                 lds_f1 = sizeof(FoldCache<kFoldEntries1>) + sizeof(spill::Stage) + kFoldDoorBits / 8, lds_f2 = sizeof(FoldCache<kFoldEntries>);
    static_assert(sizeof(ClaimCache<kClaimEntries>) + sizeof(spill::Stage) + kClaimDoorBits / 8 <= 160 * 1024, "LDS of one CU");
    static_assert(sizeof(FoldCache<kFoldEntries>) <= 160 * 1024 && sizeof(FoldCache<kFoldEntries1>) + sizeof(spill::Stage) + kFoldDoorBits / 8 <= 160 * 1024, "LDS of one CU");
    static bool attr_set_dev[64] = {};   // per device: a process may drive several GPUs
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_claim_cached<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c1);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_claim_cached<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c2);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_fold_cached<false, kFoldEntries1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f1);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dedup_fold_cached<true, kFoldEntries>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f2);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const uint64_t tiles = (n + kBlock - 1) / kBlock;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    hipError_t e;
#define NF_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); e = hipGetLastError(); if (e != hipSuccess) return e; } while (0)
    NF_LAUNCH((k_dedup_claim_cached<false>), dim3(grid), dim3(kBlock), lds_c1, s, t, q, d_records, n, seq_base);
    NF_LAUNCH((k_dedup_claim_cached<true>), dim3(kSpillParts), dim3(kBlock), lds_c2, s, t, q, d_records, n, seq_base);
    NF_LAUNCH((k_dedup_overflow<false>), dim3(256), dim3(256), 0, s, t, q, d_records, seq_base);
    e = hipMemsetAsync(q.ovf_tail, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    NF_LAUNCH((k_dedup_fold_cached<false, kFoldEntries1>), dim3(grid), dim3(kBlock), lds_f1, s, t, q, d_records, n, seq_base);
    NF_LAUNCH((k_dedup_fold_cached<true, kFoldEntries>), dim3(kSpillParts), dim3(kBlock), lds_f2, s, t, q, d_records, n, seq_base);
    NF_LAUNCH((k_dedup_overflow<true>), dim3(256), dim3(256), 0, s, t, q, d_records, seq_base);
#undef NF_LAUNCH
    return hipMemsetAsync(q.ovf_tail, 0, sizeof(uint32_t), s);
}

}  // namespace nfagg
