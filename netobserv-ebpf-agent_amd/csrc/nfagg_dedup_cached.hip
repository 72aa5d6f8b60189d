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
// Sub-flows that get no entry (probe window full) take the direct per-record path.
// Merging DedupPartials is associative and commutative (everything is a sum, an OR, a
// max over sequence-tagged words, or a top-2 over sequence-tagged words), so the result is
// the same as the direct passes': bit-exact vs the oracle's sequential fold.
#include "nfagg_dedup.h"

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
            const uint32_t b = (uint32_t)(hs >> 14) & (DOORBITS - 1), m = 1u << (b & 31);
            if (!(door[b >> 5] & m) && !(atomicOr(&door[b >> 5], m) & m)) return -1;
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

__global__ __launch_bounds__(kBlock) void k_dedup_claim_cached(TableView t, const void* __restrict__ recs, uint64_t n, uint64_t seq_base) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ClaimCache<kClaimEntries>& L = *reinterpret_cast<ClaimCache<kClaimEntries>*>(lds_raw);
    uint32_t* door = reinterpret_cast<uint32_t*>(lds_raw + sizeof(ClaimCache<kClaimEntries>));
    const int tid = threadIdx.x;
    for (int e = tid; e < kClaimEntries; e += kBlock) { L.h64[e] = 0; L.min_seq[e] = 0xffffffffu; }
    for (int e = tid; e < kClaimDoorBits / 32; e += kBlock) door[e] = 0;
    __syncthreads();
    const uint64_t n_tiles = (n + kBlock - 1) / kBlock;
    unsigned long long skipped = 0;
    // software pipeline: the next tile's record is requested (unconditionally, on a clamped index) before this one is processed
    uint64_t i = (uint64_t)blockIdx.x * kBlock + tid;
    bool valid = i < n;
    Rec r;
    load_record(recs, valid ? i : 0, r);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t i_n = (tile + gridDim.x) * kBlock + tid;
        const bool valid_n = i_n < n;
        Rec r_n;
        load_record(recs, valid_n ? i_n : 0, r_n);
        uint64_t w[5], h = 0;
        if (valid && !record_keys(t, r, w, h)) { valid = false; skipped++; }
        const uint32_t seq32 = (uint32_t)(seq_base + i);
        const uint32_t ifx = valid ? r.d[21] : 0;
        const int ent0 = valid ? claim<ClaimCache<kClaimEntries>, kClaimEntries, kClaimDoorBits>(L, door, subflow_hash(h, ifx), w, ifx) : -1;
        __syncthreads();
        if (valid) {
            if (ent0 >= 0 && same_subflow(L, ent0, w, ifx)) {
                if (L.min_seq[ent0] > seq32) atomicMin(&L.min_seq[ent0], seq32);
            } else {
                dedup_claim_record(t, r, w, h, seq32);          // no entry for this sub-flow
            }
        }
        // the next tile's claims only write h64/key/ifx of NEW entries; min_seq reads/updates are ordered by its barrier
        r = r_n; valid = valid_n; i = i_n;
    }
    __syncthreads();
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
}

__global__ __launch_bounds__(kBlock) void k_dedup_fold_cached(TableView t, const void* __restrict__ recs, uint64_t n, uint64_t seq_base) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FoldCache<kFoldEntries>& L = *reinterpret_cast<FoldCache<kFoldEntries>*>(lds_raw);
    uint32_t* door = reinterpret_cast<uint32_t*>(lds_raw + sizeof(FoldCache<kFoldEntries>));
    const int tid = threadIdx.x;
    for (int e = tid; e < kFoldDoorBits / 32; e += kBlock) door[e] = 0;
    for (int e = tid; e < kFoldEntries; e += kBlock) {
        L.h64[e] = 0; L.bytes[e] = 0; L.endl_lo[e] = 0; L.endl_hi[e] = 0; L.dscp_tag[e] = 0; L.samp_tag[e] = 0;
        L.ssl_first[e] = 0; L.cs_tag[e] = 0; L.ks_tag[e] = 0; L.dir[0][e] = 0; L.dir[1][e] = 0;
        L.packets[e] = 0; L.flags[e] = 0; L.ssl_max[e] = 0; L.ssl_minv[e] = 0; L.min_seq[e] = 0xffffffffu;
    }
    __syncthreads();
    const uint64_t n_tiles = (n + kBlock - 1) / kBlock;
    unsigned long long direct = 0;
    // software pipeline, as in the claim pass
    uint64_t i = (uint64_t)blockIdx.x * kBlock + tid;
    bool valid = i < n;
    Rec r;
    load_record(recs, valid ? i : 0, r);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t i_n = (tile + gridDim.x) * kBlock + tid;
        const bool valid_n = i_n < n;
        Rec r_n;
        load_record(recs, valid_n ? i_n : 0, r_n);
        uint64_t w[5], h = 0;
        if (valid && !record_keys(t, r, w, h)) valid = false;
        const uint32_t seq32 = (uint32_t)(seq_base + i);
        const uint32_t ifx = valid ? r.d[21] : 0;
        const int ent0 = valid ? claim<FoldCache<kFoldEntries>, kFoldEntries, kFoldDoorBits>(L, door, subflow_hash(h, ifx), w, ifx) : -1;
        __syncthreads();
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
            } else {
                direct++;
                dedup_fold_record(t, r, w, h, seq32);            // no entry for this sub-flow
            }
        }
        r = r_n; valid = valid_n; i = i_n;
    }
    __syncthreads();
    for (int e = tid; e < kFoldEntries; e += kBlock) {
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
    if (direct) aadd(&t.ctr->n_bypassed, direct);
}

}  // namespace dcache

hipError_t launch_ingest_dedup_cached(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (!t.aux) return hipErrorInvalidValue;
    const size_t lds1 = sizeof(dcache::ClaimCache<dcache::kClaimEntries>) + dcache::kClaimDoorBits / 8,
                 lds2 = sizeof(dcache::FoldCache<dcache::kFoldEntries>) + dcache::kFoldDoorBits / 8;
    static_assert(sizeof(dcache::FoldCache<dcache::kFoldEntries>) + dcache::kFoldDoorBits / 8 <= 160 * 1024, "LDS of one CU");
    static bool attr_set_dev[64] = {};   // per device: a process may drive several GPUs
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcache::k_dedup_claim_cached), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcache::k_dedup_fold_cached), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const uint64_t tiles = (n + dcache::kBlock - 1) / dcache::kBlock;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(dcache::k_dedup_claim_cached, dim3(grid), dim3(dcache::kBlock), lds1, s, t, d_records, n, seq_base);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dcache::k_dedup_fold_cached, dim3(grid), dim3(dcache::kBlock), lds2, s, t, d_records, n, seq_base);
    return hipGetLastError();
}

}  // namespace nfagg
