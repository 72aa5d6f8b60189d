// nfagg_epoch_par.hip — the evict-on-full loop of Accounter.Account (pkg/flow/account.go:81-96) WITHOUT its sequential chain
// (ingest_variant 31 of nfagg_account[_device]; the rule is pinned on the CPU by tests/test_epoch_boundaries.py).
//
// The loop is sequential only in WHERE the epochs end. With prev(i) = the index of the previous record of record i's flow in the
// call (-1: none), record i >= s starts a new flow in the epoch that began at record s exactly when prev(i) < s (first epoch of a
// call: and the flow is not live in the table), so the epoch ends at the record where the count of such records reaches
// max_entries + 1. Given the cuts every epoch is an independent, order-free fold, so the complete epochs in the
// middle of a call are folded TOGETHER: their records are copied with the epoch's number (mod 256 within a group of at most 255
// epochs) in key byte 39 — Go's blank field, which no key comparison of the reference sees and every kernel here clears — into a
// scratch batch, folded by the ordinary two-pass kernels into a scratch table whose keys keep that byte (K39), evicted by the
// ordinary eviction kernel and put into epoch order (byte 39 cleared again) by k_par_regroup. The first epoch of the call (it
// continues what the table holds) and the last, incomplete one (it stays live) go through the ordinary ingest path.
//
//   k_par_hash      (key hash, index) per record
//   rocPRIM radix sort of the pairs by hash (stable: equal hashes stay in index order)
//   k_par_links     prev(i) from neighbours in the sorted order; full keys compared: two flows with one 64-bit hash raise a flag and
//                   the call takes the kernel chain instead
//   k_par_live      first occurrences whose flow is live in the table: prev = -2 (not new in the first epoch)
//   k_par_cuts      ONE workgroup walks the epochs: a prefix count over prev[] from the epoch's first record on
//   k_par_tag_copy  records of a group of complete epochs -> scratch batch, epoch number in byte 39
//   k_par_regroup   evicted scratch flows -> the caller's buffer, epoch by epoch, byte 39 cleared
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "nfagg_device.h"

namespace nfagg {

constexpr int kParBlock = 256;
static inline int par_grid(uint64_t n, int per_block = kParBlock, int cap = 8192) {
    uint64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > (uint64_t)cap) g = cap;
    return (int)g;
}

NF_DEV void par_key(const void* recs, uint64_t i, uint64_t w[5]) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + i * kRecordBytes);
    const uint4 a = p[0], b = p[1], c = p[2];
    w[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); w[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    w[2] = (uint64_t)b.x | ((uint64_t)b.y << 32); w[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
    w[4] = ((uint64_t)c.x | ((uint64_t)c.y << 32)) & 0x00FFFFFFFFFFFFFFull;      // key byte 39: Go's blank field, not part of the key
}

__global__ __launch_bounds__(kParBlock) void k_par_hash(const void* __restrict__ recs, uint64_t n, uint64_t* __restrict__ hash,
                                                        uint32_t* __restrict__ idx) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t w[5];
        par_key(recs, i, w);
        hash[i] = key_hash(w);
        idx[i] = (uint32_t)i;
    }
}

// sorted position p: the record idx_s[p]. Same hash as its left neighbour -> same flow (checked) -> that neighbour is its previous
// occurrence (the sort is stable: equal hashes are in index order).
__global__ __launch_bounds__(kParBlock) void k_par_links(const void* __restrict__ recs, const uint64_t* __restrict__ hash_s,
                                                         const uint32_t* __restrict__ idx_s, uint64_t n, int32_t* __restrict__ prev,
                                                         uint32_t* __restrict__ collision) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const uint32_t i = idx_s[p];
        int32_t pv = -1;
        if (p > 0 && hash_s[p] == hash_s[p - 1]) {
            const uint32_t j = idx_s[p - 1];
            uint64_t a[5], b[5];
            par_key(recs, i, a); par_key(recs, j, b);
            const bool same = ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3]) | (a[4] ^ b[4])) == 0;
            if (same) pv = (int32_t)j;
            else atomicExch(collision, 1u);                          // two flows, one 64-bit hash: the links of this call are not to be trusted
        }
        prev[i] = pv;
    }
}

// c.entries[key] without inserting (the table is quiescent: no fold is running). Probing as find_or_claim does: a slot whose tag
// is not of this epoch ends the chain.
NF_DEV bool par_is_live(const TableView& t, const uint64_t w[5], uint64_t h) {
    const uint64_t ready = tag_ready(t, h);
    uint64_t idx = h & t.mask;
    for (uint64_t probes = 0; probes <= t.mask; probes++) {
        const SlotHot* s = &t.hot[idx];
        const uint64_t tag = ald(&s->tag);
        if (tag_is_free(t, tag)) return false;
        if (tag == ready) {
            bool eq = true;
#pragma unroll
            for (int k = 0; k < 5; k++) eq &= (ald(&s->key[k]) == w[k]);
            if (eq) return true;
        }
        idx = (idx + 1) & t.mask;
    }
    return false;
}

__global__ __launch_bounds__(kParBlock) void k_par_live(TableView t, const void* __restrict__ recs, int32_t* __restrict__ prev, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (prev[i] != -1) continue;
        uint64_t w[5];
        par_key(recs, i, w);
        if (par_is_live(t, w, key_hash(w))) prev[i] = -2;
    }
}

// ONE workgroup. Epoch by epoch: from its first record s on, count the records that start a new flow in it — prev < s; in the
// first epoch of the call: prev == -1 (a flow the table holds, prev == -2, is no new entry) — until the count passes `budget`
// (max_entries less what is live, for the first epoch): that record finds the map full (account.go:85): the epoch is evicted and
// the record opens the next one. cuts[k] = that record; at most max_cuts of them (*n_cuts says how many were found; the walk stops
// there, and the caller treats the rest of the call as the last, incomplete epoch of this launch).
// (Tried and dropped, per 8 M-record call with 557 epochs, this version 3.2 ms: a streaming walk by the same workgroup — the call
// walked once, the next step's values in flight, every epoch end inside a step found without new loads — 4.2 ms: a step costs
// ~4 us whatever it loads, and the streaming walk makes steps + cuts of them; ONE wave walking 4096 records per step with
// wave-uniform counts, no LDS and no barrier — 49 ms: 64 rows per lane in registers, nothing to hide a latency behind.)
constexpr int kCutBlock = 1024;
constexpr int kCutPer = 16;                                           // records per lane and step: 16 Ki records per step — about one epoch at 5000 entries
__global__ __launch_bounds__(kCutBlock) void k_par_cuts(const int32_t* __restrict__ prev, uint64_t n, uint32_t max_entries, uint32_t live0,
                                                        uint32_t* __restrict__ cuts, uint32_t max_cuts, uint32_t* __restrict__ n_cuts) {
    constexpr int kWaves = kCutBlock / 64;
    __shared__ uint32_t cnt[kCutPer * kWaves];                        // new flows per (row j, wave): record order is row-major
    __shared__ uint32_t found;                                        // index of the record that ends the epoch, or 0xffffffff
    __shared__ uint32_t step_total;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t s = 0;
    uint32_t k = 0;
    while (k < max_cuts && s < n) {
        const bool first = k == 0;                                    // the epoch the table's live flows belong to (it may end at record 0)
        uint32_t budget = max_entries;
        if (first) budget = live0 >= max_entries ? 0u : max_entries - live0;
        uint32_t before = 0;                                          // new flows of the epoch in the steps already walked
        bool ended = false;
        for (uint64_t t0 = s; t0 < n; t0 += (uint64_t)kCutPer * kCutBlock) {
            // row j of the step: records t0 + j * 1024 + tid; all sixteen loads of a lane are in flight together
            int32_t pv[kCutPer];
#pragma unroll
            for (int j = 0; j < kCutPer; j++) {
                const uint64_t i = t0 + (uint64_t)j * kCutBlock + tid;
                pv[j] = i < n ? prev[i] : 0x7fffffff;                 // beyond the call: never new
            }
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < kCutPer; j++) {
                const bool is_new = first ? (pv[j] == -1) : ((int64_t)pv[j] < (int64_t)s);
                bits |= (is_new ? 1u : 0u) << j;
                const unsigned long long m = __ballot(is_new);
                if (lane == 0) cnt[j * kWaves + wv] = (uint32_t)__popcll(m);
            }
            if (tid == 0) found = 0xffffffffu;
            __syncthreads();
            if (tid == 0) {                                           // exclusive prefix over the 256 (row, wave) counts, in record order
                uint32_t run = 0;
                for (int q = 0; q < kCutPer * kWaves; q++) { const uint32_t c = cnt[q]; cnt[q] = run; run += c; }
                step_total = run;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kCutPer; j++) {
                const bool is_new = (bits >> j) & 1u;
                const unsigned long long m = __ballot(is_new);
                const uint32_t mine = before + cnt[j * kWaves + wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));   // new flows of the epoch before this record
                if (is_new && mine == budget) found = (uint32_t)(t0 + (uint64_t)j * kCutBlock + tid);   // entry number budget + 1: exactly one record
            }
            __syncthreads();
            const uint32_t f = found, tot = step_total;
            __syncthreads();                                          // (found, cnt and step_total are rewritten in the next step)
            if (f != 0xffffffffu) {
                if (tid == 0) cuts[k] = f;
                k++;
                s = f;
                ended = true;
                break;
            }
            before += tot;
        }
        if (!ended) break;                                            // the call ends inside this epoch
    }
    if (tid == 0) *n_cuts = k;
}

// Records [first, first + m) of the call belong to the epochs e0 .. e0 + n_ep - 1 (cuts[e] = first record of epoch e + 1, i.e.
// epoch e + 1 starts at cuts[e]; epoch e0 starts at `first`): copy them to dst with the epoch's number within the group in key
// byte 39.
__global__ __launch_bounds__(kParBlock) void k_par_tag_copy(const void* __restrict__ recs, uint64_t first, uint64_t m,
                                                            const uint32_t* __restrict__ cuts, uint32_t e0, uint32_t n_ep,
                                                            void* __restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += stride) {
        const uint64_t i = first + r;
        // the epoch of record i: the number of cuts at or below i, among cuts[e0 .. e0 + n_ep - 1) (epoch e0 + j starts at cuts[e0 + j - 1])
        uint32_t lo = 0, hi = n_ep - 1;                               // j in [0, n_ep - 1]: the largest j with start(j) <= i
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if ((uint64_t)cuts[e0 + mid - 1] <= i) lo = mid; else hi = mid - 1;
        }
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + i * kRecordBytes);
        uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + r * kRecordBytes);
        uint4 v[9];
#pragma unroll
        for (int q = 0; q < 9; q++) v[q] = src[q];
        v[2].y = (v[2].y & 0x00ffffffu) | (lo << 24);                 // dword 9, byte 39
#pragma unroll
        for (int q = 0; q < 9; q++) o[q] = v[q];
    }
}

// Evicted scratch flows (any order; byte 39 = the epoch's number in the group) -> out, epoch j at out[j * max_entries ...). A block
// takes kRegroupPer consecutive flows per thread: it counts them by epoch in LDS (the LDS atomic hands every flow its rank inside
// the block), reserves one range per epoch with ONE global atomic (a returning atomic on one address retires every ~14 ns: one
// per flow was 0.6 ms per group), and writes.
constexpr int kRegroupPer = 8;
__global__ __launch_bounds__(kParBlock) void k_par_regroup(const void* __restrict__ ev, uint64_t n_flows, uint32_t max_entries, uint32_t n_ep,
                                                           void* __restrict__ out, uint32_t* __restrict__ cnt, uint32_t* __restrict__ bad) {
    __shared__ uint32_t lcnt[256], lbase[256];
    const uint64_t chunk = (uint64_t)kParBlock * kRegroupPer;
    for (uint64_t c0 = (uint64_t)blockIdx.x * chunk; c0 < n_flows; c0 += (uint64_t)gridDim.x * chunk) {
        lcnt[threadIdx.x] = 0;                                        // kParBlock == 256 epochs' worth of counters
        __syncthreads();
        uint32_t ep[kRegroupPer], rank[kRegroupPer];
#pragma unroll
        for (int q = 0; q < kRegroupPer; q++) {
            const uint64_t f = c0 + (uint64_t)q * kParBlock + threadIdx.x;
            ep[q] = 0xffffffffu; rank[q] = 0;
            if (f < n_flows) {
                const uint32_t d9 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ev) + f * kRecordBytes)[9];
                const uint32_t j = d9 >> 24;
                if (j >= n_ep) atomicExch(bad, 1u);
                else { ep[q] = j; rank[q] = atomicAdd(&lcnt[j], 1u); }
            }
        }
        __syncthreads();
        {
            const uint32_t c = lcnt[threadIdx.x];
            lbase[threadIdx.x] = c ? atomicAdd(&cnt[threadIdx.x], c) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kRegroupPer; q++) {
            if (ep[q] == 0xffffffffu) continue;
            const uint64_t f = c0 + (uint64_t)q * kParBlock + threadIdx.x;
            const uint32_t at = lbase[ep[q]] + rank[q];
            if (at >= max_entries) { atomicExch(bad, 2u); continue; }  // an epoch of the middle holds exactly max_entries flows
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(ev) + f * kRecordBytes);
            uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + ((uint64_t)ep[q] * max_entries + at) * kRecordBytes);
            uint4 v[9];
#pragma unroll
            for (int k = 0; k < 9; k++) v[k] = src[k];
            v[2].y &= 0x00ffffffu;                                    // byte 39 back to what the reference sees
#pragma unroll
            for (int k = 0; k < 9; k++) o[k] = v[k];
        }
        __syncthreads();                                              // lcnt / lbase are reused by the next chunk
    }
}

// ---- launch wrappers ----------------------------------------------------------------------------------------------------------
hipError_t launch_par_hash(const void* d_records, uint64_t n, uint64_t* d_hash, uint32_t* d_idx, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_hash, dim3(par_grid(n)), dim3(kParBlock), 0, s, d_records, n, d_hash, d_idx);
    return hipGetLastError();
}

// temp == nullptr: only *temp_bytes is written
hipError_t launch_par_sort(void* temp, size_t* temp_bytes, const uint64_t* k_in, uint64_t* k_out, const uint32_t* v_in, uint32_t* v_out,
                           uint64_t n, hipStream_t s) {
    return rocprim::radix_sort_pairs(temp, *temp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0, 64, s);
}

hipError_t launch_par_links(const void* d_records, const uint64_t* d_hash_s, const uint32_t* d_idx_s, uint64_t n, int32_t* d_prev,
                            uint32_t* d_collision, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_links, dim3(par_grid(n)), dim3(kParBlock), 0, s, d_records, d_hash_s, d_idx_s, n, d_prev, d_collision);
    return hipGetLastError();
}

hipError_t launch_par_live(const TableView& t, const void* d_records, int32_t* d_prev, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_live, dim3(par_grid(n)), dim3(kParBlock), 0, s, t, d_records, d_prev, n);
    return hipGetLastError();
}

hipError_t launch_par_cuts(const int32_t* d_prev, uint64_t n, uint32_t max_entries, uint32_t live0, uint32_t* d_cuts, uint32_t max_cuts,
                           uint32_t* d_n_cuts, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_cuts, dim3(1), dim3(kCutBlock), 0, s, d_prev, n, max_entries, live0, d_cuts, max_cuts, d_n_cuts);
    return hipGetLastError();
}

hipError_t launch_par_tag_copy(const void* d_records, uint64_t first, uint64_t m, const uint32_t* d_cuts, uint32_t e0, uint32_t n_ep,
                               void* d_dst, hipStream_t s) {
    if (m == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_tag_copy, dim3(par_grid(m)), dim3(kParBlock), 0, s, d_records, first, m, d_cuts, e0, n_ep, d_dst);
    return hipGetLastError();
}

hipError_t launch_par_regroup(const void* d_evicted, uint64_t n_flows, uint32_t max_entries, uint32_t n_ep, void* d_out, uint32_t* d_cnt,
                              uint32_t* d_bad, hipStream_t s) {
    if (n_flows == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_regroup, dim3(par_grid(n_flows, kParBlock * kRegroupPer)), dim3(kParBlock), 0, s, d_evicted, n_flows, max_entries, n_ep, d_out, d_cnt, d_bad);
    return hipGetLastError();
}

}  // namespace nfagg
