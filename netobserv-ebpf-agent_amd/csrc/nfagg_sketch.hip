// nfagg_sketch.hip — per-endpoint Count-Min and HyperLogLog sketches.
// New functionality: the reference has no sketch; the spec
// is in DESIGN.md §sketches and restated independently in oracle/.
//   CM : depth rows x 2^log2w uint64 counters; row r index =
//        (ha + r*hb) >> (64-log2w), ha = ip_hash(ip,0), hb = ip_hash(ip,1)|1;
//        adds metrics.bytes. One sketch keyed by src IP, one by dst IP.
//   HLL: m = 2^p registers; h = ip_hash(ip,2); idx = h >> (64-p);
//        rho = clz((h<<p) | 1<<(p-1)) + 1; register = max.
#include <hipcub/hipcub.hpp>
#include "nfagg_device.h"

namespace nfagg {

// ---- wave-wide sum of a 64-bit value with DPP (64 lanes = four rows of 16): an inclusive scan inside each row
// (row_shr:1,2,4,8, zeroes shifted in), then row 0's total into row 1 and row 2's into row 3 (row_bcast:15), then the total of
// rows 0-1 into rows 2-3 (row_bcast:31): lane 63 holds the sum. Every lane of the wave must be active.
template <int CTRL, int ROW_MASK>
NF_DEV uint64_t dpp_add_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, true);
    return v + ((uint64_t)lo | ((uint64_t)hi << 32));
}
NF_DEV uint64_t wave_sum_u64(uint64_t v) {
    v = dpp_add_u64<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add_u64<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add_u64<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add_u64<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add_u64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add_u64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// One side's sketch update for the wave's 64 records. A hot endpoint (configs[4]: 90 % of the records are one flow) would
// send ~58 lanes' Count-Min adds to the same 4 counters — and same-address atomics retire one every ~14 ns however many waves
// issue them. So (1) the lanes that carry the first pending lane's address are combined: their byte counts summed across the
// wave with DPP (Count-Min is linear, HyperLogLog idempotent); (2) the combined sum is not sent to the sketch at once but
// CARRIED by the wave (wave-uniform registers) as long as the following tiles' leaders bring the same address, and flushed
// when another address takes its place or the wave is done: the hot endpoint reaches its counters once per wave and kernel,
// not once per tile. The carried address is matched first, then up to three leader rounds, the rest of the lanes on their own.
struct Carried { uint64_t lo = 0, hi = 0, sum = 0; bool valid = false; };

NF_DEV void carried_flush(const SketchView& sk, int side, Carried& c) {
    if (c.valid && (threadIdx.x & 63) == 0) sketch_add_side(sk, side, c.lo, c.hi, c.sum);
    c.valid = false;
}

NF_DEV void sketch_side_wave(const SketchView& sk, int side, bool active, uint64_t lo, uint64_t hi, uint64_t bytes, Carried& c) {
    const int lane = threadIdx.x & 63;
    unsigned long long pending = __ballot(active), solo = 0;
    bool carried_seen = false;
    if (c.valid) {                                                             // the carried address first, wherever its lanes sit
        const bool in = ((pending >> lane) & 1ull) && lo == c.lo && hi == c.hi;
        const unsigned long long grp = __ballot(in);
        if (grp) { c.sum += wave_sum_u64(in ? bytes : 0ull); pending &= ~grp; carried_seen = true; }
    }
    for (int round = 0; round < 3 && pending; round++) {
        const int leader = __ffsll((long long)pending) - 1;
        const uint64_t llo = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)lo, leader) |
                             ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(lo >> 32), leader) << 32);
        const uint64_t lhi = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hi, leader) |
                             ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hi >> 32), leader) << 32);
        const bool in = ((pending >> lane) & 1ull) && lo == llo && hi == lhi;
        const unsigned long long grp = __ballot(in);
        pending &= ~grp;
        if (__popcll(grp) < 2) { solo |= grp; continue; }                      // alone: on its own below; look at the next lane
        const uint64_t total = wave_sum_u64(in ? bytes : 0ull);
        if (!carried_seen) {
            // the carried address (if any) did not show up in this tile and this one came with company: it takes its place
            carried_flush(sk, side, c);
            c.lo = llo; c.hi = lhi; c.sum = total; c.valid = true; carried_seen = true;
        } else if (lane == leader) {
            sketch_add_side(sk, side, lo, hi, total);
        }
    }
    solo |= pending;
    if ((solo >> lane) & 1ull) sketch_add_side(sk, side, lo, hi, bytes);
}

__global__ __launch_bounds__(256) void k_sketch_update(SketchView sk, TableView t, const void* __restrict__ recs, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    Carried cs, cd;
    // the loop bound is uniform for the wave: every lane stays in (the DPP reduction needs all 64), `active` gates the work
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint64_t i = base + threadIdx.x;
        bool active = i < n;
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + (active ? i : 0) * kRecordBytes);
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        uint64_t w[5];
        w[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); w[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
        w[2] = (uint64_t)b.x | ((uint64_t)b.y << 32); w[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
        w[4] = ((uint64_t)c.x | ((uint64_t)c.y << 32)) & 0x00ffffffffffffffull;
        if (active && t.n_shards > 1 && shard_of_hash(key_hash(w), t.n_shards) != t.shard_id) active = false;
        const uint64_t bytes = (uint64_t)d.z | ((uint64_t)d.w << 32);   // metrics.bytes @56
        sketch_side_wave(sk, 0, active, w[0], w[1], bytes, cs);
        sketch_side_wave(sk, 1, active, w[2], w[3], bytes, cd);
    }
    carried_flush(sk, 0, cs);
    carried_flush(sk, 1, cd);
}

__global__ __launch_bounds__(256) void k_hll_histogram(const uint8_t* __restrict__ regs, uint32_t p, uint32_t* __restrict__ hist) {
    __shared__ unsigned int sh[65];
    for (int k = threadIdx.x; k < 65; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    const uint32_t m = 1u << p;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        uint32_t v = regs[i];
        atomicAdd(&sh[v > 64 ? 64 : v], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 65; k += blockDim.x) hist[k] = sh[k];
}

// Heavy hitters: Count-Min stores no keys, so the candidates are the addresses that occur in a record batch (typically the
// one nfagg_evict just returned). est[i] = min over rows of the counter of record i's src (side 0) / dst (side 1) address.
__global__ __launch_bounds__(256) void k_cm_estimate(const uint64_t* __restrict__ cm, uint32_t depth, uint32_t log2w, int side,
                                                     const void* __restrict__ recs, uint64_t n, uint64_t* __restrict__ est, uint32_t* __restrict__ idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 a = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + i * kRecordBytes)[side];
    const uint64_t lo = (uint64_t)a.x | ((uint64_t)a.y << 32), hi = (uint64_t)a.z | ((uint64_t)a.w << 32);
    const uint64_t ha = ip_hash(lo, hi, 0), hb = ip_hash(lo, hi, 1) | 1ull;
    uint64_t best = ~0ull;
    for (uint32_t r = 0; r < depth; r++) {
        const uint64_t v = cm[((uint64_t)r << log2w) + cm_index(ha, hb, r, log2w)];
        best = v < best ? v : best;
    }
    est[i] = best;
    idx[i] = (uint32_t)i;
}

// rows[j] = {address of record idx[j], est[j]} for the first m entries of the sorted order
__global__ __launch_bounds__(256) void k_cm_gather(const void* __restrict__ recs, int side, const uint64_t* __restrict__ est_sorted,
                                                   const uint32_t* __restrict__ idx_sorted, uint64_t m, uint64_t* __restrict__ rows) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint4 a = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + (uint64_t)idx_sorted[j] * kRecordBytes)[side];
    rows[3 * j] = (uint64_t)a.x | ((uint64_t)a.y << 32);
    rows[3 * j + 1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    rows[3 * j + 2] = est_sorted[j];
}

hipError_t launch_cm_estimate(const uint64_t* d_cm, uint32_t depth, uint32_t log2w, int side, const void* d_records, uint64_t n,
                              uint64_t* d_est, uint32_t* d_idx, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_cm_estimate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_cm, depth, log2w, side, d_records, n, d_est, d_idx);
    return hipGetLastError();
}
hipError_t launch_cm_sort_desc(const uint64_t* d_est, uint64_t* d_est_sorted, const uint32_t* d_idx, uint32_t* d_idx_sorted, uint64_t n,
                               void* d_temp, size_t* temp_bytes, hipStream_t s) {
    return hipcub::DeviceRadixSort::SortPairsDescending(d_temp, *temp_bytes, d_est, d_est_sorted, d_idx, d_idx_sorted, (int)n, 0, 64, s);
}
hipError_t launch_cm_gather(const void* d_records, int side, const uint64_t* d_est_sorted, const uint32_t* d_idx_sorted, uint64_t m,
                            uint64_t* d_rows, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_cm_gather, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_records, side, d_est_sorted, d_idx_sorted, m, d_rows);
    return hipGetLastError();
}

hipError_t launch_sketch_update(const SketchView& sk, const TableView& t, const void* d_records, uint64_t n, hipStream_t s) {
    if (n == 0 || sk.flags == 0) return hipSuccess;
    uint64_t g = (n + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    (void)hipGetLastError(); hipLaunchKernelGGL(k_sketch_update, dim3((unsigned)g), dim3(256), 0, s, sk, t, d_records, n);
    return hipGetLastError();
}

hipError_t launch_hll_histogram(const uint8_t* d_regs, uint32_t p, uint32_t* d_hist65, hipStream_t s) {
    (void)hipGetLastError(); hipLaunchKernelGGL(k_hll_histogram, dim3(1), dim3(256), 0, s, d_regs, p, d_hist65);
    return hipGetLastError();
}

}  // namespace nfagg
