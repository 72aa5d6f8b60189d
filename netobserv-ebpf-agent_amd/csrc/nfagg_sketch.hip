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

__global__ __launch_bounds__(256) void k_sketch_update(SketchView sk, TableView t, const void* __restrict__ recs, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + i * kRecordBytes);
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        uint64_t w[5];
        w[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); w[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
        w[2] = (uint64_t)b.x | ((uint64_t)b.y << 32); w[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
        w[4] = ((uint64_t)c.x | ((uint64_t)c.y << 32)) & 0x00ffffffffffffffull;
        if (t.n_shards > 1 && shard_of_hash(key_hash(w), t.n_shards) != t.shard_id) continue;
        const uint64_t bytes = (uint64_t)d.z | ((uint64_t)d.w << 32);   // metrics.bytes @56
        sketch_add(sk, w, bytes);
    }
}

__global__ __launch_bounds__(256) void k_hll_histogram(const uint32_t* __restrict__ regs, uint32_t p, uint32_t* __restrict__ hist) {
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

__global__ __launch_bounds__(256) void k_hll_pack(const uint32_t* __restrict__ regs, uint32_t m, uint8_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) out[i] = (uint8_t)regs[i];
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

hipError_t launch_hll_histogram(const uint32_t* d_regs, uint32_t p, uint32_t* d_hist65, hipStream_t s) {
    (void)hipGetLastError(); hipLaunchKernelGGL(k_hll_histogram, dim3(1), dim3(256), 0, s, d_regs, p, d_hist65);
    return hipGetLastError();
}

hipError_t launch_hll_pack(const uint32_t* d_regs, uint32_t p, uint8_t* d_out, hipStream_t s) {
    const uint32_t m = 1u << p;
    (void)hipGetLastError(); hipLaunchKernelGGL(k_hll_pack, dim3((m + 255) / 256), dim3(256), 0, s, d_regs, m, d_out);
    return hipGetLastError();
}

}  // namespace nfagg
