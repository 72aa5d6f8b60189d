// nfagg_partition.hip — stable partition of a record batch by flow-key shard (the device-side router of the multi-GPU
// group, nfagg_group.inc; SURVEY.md §8(e): "a GPU partition kernel + P2P scatter").
//
// Records shard by nfagg_shard_of(key, n_shards) (nfagg_hash.h; the Go agent is ONE process in front of N GPUs,
// pkg/agent/agent.go:387-442). The partition is STABLE — within a bucket the records keep their arrival order — because the
// fold of a flow depends on the order of its records (flow_content.go:45-59, account.go:95) and every record of a flow
// lands in one bucket. Three launches:
//   k_part_count    tile t (kTile consecutive records): histogram of the tile by shard                 -> hist[shard][tile]
//   k_part_scan     exclusive scan of hist in (shard, tile) order                                      -> offs[shard][tile], base[shard]
//   k_part_scatter  tile t again: record i goes to out[offs[shard][t] + its rank among the tile's records of that shard]
// The rank inside a tile is computed round by round (one record per lane, 256 consecutive records per round): a ballot per
// shard inside the wave, an LDS prefix over the four waves, a running per-shard count across rounds. HBM-bound: the batch
// is read twice (key only needed, but 40 of every 144 bytes touch every line) and written once: 432 B per record.
#include "nfagg_device.h"

namespace nfagg {

constexpr int kPartBlock = 256;
constexpr int kPartRounds = 16;
constexpr int kPartTile = kPartBlock * kPartRounds;   // 4096 records per tile
constexpr int kMaxShards = 64;

NF_DEV uint32_t record_shard(const void* recs, uint64_t i, uint32_t n_shards) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + i * kRecordBytes);
    const uint4 a = p[0], b = p[1], c = p[2];
    uint64_t w[5];
    w[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); w[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    w[2] = (uint64_t)b.x | ((uint64_t)b.y << 32); w[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
    w[4] = ((uint64_t)c.x | ((uint64_t)c.y << 32)) & 0x00FFFFFFFFFFFFFFull;      // key byte 39: Go's blank field, not part of the key
    return shard_of_hash(key_hash(w), n_shards);
}

__global__ __launch_bounds__(kPartBlock) void k_part_count(const void* __restrict__ recs, uint64_t n, uint32_t n_shards,
                                                           uint32_t n_tiles, uint32_t* __restrict__ hist) {
    __shared__ uint32_t cnt[kMaxShards];
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (threadIdx.x < kMaxShards) cnt[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t t0 = (uint64_t)tile * kPartTile;
#pragma unroll 4
        for (int r = 0; r < kPartRounds; r++) {
            const uint64_t i = t0 + (uint64_t)r * kPartBlock + threadIdx.x;
            if (i < n) atomicAdd(&cnt[record_shard(recs, i, n_shards)], 1u);
        }
        __syncthreads();
        if (threadIdx.x < n_shards) hist[(uint64_t)threadIdx.x * n_tiles + tile] = cnt[threadIdx.x];
        __syncthreads();
    }
}

// One workgroup per shard: exclusive scan of its row of the histogram; the row total goes to count[shard]. The bucket bases
// (exclusive scan of the counts over the shards) are added by k_part_scatter, which reads all n_shards counts anyway.
__global__ __launch_bounds__(1024) void k_part_scan(uint32_t* __restrict__ hist, uint32_t n_tiles, uint64_t* __restrict__ count) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry;
    uint32_t* row = hist + (uint64_t)blockIdx.x * n_tiles;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < n_tiles ? row[t] : 0u;
        uint64_t x = v;                                               // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d); if ((int)(threadIdx.x & 63) >= d) x += y; }
        if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = x;
        __syncthreads();
        uint64_t before = carry;
        for (int wv = 0; wv < (int)(threadIdx.x >> 6); wv++) before += wave_sum[wv];
        if (t < n_tiles) row[t] = (uint32_t)(before + x - v);        // offsets inside one bucket fit 32 bits (n < 2^32)
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) count[blockIdx.x] = carry;
}

__global__ __launch_bounds__(kPartBlock) void k_part_scatter(const void* __restrict__ recs, uint64_t n, uint32_t n_shards,
                                                             uint32_t n_tiles, const uint32_t* __restrict__ offs,
                                                             const uint64_t* __restrict__ count, void* __restrict__ out,
                                                             uint32_t* __restrict__ out_orig) {
    __shared__ uint64_t base[kMaxShards];            // first record of every bucket in `out`
    __shared__ uint32_t run[kMaxShards];             // records of the tile already placed, per shard
    __shared__ uint32_t wcnt[4][kMaxShards];         // this round: records per shard in each wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { uint64_t b = 0; for (uint32_t s = 0; s < n_shards; s++) { base[s] = b; b += count[s]; } }
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < n_shards) run[threadIdx.x] = offs[(uint64_t)threadIdx.x * n_tiles + tile];
        const uint64_t t0 = (uint64_t)tile * kPartTile;
        for (int r = 0; r < kPartRounds; r++) {
            const uint64_t i = t0 + (uint64_t)r * kPartBlock + threadIdx.x;
            const bool valid = i < n;
            const uint32_t s = valid ? record_shard(recs, i, n_shards) : 0xffffffffu;
            // rank among the lanes of this wave with the same shard (lower lanes first: arrival order)
            uint32_t rank = 0, mine = 0;
            for (uint32_t q = 0; q < n_shards; q++) {
                const unsigned long long m = __ballot(s == q);
                if (s == q) { rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); mine = (uint32_t)__popcll(m); }
                if (lane == 0) wcnt[wv][q] = (uint32_t)__popcll(m);
            }
            __syncthreads();
            if (valid) {
                uint32_t before = run[s];
                for (int v = 0; v < wv; v++) before += wcnt[v][s];
                const uint64_t dst = base[s] + before + rank;
                const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + i * kRecordBytes);
                uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + dst * kRecordBytes);
#pragma unroll
                for (int k = 0; k < 9; k++) o[k] = src[k];
                out_orig[dst] = (uint32_t)i;
            }
            (void)mine;
            __syncthreads();
            if (threadIdx.x < n_shards) run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
            __syncthreads();
        }
    }
}

// cnt[s] = number of entries of bucket s whose original index is below m (buckets are ascending in the original index).
__global__ void k_part_prefix_counts(const uint32_t* __restrict__ orig, const uint64_t* __restrict__ count, uint32_t n_shards,
                                     uint64_t m, uint64_t* __restrict__ cnt) {
    const uint32_t s = threadIdx.x;
    if (s >= n_shards) return;
    uint64_t b = 0;
    for (uint32_t q = 0; q < s; q++) b += count[q];
    uint64_t lo = 0, hi = count[s];
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)orig[b + mid] < m) lo = mid + 1; else hi = mid; }
    cnt[s] = lo;
}

size_t partition_scratch_bytes(uint64_t n, uint32_t n_shards) {
    const uint64_t n_tiles = (n + kPartTile - 1) / kPartTile;
    return (size_t)(n_tiles * n_shards * sizeof(uint32_t) + 256);
}

// d_count: n_shards x uint64 (DEVICE). d_out: n records, buckets back to back in shard order; d_orig: n x uint32.
hipError_t launch_partition(const void* d_records, uint64_t n, uint32_t n_shards, void* d_out, uint32_t* d_orig,
                            uint64_t* d_count, void* d_scratch, hipStream_t s) {
    if (n_shards == 0 || n_shards > (uint32_t)kMaxShards || n >= (1ull << 32)) return hipErrorInvalidValue;
    if (n == 0) return hipMemsetAsync(d_count, 0, n_shards * sizeof(uint64_t), s);
    const uint32_t n_tiles = (uint32_t)((n + kPartTile - 1) / kPartTile);
    uint32_t* hist = reinterpret_cast<uint32_t*>(d_scratch);
    const unsigned grid = n_tiles < 2048u ? n_tiles : 2048u;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_part_count, dim3(grid), dim3(kPartBlock), 0, s, d_records, n, n_shards, n_tiles, hist);
    hipLaunchKernelGGL(k_part_scan, dim3(n_shards), dim3(1024), 0, s, hist, n_tiles, d_count);
    hipLaunchKernelGGL(k_part_scatter, dim3(grid), dim3(kPartBlock), 0, s, d_records, n, n_shards, n_tiles, hist, d_count, d_out, d_orig);
    return hipGetLastError();
}

hipError_t launch_partition_prefix_counts(const uint32_t* d_orig, const uint64_t* d_count, uint32_t n_shards, uint64_t m,
                                          uint64_t* d_cnt, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_part_prefix_counts, dim3(1), dim3(kMaxShards), 0, s, d_orig, d_count, n_shards, m, d_cnt);
    return hipGetLastError();
}

}  // namespace nfagg
