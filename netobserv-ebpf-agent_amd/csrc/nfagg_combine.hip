// nfagg_combine.hip — flow PARTIALS: exporting the live slots of a table grouped by the shard that owns them, and merging
// such partials into the owner's table. This is the tick-time exchange of the local-fold mode (SURVEY.md §8(e) "hot-key
// replication with commutative partials merged at tick ... order-dependent fields need the seq tags"), used by the
// one-process group (nfagg_group.inc, NFAGG_GROUP_LOCAL_FOLD) and, through nfagg_partials_export_device /
// nfagg_partials_merge_device / nfagg_evict_owned_device, by one-process-per-GPU ranks (bench.py --gpus N).
//
// In that mode every GPU folds whatever arrives at it — no per-record routing, a flow may live on several GPUs at once —
// with sequence numbers that are global to the job. A slot (128-byte hot line + 64-byte cold half line, nfagg_internal.h)
// is then exactly a mergeable partial of its flow: sums, ORs, maxima, and sequence-tagged words whose larger tag wins. At
// the tick:
//   export   k_export_count / k_export_scatter   the live slots as 192-byte partials, GROUPED BY OWNER (nfagg_shard_of):
//                             segment o holds the flows shard o owns; the exporter's own flows stay where they are. Only
//                             segment o crosses the link to o: 192 B per (flow, GPU) / N instead of the whole export
//                             (round 2 sent every export whole to every owner).
//   merge    k_merge_raw      (owner) find or claim the flow's slot; bytes/packets add, flags OR, end / ~start /
//                             last-non-zero tags / first-record tag (with if_index_first_seen) / first-non-zero MAC words:
//                             atomic max
//            k_merge_identity (after it) the partial whose first-record tag won hands over its twelve plain identity dwords
//   evict    k_evict<FILTER>  with the shard filter: the flows the shard owns; what it folded of other shards' flows
//                             expires with the epoch (epoch tags: nothing is cleared).
// xGMI carries 192 bytes per (flow, GPU) instead of 144 bytes per record, and a hot flow is folded where its records
// arrive — by all GPUs — instead of by the one that owns it. Exactness is that of any other partial merge:
// every operator is associative and commutative once order is carried by the tags (pkg/model/flow_content.go:28-61,
// pkg/flow/account.go:95).
#include "nfagg_dedup.h"

namespace nfagg {

struct alignas(64) RawPartial {  // NFAGG_PARTIAL_BYTES: a slot's hot line (tag word unused) + its cold half line
    uint4 hot[8];
    uint4 cold[4];
};
static_assert(sizeof(RawPartial) == 192, "partial");

// Kernel-dedup mode: the tables of local-fold ranks are keyed by the SUB-FLOW (flow key, interface) and hold nothing that depends
// on the flow's first interface (nfagg_dedup.h), so a slot is a mergeable partial here too. It travels with the eight aux words a
// sub-flow uses; the sixth key word (the interface) rides in the hot line's `end` word. The OWNER is the owner of the FLOW.
struct alignas(64) RawPartialSub {  // kPartialBytesDedup
    uint4 hot[8];
    uint4 cold[4];
    uint64_t aux[8];             // SlotAux words 7..14: endl_lo, endl_hi, ssl_first, ssl_max | ssl_minv << 32, cs_tag, ks_tag, dir[0][0..1]
};
static_assert(sizeof(RawPartialSub) == kPartialBytesDedup, "sub-flow partial");
static_assert(offsetof(SlotAux, endl_lo) == 56 && offsetof(SlotAux, dir) == 104, "the aux words a sub-flow partial carries");

// Is the slot whose hot line is in hq[] a flow of this epoch (careful-path leftovers are not), and who owns it?
NF_DEV bool partial_owner(const uint4 hq[8], uint64_t seq_limit, uint32_t n_shards, uint32_t& owner, uint64_t& hash) {
    const uint32_t first_inv = hq[6].w;                          // id0 = word 13: its tag half
    if (first_inv == 0 || (uint64_t)(~first_inv) >= seq_limit) return false;
    const uint64_t w[5] = {(uint64_t)hq[0].z | ((uint64_t)hq[0].w << 32), (uint64_t)hq[1].x | ((uint64_t)hq[1].y << 32),
                           (uint64_t)hq[1].z | ((uint64_t)hq[1].w << 32), (uint64_t)hq[2].x | ((uint64_t)hq[2].y << 32),
                           (uint64_t)hq[2].z | ((uint64_t)hq[2].w << 32)};
    hash = key_hash(w);
    owner = shard_of_hash(hash, n_shards);
    return true;
}

// counts[o] += live flows of this table that shard o owns (self_shard's stay: not counted)
__global__ __launch_bounds__(256) void k_export_count(TableView t, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards,
                                                      uint32_t self_shard, unsigned long long* __restrict__ counts) {
    __shared__ unsigned int hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_live; i += stride) {
        const uint4* L = reinterpret_cast<const uint4*>(&t.hot[t.live_list[i]]);
        uint4 hq[8];
        hq[0] = L[0]; hq[1] = L[1]; hq[2] = L[2]; hq[6] = L[6];
        uint32_t owner; uint64_t hash;
        if (partial_owner(hq, seq_limit, n_shards, owner, hash) && owner != self_shard) atomicAdd(&hist[owner], 1u);
    }
    __syncthreads();
    if (threadIdx.x < n_shards && hist[threadIdx.x]) aadd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// out[cursor[o]++] = the slot, for every live flow owned by o != self_shard. A wave takes 64 slots; the lanes of one owner
// reserve their positions with ONE atomic (ballot + rank). Order inside a segment is unspecified (merging commutes).
template <typename P>
__global__ __launch_bounds__(256) void k_export_scatter(TableView t, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards,
                                                        uint32_t self_shard, unsigned long long* __restrict__ cursor,
                                                        P* __restrict__ out) {
    constexpr bool SUB = sizeof(P) == sizeof(RawPartialSub);
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave * 64; base < n_live; base += n_waves * 64) {
        const uint64_t i = base + lane;
        bool ok = i < n_live;
        const uint32_t idx = ok ? t.live_list[i] : 0u;
        uint4 hq[8], cq[4];
        uint32_t owner = 0xffffffffu;
        if (ok) {
            const uint4* L = reinterpret_cast<const uint4*>(&t.hot[idx]);
#pragma unroll
            for (int k = 0; k < 8; k++) hq[k] = L[k];
            uint64_t hash;
            ok = partial_owner(hq, seq_limit, n_shards, owner, hash) && owner != self_shard;
        }
        if (ok) {
            const uint4* Cc = reinterpret_cast<const uint4*>(&t.cold[idx]);
#pragma unroll
            for (int k = 0; k < 4; k++) cq[k] = Cc[k];
        }
        unsigned long long pos = 0, todo = __ballot(ok);
        while (todo) {                                           // wave-uniform: one trip per distinct owner among the 64 slots
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t o = (uint32_t)__shfl((int)owner, leader);
            const bool mine = ok && owner == o;
            const unsigned long long m = __ballot(mine);
            unsigned long long b = 0;
            if (lane == leader) b = aadd(&cursor[o], (unsigned long long)__popcll(m));
            b = __shfl(b, leader);
            if (mine) pos = b + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
            todo &= ~m;
        }
        if (ok) {
            P* p = out + pos;
            hq[0].x = 0; hq[0].y = 0;                            // the tag word means nothing outside its table
#pragma unroll
            for (int k = 0; k < 8; k++) p->hot[k] = hq[k];
#pragma unroll
            for (int k = 0; k < 4; k++) p->cold[k] = cq[k];
            if constexpr (SUB) {
                const uint64_t* aw = reinterpret_cast<const uint64_t*>(&t.aux[idx]) + 7;
                uint64_t* dst = reinterpret_cast<uint64_t*>(p) + 24;
#pragma unroll
                for (int k = 0; k < 8; k++) dst[k] = aw[k];
            }
        }
    }
}

NF_DEV uint64_t q2(const uint4& v, int hi) { return hi ? ((uint64_t)v.z | ((uint64_t)v.w << 32)) : ((uint64_t)v.x | ((uint64_t)v.y << 32)); }

// Merge partials (all owned by t.shard_id of t.n_shards; a partial of another shard raises error 7) into the table.
__global__ __launch_bounds__(256) void k_merge_raw(TableView t, const RawPartial* __restrict__ raw, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 hq[8];
#pragma unroll
        for (int k = 0; k < 8; k++) hq[k] = raw[i].hot[k];
        uint32_t owner; uint64_t hash;
        if (!partial_owner(hq, ~0ull, t.n_shards, owner, hash)) continue;
        if (owner != t.shard_id) { atomicExch(&t.ctr->error, 7u); continue; }
        const uint64_t w[5] = {q2(hq[0], 1), q2(hq[1], 0), q2(hq[1], 1), q2(hq[2], 0), q2(hq[2], 1)};
        const uint32_t idx = find_or_claim(t, w, hash);
        if (idx == kNoSlot) continue;                            // claim refused: `aborted` is raised, the caller reports it
        SlotHot* H = &t.hot[idx];
        SlotCold* C = &t.cold[idx];
        const uint64_t bytes = q2(hq[3], 0), end = q2(hq[3], 1), start_inv = q2(hq[4], 0), eth_tag = q2(hq[5], 0),
                       dscp_tag = q2(hq[5], 1), samp_tag = q2(hq[6], 0), id0 = q2(hq[6], 1), smac_lo = q2(hq[7], 0), dmac_lo = q2(hq[7], 1);
        const uint32_t packets = hq[4].z, flags = hq[4].w;
        if (bytes) aadd(&H->bytes, bytes);
        if (packets) aadd(&H->packets, packets);
        if (flags) aor(&H->flags, flags);
        if (end) amax(&H->end, end);
        if (start_inv) amax(&H->start_inv, start_inv);
        if (eth_tag) amax(&H->eth_tag, eth_tag);
        if (dscp_tag) amax(&H->dscp_tag, dscp_tag);
        if (samp_tag) amax(&H->samp_tag, samp_tag);
        amax(&H->id0, id0);                                      // earliest first record wins, its if_index_first_seen with it
        if (smac_lo || dmac_lo) {
            const uint4 c0 = raw[i].cold[0];
            if (smac_lo) { amax(&H->smac_lo, smac_lo); amax(&C->smac_hi, q2(c0, 0)); }
            if (dmac_lo) { amax(&H->dmac_lo, dmac_lo); amax(&C->dmac_hi, q2(c0, 1)); }
        }
    }
}

__global__ __launch_bounds__(256) void k_merge_identity(TableView t, const RawPartial* __restrict__ raw, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 hq[8];
        hq[0] = raw[i].hot[0]; hq[1] = raw[i].hot[1]; hq[2] = raw[i].hot[2]; hq[6] = raw[i].hot[6];
        uint32_t owner; uint64_t hash;
        if (!partial_owner(hq, ~0ull, t.n_shards, owner, hash) || owner != t.shard_id) continue;
        const uint64_t w[5] = {q2(hq[0], 1), q2(hq[1], 0), q2(hq[1], 1), q2(hq[2], 0), q2(hq[2], 1)};
        const uint32_t idx = find_or_claim(t, w, hash);          // exists since k_merge_raw: this only walks the probe sequence
        if (idx == kNoSlot) continue;
        // sequence numbers are unique in the job: equal tags = the same record = this partial holds the flow's first record
        if ((uint32_t)(t.hot[idx].id0 >> 32) != hq[6].w) continue;
        uint4* dst = reinterpret_cast<uint4*>(&t.cold[idx]);
        dst[1] = raw[i].cold[1]; dst[2] = raw[i].cold[2]; dst[3] = raw[i].cold[3];
    }
}

// ---- the same for sub-flow partials (kernel-dedup mode). Everything that combines: sums, ORs, "last value" tags, the TLS words,
// the two earliest directions (top-2 over sequence-tagged words), and the tag of the sub-flow's first record.
__global__ __launch_bounds__(256) void k_merge_sub(TableView t, const RawPartialSub* __restrict__ raw, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 hq[8];
#pragma unroll
        for (int k = 0; k < 8; k++) hq[k] = raw[i].hot[k];
        uint32_t owner; uint64_t hash;
        if (!partial_owner(hq, ~0ull, t.n_shards, owner, hash)) continue;
        if (owner != t.shard_id) { atomicExch(&t.ctr->error, 7u); continue; }
        const uint64_t w[5] = {q2(hq[0], 1), q2(hq[1], 0), q2(hq[1], 1), q2(hq[2], 0), q2(hq[2], 1)};
        const uint64_t kx = q2(hq[3], 1);                        // the sixth key word: 1 << 32 | interface
        if ((kx >> 32) != 1ull) { atomicExch(&t.ctr->error, 7u); continue; }
        const uint32_t ifx = (uint32_t)kx;
        const uint32_t idx = find_or_claim(t, w, sub_hash(t, hash, ifx), nullptr, nullptr, nullptr, kx);
        if (idx == kNoSlot) continue;                            // claim refused: `aborted` is raised, the caller reports it
        SlotHot* H = &t.hot[idx];
        SlotAux* A = &t.aux[idx];
        const uint64_t bytes = q2(hq[3], 0), dscp_tag = q2(hq[5], 1), samp_tag = q2(hq[6], 0), id0 = q2(hq[6], 1);
        const uint32_t packets = hq[4].z, flags = hq[4].w;
        if (bytes) aadd(&H->bytes, bytes);
        if (packets) aadd(&H->packets, packets);
        if (flags) aor(&H->flags, flags);
        if (dscp_tag) amax(&H->dscp_tag, dscp_tag);
        if (samp_tag) amax(&H->samp_tag, samp_tag);
        amax(&H->id0, id0);                                      // the sub-flow's earliest record wins (its data: k_merge_sub_identity)
        uint64_t a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = raw[i].aux[k];
        if (a[0]) amax(&A->endl_lo, a[0]);
        if (a[1]) amax(&A->endl_hi, a[1]);
        if (a[2]) {
            amax(&A->ssl_first, a[2]);
            atomicMax(&A->ssl_max, (uint32_t)a[3]);
            atomicMax(&A->ssl_minv, (uint32_t)(a[3] >> 32));
        }
        if (a[4]) amax(&A->cs_tag, a[4]);
        if (a[5]) amax(&A->ks_tag, a[5]);
        if (a[6]) topk_insert<2, 0xffull>(t, A->dir[0], a[6]);
        if (a[7]) topk_insert<2, 0xffull>(t, A->dir[0], a[7]);
    }
}

// the partial whose first-record tag won hands over what the first record stores whole (account.go:95): raw start and
// eth_protocol, the MACs, the identity dwords
__global__ __launch_bounds__(256) void k_merge_sub_identity(TableView t, const RawPartialSub* __restrict__ raw, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 hq[8];
#pragma unroll
        for (int k = 0; k < 8; k++) hq[k] = raw[i].hot[k];
        uint32_t owner; uint64_t hash;
        if (!partial_owner(hq, ~0ull, t.n_shards, owner, hash) || owner != t.shard_id) continue;
        const uint64_t w[5] = {q2(hq[0], 1), q2(hq[1], 0), q2(hq[1], 1), q2(hq[2], 0), q2(hq[2], 1)};
        const uint64_t kx = q2(hq[3], 1);
        if ((kx >> 32) != 1ull) continue;
        const uint32_t idx = find_or_claim(t, w, sub_hash(t, hash, (uint32_t)kx), nullptr, nullptr, nullptr, kx);   // exists since k_merge_sub
        if (idx == kNoSlot) continue;
        SlotHot* H = &t.hot[idx];
        if ((uint32_t)(H->id0 >> 32) != hq[6].w) continue;       // sequence numbers are unique in the job: equal tags = the same record
        H->start_inv = q2(hq[4], 0);
        H->eth_tag = q2(hq[5], 0);
        H->smac_lo = q2(hq[7], 0);
        H->dmac_lo = q2(hq[7], 1);
        uint4* dst = reinterpret_cast<uint4*>(&t.cold[idx]);
#pragma unroll
        for (int k = 0; k < 4; k++) dst[k] = raw[i].cold[k];
    }
}

// flows of this table that this shard owns (what k_evict with the shard filter will write)
__global__ __launch_bounds__(256) void k_count_owned(TableView t, uint64_t n_live, uint64_t seq_limit, unsigned long long* count) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_live; i += stride) {
        const uint4* L = reinterpret_cast<const uint4*>(&t.hot[t.live_list[i]]);
        uint4 hq[8];
        hq[0] = L[0]; hq[1] = L[1]; hq[2] = L[2]; hq[6] = L[6];
        uint32_t owner; uint64_t hash;
        if (partial_owner(hq, seq_limit, t.n_shards, owner, hash) && owner == t.shard_id) mine++;
    }
    if (mine) aadd(count, mine);
}

static inline int grid_for(uint64_t n) { uint64_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096; return (int)g; }

// d_counts: 64 words, zeroed here. Asynchronous.
hipError_t launch_export_count(const TableView& t, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards, uint32_t self_shard,
                               unsigned long long* d_counts, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_counts, 0, 64 * sizeof(unsigned long long), s);
    if (e != hipSuccess || n_live == 0) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_export_count, dim3(grid_for(n_live)), dim3(256), 0, s, t, n_live, seq_limit, n_shards, self_shard, d_counts);
    return hipGetLastError();
}

// d_cursor[o] = first position of segment o on entry (advanced by the kernel)
hipError_t launch_export_scatter(const TableView& t, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards, uint32_t self_shard,
                                 unsigned long long* d_cursor, void* d_out, hipStream_t s) {
    if (n_live == 0) return hipSuccess;
    (void)hipGetLastError();
    if (t.subflow) hipLaunchKernelGGL(k_export_scatter<RawPartialSub>, dim3(grid_for(n_live)), dim3(256), 0, s, t, n_live, seq_limit, n_shards, self_shard,
                                      d_cursor, reinterpret_cast<RawPartialSub*>(d_out));
    else hipLaunchKernelGGL(k_export_scatter<RawPartial>, dim3(grid_for(n_live)), dim3(256), 0, s, t, n_live, seq_limit, n_shards, self_shard, d_cursor,
                            reinterpret_cast<RawPartial*>(d_out));
    return hipGetLastError();
}

hipError_t launch_merge_raw(const TableView& t, const void* d_partials, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    if (t.subflow) {
        if (!t.aux) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_merge_sub, dim3(grid_for(n)), dim3(256), 0, s, t, reinterpret_cast<const RawPartialSub*>(d_partials), n);
        const hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return e1;
        hipLaunchKernelGGL(k_merge_sub_identity, dim3(grid_for(n)), dim3(256), 0, s, t, reinterpret_cast<const RawPartialSub*>(d_partials), n);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_merge_raw, dim3(grid_for(n)), dim3(256), 0, s, t, reinterpret_cast<const RawPartial*>(d_partials), n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_merge_identity, dim3(grid_for(n)), dim3(256), 0, s, t, reinterpret_cast<const RawPartial*>(d_partials), n);
    return hipGetLastError();
}

hipError_t launch_count_owned(const TableView& t, uint64_t n_live, uint64_t seq_limit, unsigned long long* d_count, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || n_live == 0) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_count_owned, dim3(grid_for(n_live)), dim3(256), 0, s, t, n_live, seq_limit, d_count);
    return hipGetLastError();
}

}  // namespace nfagg
