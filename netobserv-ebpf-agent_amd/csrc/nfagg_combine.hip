// nfagg_combine.hip — merging RAW SLOTS of another table into this one: the tick-time exchange of the multi-GPU group's
// local-fold mode (nfagg_group.inc, NFAGG_GROUP_LOCAL_FOLD; SURVEY.md §8(e) "hot-key replication with commutative partials
// merged at tick ... order-dependent fields need the seq tags").
//
// In that mode every member folds whatever arrives at it — no per-record routing, a flow may live on several GPUs at once —
// with sequence numbers that are global to the group. A slot (128-byte hot line + 64-byte cold half line, nfagg_internal.h)
// is then exactly a mergeable partial of its flow: sums, ORs, maxima, and sequence-tagged words whose larger tag wins. At
// the tick each member's live slots are exported raw (k_snapshot's format) and the member that OWNS a flow
// (nfagg_shard_of) merges the others' slots of it into its own table:
//   phase 1  k_merge_raw      find or claim the flow's slot; bytes/packets add, flags OR, end / ~start / last-non-zero tags /
//                             first-record tag (with if_index_first_seen) / first-non-zero MAC words: atomic max
//   phase 2  k_merge_identity (after every phase-1 launch of the owner) the raw slot whose first-record tag won hands over
//                             its twelve plain identity dwords
// and evicts the flows it owns (k_evict with the shard filter); what it folded of other members' flows simply expires with
// the epoch. xGMI carries 192 bytes per (flow, member) instead of 144 bytes per record, and a hot flow is folded where its
// records arrive — by all GPUs — instead of by the one that owns it. Exactness is that of any other partial merge (§2 of
// DESIGN.md): every operator is associative and commutative once order is carried by the tags.
#include "nfagg_device.h"

namespace nfagg {

struct RawView {                 // k_snapshot's layout: n hot lines, then n cold half lines
    const SlotHot* hot;
    const SlotCold* cold;
    uint64_t n;
    uint64_t seq_limit;          // slots whose first record lies at or after it are not part of the epoch (careful-path leftovers)
};

NF_DEV bool raw_owned(const TableView& t, const SlotHot& h, uint64_t seq_limit, uint64_t& hash) {
    const uint32_t first_inv = (uint32_t)(h.id0 >> 32);
    if (first_inv == 0 || (uint64_t)(~first_inv) >= seq_limit) return false;
    hash = key_hash(h.key);
    return shard_of_hash(hash, t.n_shards) == t.shard_id;
}

__global__ __launch_bounds__(256) void k_merge_raw(TableView t, RawView r) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n; i += stride) {
        const SlotHot h = r.hot[i];
        uint64_t hash;
        if (!raw_owned(t, h, r.seq_limit, hash)) continue;
        const uint32_t idx = find_or_claim(t, h.key, hash);
        if (idx == kNoSlot) continue;                            // claim refused: `aborted` is raised, the caller reports it
        SlotHot* H = &t.hot[idx];
        SlotCold* C = &t.cold[idx];
        if (h.bytes) aadd(&H->bytes, h.bytes);
        if (h.packets) aadd(&H->packets, h.packets);
        if (h.flags) aor(&H->flags, h.flags);
        if (h.end) amax(&H->end, h.end);
        if (h.start_inv) amax(&H->start_inv, h.start_inv);
        if (h.eth_tag) amax(&H->eth_tag, h.eth_tag);
        if (h.dscp_tag) amax(&H->dscp_tag, h.dscp_tag);
        if (h.samp_tag) amax(&H->samp_tag, h.samp_tag);
        amax(&H->id0, h.id0);                                    // earliest first record wins, its if_index_first_seen with it
        if (h.smac_lo) { amax(&H->smac_lo, h.smac_lo); amax(&C->smac_hi, r.cold[i].smac_hi); }
        if (h.dmac_lo) { amax(&H->dmac_lo, h.dmac_lo); amax(&C->dmac_hi, r.cold[i].dmac_hi); }
    }
}

__global__ __launch_bounds__(256) void k_merge_identity(TableView t, RawView r) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n; i += stride) {
        const SlotHot h = r.hot[i];
        uint64_t hash;
        if (!raw_owned(t, h, r.seq_limit, hash)) continue;
        const uint32_t idx = find_or_claim(t, h.key, hash);      // exists since phase 1: this only walks the probe sequence
        if (idx == kNoSlot) continue;
        // sequence numbers are unique in the group: equal tags = the same record = this raw slot holds the flow's first record
        if ((uint32_t)(t.hot[idx].id0 >> 32) != (uint32_t)(h.id0 >> 32)) continue;
        const uint4* src = reinterpret_cast<const uint4*>(&r.cold[i]);
        uint4* dst = reinterpret_cast<uint4*>(&t.cold[idx]);
        dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    }
}

// flows of this table that this shard owns (what k_evict with the shard filter will write)
__global__ __launch_bounds__(256) void k_count_owned(TableView t, uint64_t n_live, uint64_t seq_limit, unsigned long long* count) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_live; i += stride) {
        const SlotHot& h = t.hot[t.live_list[i]];
        const uint32_t first_inv = (uint32_t)(h.id0 >> 32);
        if (first_inv == 0 || (uint64_t)(~first_inv) >= seq_limit) continue;
        uint64_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = h.key[k];
        if (shard_of_hash(key_hash(w), t.n_shards) == t.shard_id) mine++;
    }
    if (mine) aadd(count, mine);
}

static inline int grid_for(uint64_t n) { uint64_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096; return (int)g; }

hipError_t launch_merge_raw(const TableView& t, const void* d_raw, uint64_t n, uint64_t seq_limit, bool identity_phase, hipStream_t s) {
    if (n == 0) return hipSuccess;
    RawView r;
    r.hot = reinterpret_cast<const SlotHot*>(d_raw);
    r.cold = reinterpret_cast<const SlotCold*>(reinterpret_cast<const char*>(d_raw) + n * sizeof(SlotHot));
    r.n = n; r.seq_limit = seq_limit;
    (void)hipGetLastError();
    if (identity_phase) hipLaunchKernelGGL(k_merge_identity, dim3(grid_for(n)), dim3(256), 0, s, t, r);
    else hipLaunchKernelGGL(k_merge_raw, dim3(grid_for(n)), dim3(256), 0, s, t, r);
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
