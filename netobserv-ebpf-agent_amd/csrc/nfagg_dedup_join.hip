// nfagg_dedup_join.hip — the join that ends the epoch of a SUB-FLOW table (kernel-dedup mode of a local-fold rank; nfagg_dedup.h
// "Local fold across GPUs", DESIGN.md §7 a').
//
// Such a table is keyed by (flow key, if_index_first_seen) and every slot holds both roles of its interface, because which
// interface of a flow is the COUNTED one (bpf/flows.c:100-126: if_index == if_index_first_seen) is decided by the flow's earliest
// record anywhere in the job — known only when the sub-flows of all GPUs have met at the flow's owner (nfagg_combine.hip merged
// them slot by slot). Now every sub-flow of the table that this shard owns is one pre-folded run of records of ONE (flow,
// interface) — exactly what an exported cache entry is to the partition pass of nfagg_dedup_cached.hip — and goes through the
// same two phases into a flow-keyed table J of the ordinary kernel-dedup layout:
//   k_join_claim  find or claim the flow's slot in J; the sub-flow's first sequence number competes for "first record of the
//                 flow" (tagged max on id0, its interface with it: F) and for the seven earliest interfaces (dedup_claim);
//   k_join_merge  (F final: the claims are a previous kernel) update_existing_flow for the sub-flow as a whole (dedup_merge:
//                 counted when its interface is F, a side interface with its two earliest directions otherwise, ignored when its
//                 interface is 0 != F); the sub-flow whose first record is the flow's first hands over what that record stores
//                 whole (account.go:95): raw start, eth_protocol, MACs, identity dwords.
// J is then evicted by k_evict_dedup (nfagg_dedup.hip), which replays add_observed_intf (flows.c:76-96) over the recorded events.
// Exactness is that of the partition pass: merging sub-flow partials is associative and commutative, and a sub-flow partial is
// the same whether its records were folded on one GPU or on eight.
#include "nfagg_dedup.h"

namespace nfagg {

// Live slot `idx` of S: a sub-flow of this epoch (careful-path leftovers are not) that (n_shards, shard_id) owns?
NF_DEV bool join_subflow(const TableView& S, uint32_t idx, uint64_t seq_limit, uint32_t n_shards, uint32_t shard_id, uint64_t w[5], uint64_t& h,
                         uint32_t& ifx, uint32_t& ms) {
    const uint4* L = reinterpret_cast<const uint4*>(&S.hot[idx]);
    const uint4 a = L[0], b = L[1], c = L[2], l3 = L[3], l6 = L[6];
    const uint32_t first_inv = l6.w;                             // tag half of id0
    if (first_inv == 0 || (uint64_t)(~first_inv) >= seq_limit) return false;
    w[0] = (uint64_t)a.z | ((uint64_t)a.w << 32); w[1] = (uint64_t)b.x | ((uint64_t)b.y << 32); w[2] = (uint64_t)b.z | ((uint64_t)b.w << 32);
    w[3] = (uint64_t)c.x | ((uint64_t)c.y << 32); w[4] = (uint64_t)c.z | ((uint64_t)c.w << 32);
    h = key_hash(w);
    if (n_shards > 1 && shard_of_hash(h, n_shards) != shard_id) return false;
    ifx = l3.z;                                                  // low half of the sixth key word (SlotHot.end)
    ms = ~first_inv;
    return true;
}

__global__ __launch_bounds__(256) void k_join_claim(TableView S, TableView J, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards,
                                                    uint32_t shard_id, uint32_t* __restrict__ slot_of) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_live; i += stride) {
        uint64_t w[5], h;
        uint32_t ifx, ms, at = kNoSlot;
        if (join_subflow(S, S.live_list[i], seq_limit, n_shards, shard_id, w, h, ifx, ms)) {
            Hints x;
            at = probe_home(J, w, h, x);
            if (at == kNoSlot) { at = find_or_claim(J, w, h); x.id0 = 0; }
            if (at != kNoSlot) dedup_claim(J, at, x.id0, ifx, ms);
        }
        slot_of[i] = at;
    }
}

__global__ __launch_bounds__(256) void k_join_merge(TableView S, TableView J, uint64_t n_live, const uint32_t* __restrict__ slot_of) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_live; i += stride) {
        const uint32_t j = slot_of[i];
        if (j == kNoSlot) continue;
        const uint32_t idx = S.live_list[i];
        const SlotHot hv = S.hot[idx];
        const SlotAux* A = &S.aux[idx];
        DedupPartial p;
        p.bytes = hv.bytes; p.packets = hv.packets; p.flags = hv.flags;
        p.endl_lo = A->endl_lo; p.endl_hi = A->endl_hi;
        p.dscp_tag = hv.dscp_tag; p.samp_tag = hv.samp_tag;
        p.ssl_first = A->ssl_first; p.ssl_max = A->ssl_max; p.ssl_minv = A->ssl_minv;
        p.cs_tag = A->cs_tag; p.ks_tag = A->ks_tag;
        p.dir0 = A->dir[0][0]; p.dir1 = A->dir[0][1];
        p.ifx = (uint32_t)hv.end;
        Hints hx;
        hx.flags = 0;
        hx.id0 = J.hot[j].id0;                                   // final: every claim is a previous kernel
        if ((uint32_t)(hx.id0 >> 32) == (uint32_t)(hv.id0 >> 32)) {
            // this sub-flow's first record is the flow's first record. Other lanes are merging into the same hot line: its words
            // go out as agent-scope stores, like dedup_publish_first's; the cold half line is this lane's alone.
            SlotHot* H = &J.hot[j];
            ast(&H->start_inv, hv.start_inv);
            ast(&H->eth_tag, hv.eth_tag);
            ast(&H->smac_lo, hv.smac_lo);
            ast(&H->dmac_lo, hv.dmac_lo);
            const uint4* src = reinterpret_cast<const uint4*>(&S.cold[idx]);
            uint4* dst = reinterpret_cast<uint4*>(&J.cold[j]);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = src[k];
        }
        dedup_merge<false>(J, j, hx, p);
    }
}

static inline int grid_for(uint64_t n) { uint64_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096; return (int)g; }

hipError_t launch_subflow_join(const TableView& S, const TableView& J, uint64_t n_live, uint64_t seq_limit, uint32_t n_shards,
                               uint32_t shard_id, uint32_t* d_slot_of, hipStream_t s) {
    if (n_live == 0) return hipSuccess;
    if (!S.subflow || J.subflow || !S.aux || !J.aux || !d_slot_of) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_join_claim, dim3(grid_for(n_live)), dim3(256), 0, s, S, J, n_live, seq_limit, n_shards, shard_id, d_slot_of);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_join_merge, dim3(grid_for(n_live)), dim3(256), 0, s, S, J, n_live, d_slot_of);
    return hipGetLastError();
}

}  // namespace nfagg
