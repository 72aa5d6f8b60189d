// nfagg_spill.h — the spill side of a two-pass partitioned fold, as a per-lane helper: a record that gets no LDS cache
// entry in pass 1 is appended (by index) to the queue of its partition; pass 2 runs one workgroup per partition. Same
// protocol as nfagg_ingest_part.hip (which keeps its own, hand-scheduled copy): indices are staged four at a time per
// partition in LDS so that a spill costs one 16-byte store and a quarter of a queue reservation; the lane that fills a group
// drains it; the reservation is a returning atomic whose result is used one tile later; a spill that finds its group full is
// retried next tile and goes to the overflow list the second time. Used by the kernel-dedup passes (nfagg_dedup_cached.hip).
//
// Per tile:   ... phase A ... __syncthreads(); ... phase B ...; lane.drain(); __syncthreads(); lane.append(...);
// After the loop: lane.finish()  (contains the barriers it needs).
#pragma once
#include "nfagg_device.h"

namespace nfagg {
namespace spill {

constexpr int kStage = 4;

struct Stage {
    uint32_t buf[kSpillParts][kStage];
    uint32_t cnt[kSpillParts];
};

NF_DEV uint32_t part_of(uint64_t h, uint32_t shift) { return (uint32_t)(h >> shift) & (kSpillParts - 1); }

NF_DEV void overflow_push(const SpillView& q, uint4 v) {
    const uint32_t at = aadd(q.ovf_tail, 4u);
    if (at + 4 > q.ovf_cap) { atomicExch(q.error, 5u); return; }
    if ((at & 3u) == 0) *reinterpret_cast<uint4*>(q.ovf + at) = v;
    else { q.ovf[at] = v.x; q.ovf[at + 1] = v.y; q.ovf[at + 2] = v.z; q.ovf[at + 3] = v.w; }   // behind single items: not 16-byte aligned
}
// One item (a spill that found its staging group full twice in a row): ONE slot, so that the list's worst case stays one slot
// per record — a batch whose uncached records all belong to one partition (a handful of flows) sends nearly all of them here.
NF_DEV void overflow_push_one(const SpillView& q, uint32_t item) {
    const uint32_t at = aadd(q.ovf_tail, 1u);
    if (at < q.ovf_cap) q.ovf[at] = item;
    else atomicExch(q.error, 5u);
}

template <int BLOCK>
struct Lane {
    static constexpr int kMine = kSpillParts / BLOCK;       // finish(): this lane looks after partitions tid, tid + BLOCK, ...
    static constexpr uint32_t kNoPart = 0xffffffffu;
    // Fill-driven drains (as k_pass1 of nfagg_ingest_part.hip): the lane whose append FILLS a staging group (position
    // kStage - 1) drains it one tile later and stores it another tile later, when the queue reservation (a returning atomic) has
    // long arrived — no lane polls the group counters (two LDS reads per lane and tile in the first version of this helper), no
    // HBM round trip inside a tile. A lane fills at most two groups per tile: one with a carried-over spill, one with its own.
    uint4 pend_v[2];
    uint32_t pend_at[2], pend_p[2], fill_p[2];
    bool pend[2];
    uint32_t carry, carry_p;

    NF_DEV void init(Stage& S, int tid) {
#pragma unroll
        for (int k = 0; k < 2; k++) { pend[k] = false; pend_at[k] = 0; pend_p[k] = 0; fill_p[k] = kNoPart; pend_v[k] = make_uint4(0, 0, 0, 0); }
        carry = 0xffffffffu; carry_p = 0;
        for (int p = tid; p < kSpillParts; p += BLOCK) S.cnt[p] = 0;     // the caller's next barrier publishes it
    }
    // between the two barriers of a tile: store the groups reserved one tile ago, take the groups this lane filled in the last tile
    NF_DEV void drain(Stage& S, const SpillView& q, int tid) {
        (void)tid;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (pend[k]) {
                if (pend_at[k] + kStage <= q.qcap) *reinterpret_cast<uint4*>(q.queue + (uint64_t)pend_p[k] * q.qcap + pend_at[k]) = pend_v[k];
                else overflow_push(q, pend_v[k]);                       // partition queue full (adversarial skew)
                pend[k] = false;
            }
            if (fill_p[k] != kNoPart) {                                 // every append of the group is in LDS (barriers since)
                const uint32_t p = fill_p[k];
                pend_v[k] = *reinterpret_cast<const uint4*>(S.buf[p]);
                S.cnt[p] = 0;                                           // appends resume after the next barrier
                pend_at[k] = aadd(&q.qtail[p], (uint32_t)kStage);
                pend_p[k] = p; pend[k] = true; fill_p[k] = kNoPart;
            }
        }
    }
    // after the second barrier: this lane's spill of the tile (if any), and last tile's carried one first
    NF_DEV void append(Stage& S, const SpillView& q, bool spill_now, uint32_t p, uint32_t idx) {
        if (carry != 0xffffffffu) {
            const uint32_t at = atomicAdd(&S.cnt[carry_p], 1u);
            if (at < (uint32_t)kStage) { S.buf[carry_p][at] = carry; if (at == (uint32_t)kStage - 1) fill_p[0] = carry_p; }
            else overflow_push_one(q, carry);   // full twice in a row: very rare
            carry = 0xffffffffu;
        }
        if (spill_now) {
            const uint32_t at = atomicAdd(&S.cnt[p], 1u);
            if (at < (uint32_t)kStage) { S.buf[p][at] = idx; if (at == (uint32_t)kStage - 1) fill_p[1] = p; }
            else { carry = idx; carry_p = p; }
        }
    }
    // after the tile loop (every lane of the workgroup must call it)
    NF_DEV void finish(Stage& S, const SpillView& q, int tid) {
        __syncthreads();
        if (carry != 0xffffffffu) {
            const uint32_t at = atomicAdd(&S.cnt[carry_p], 1u);
            if (at < (uint32_t)kStage) S.buf[carry_p][at] = carry;
            else overflow_push_one(q, carry);
        }
        __syncthreads();
        // pending groups, then whatever is staged (padded with invalid indices): groups filled in the last tile are still in LDS
        // (their fill marks are dropped here), every lane looks after its partitions
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (pend[k]) {
                if (pend_at[k] + kStage <= q.qcap) *reinterpret_cast<uint4*>(q.queue + (uint64_t)pend_p[k] * q.qcap + pend_at[k]) = pend_v[k];
                else overflow_push(q, pend_v[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < kMine; k++) {
            const int p = tid + k * BLOCK;
            uint32_t c = S.cnt[p];
            if (c > (uint32_t)kStage) c = kStage;
            if (c) {                                                    // a partial (or just filled) group, padded with invalid indices
                const uint32_t at = aadd(&q.qtail[p], (uint32_t)kStage);
                uint4 v = *reinterpret_cast<const uint4*>(S.buf[p]);
                if (c < 2) v.y = 0xffffffffu;
                if (c < 3) v.z = 0xffffffffu;
                if (c < 4) v.w = 0xffffffffu;
                if (at + kStage <= q.qcap) *reinterpret_cast<uint4*>(q.queue + (uint64_t)p * q.qcap + at) = v;
                else overflow_push(q, v);
            }
        }
    }
};

}  // namespace spill
}  // namespace nfagg
