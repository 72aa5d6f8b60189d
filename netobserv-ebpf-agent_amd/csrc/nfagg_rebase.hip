// nfagg_rebase.hip — the sequence WINDOW: an epoch is never ended for lack of sequence numbers.
//
// The reference's Accounter folds records in arrival order for as long as the map has room and no tick fires
// (pkg/flow/account.go:58-100): an epoch has no maximum length. The parallel fold orders records by a sequence number that
// the slots carry in 32-bit tags (nfagg_internal.h: "first" words hold ~seq, "last non-zero" words hold seq + 1). The host
// counts sequence numbers in 64 bits; what the device sees is `seq - origin`, and when the window [origin, origin + 2^32 - 16)
// is used up the tags already in the table are REBASED instead of the epoch being evicted (round 2 returned NFAGG_FULL there:
// 8 PCIe-fed GPUs reach 2^32 records in 1.4 s, less than CACHE_ACTIVE_TIMEOUT).
//
// Why that is exact: a tag is only ever compared with other tags of the SAME word of the SAME flow (atomic max per word), and
// every record still to come is later than every record folded so far. So all that the old tags of a slot have to keep is
// their order among themselves, below everything new:
//   accounter mode   every tagged word is independent of the others: the old winner of each word becomes "record 0" (a first
//                    word keeps its data with tag ~0, a last-non-zero word keeps its value with seq + 1 = 1);
//   kernel-dedup     the earliest interfaces (aux.cand[]) and their earliest directions (aux.dir[][]) are replayed in
//                    sequence order at the eviction (bpf/flows.c:76-96 add_observed_intf): the up to 21 sequence numbers of
//                    a slot are replaced by their RANKS (equal numbers — one record — stay equal); the other words as above.
// The new window starts at kRebaseKeep, above every rank. One pass over the live slots (192 B each, plus the aux lines).
//
// Across GPUs (local fold: one flow on several GPUs, tags compared when the partials meet at the owner) collapsing each GPU's
// tags on its own would lose the order between GPUs: there the flows are first brought together at their owners (partials
// exported, table emptied by the epoch tag, partials merged back at the owner: nfagg_group.inc group_consolidate,
// nfagg_window_restart_device), and the owner's single copy is rebased.
#include "nfagg_device.h"

namespace nfagg {

constexpr uint32_t kRebaseKeep = 32;     // first sequence number of the new window (ranks 0..20 are taken)

NF_DEV uint64_t collapse_first(uint64_t word) { return (word >> 32) ? tagged(0xFFFFFFFFu, (uint32_t)word) : word; }
NF_DEV uint64_t collapse_last(uint64_t word, int value_bits) {
    return word ? ((1ull << value_bits) | (word & ((1ull << value_bits) - 1ull))) : 0ull;
}

template <bool DEDUP>
__global__ __launch_bounds__(256) void k_rebase(TableView t) {
    const uint64_t n = t.ctr->n_live;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t idx = t.live_list[i];
        SlotHot* H = &t.hot[idx];
        SlotCold* C = &t.cold[idx];
        H->id0 = collapse_first(H->id0);
        H->smac_lo = collapse_first(H->smac_lo); H->dmac_lo = collapse_first(H->dmac_lo);
        C->smac_hi = collapse_first(C->smac_hi); C->dmac_hi = collapse_first(C->dmac_hi);
        H->dscp_tag = collapse_last(H->dscp_tag, 8);
        H->samp_tag = collapse_last(H->samp_tag, 32);
        if (!DEDUP) {
            H->eth_tag = collapse_last(H->eth_tag, 16);          // (kernel-dedup keeps the first record's raw eth_protocol here)
        } else {
            SlotAux* A = &t.aux[idx];
            A->endl_lo = collapse_last(A->endl_lo, 32); A->endl_hi = collapse_last(A->endl_hi, 32);
            A->ssl_first = collapse_first(A->ssl_first);
            A->cs_tag = collapse_last(A->cs_tag, 16); A->ks_tag = collapse_last(A->ks_tag, 16);
            // ranks of the (up to 21) sequence numbers of the candidate interfaces and their directions
            uint32_t inv[21];
#pragma unroll
            for (int k = 0; k < 7; k++) {
                inv[k] = (uint32_t)(A->cand[k] >> 32);
                inv[7 + 2 * k] = A->dir[k][0] ? (uint32_t)(A->dir[k][0] >> 8) : 0u;
                inv[8 + 2 * k] = A->dir[k][1] ? (uint32_t)(A->dir[k][1] >> 8) : 0u;
            }
            uint32_t rank[21];
            for (int a = 0; a < 21; a++) {
                uint32_t r = 0;
                if (inv[a]) {
                    // earlier record = larger inverted number: rank = how many DISTINCT larger values there are
                    for (int b = 0; b < 21; b++) {
                        if (inv[b] > inv[a]) {
                            bool first_of_its_value = true;
                            for (int c2 = 0; c2 < b; c2++) first_of_its_value &= (inv[c2] != inv[b]);
                            if (first_of_its_value) r++;
                        }
                    }
                }
                rank[a] = r;
            }
#pragma unroll
            for (int k = 0; k < 7; k++) {
                if (A->cand[k]) A->cand[k] = tagged(~rank[k], (uint32_t)A->cand[k]);
                if (A->dir[k][0]) A->dir[k][0] = ((uint64_t)(~rank[7 + 2 * k]) << 8) | (A->dir[k][0] & 0xffull);
                if (A->dir[k][1]) A->dir[k][1] = ((uint64_t)(~rank[8 + 2 * k]) << 8) | (A->dir[k][1] & 0xffull);
            }
        }
    }
}

uint32_t rebase_keep() { return kRebaseKeep; }

// In place, over the slots of the live list (the count is read on the device: no host round trip). Plain accesses: the launch
// sits between two kernel boundaries, one lane per slot.
hipError_t launch_rebase(const TableView& t, hipStream_t s) {
    (void)hipGetLastError();
    if (t.aux) hipLaunchKernelGGL(k_rebase<true>, dim3(1024), dim3(256), 0, s, t);
    else hipLaunchKernelGGL(k_rebase<false>, dim3(1024), dim3(256), 0, s, t);
    return hipGetLastError();
}

}  // namespace nfagg
