// nfagg_dedup.hip — NFAGG_MODE_KERNEL_DEDUP: the merge the eBPF datapath applies
// when one flow is seen on several interfaces ("direction dedup"),
// bpf/flows.c:98-143 update_existing_flow + :76-96 add_observed_intf, applied to
// whole records: the incoming record plays the role of one observation
// (packet + if_index + direction + tls info).
//
// Sequential semantics being reproduced, for the records r0,r1,... of one key in
// one eviction epoch (r0 is stored whole, account.go:95; F = r0.if_index_first_seen):
//   r.if_index == F ("counted")  : packets/bytes +=, end = r.end, flags |=, dscp = r.dscp,
//                                  sampling = r.sampling, ssl/tls rules of flows.c:112-126
//   r.if_index != F, != 0 ("side"): end = r.end, flags |=, add_observed_intf(if_index, direction)
//   r.if_index == 0 != F          : ignored
// Everything is resolved from per-record sequence numbers with order-free
// operations, in two steps — here (the direct kernels, small batches) two passes over the batch; batches of 65 536 records or
// more take ONE streaming pass + a partition pass that does both steps per partition (nfagg_dedup_cached.hip):
//   pass 1 k_dedup_claim : claim the slot, resolve the first record (tagged max, as in
//                          accounter mode) and the seven interfaces that appear earliest;
//   pass 2 k_dedup_fold  : F and the first record are now known exactly — sums, ORs,
//                          "last value" tags, first-record identity, and for side records
//                          the two earliest distinct directions of their interface.
// k_evict_dedup replays add_observed_intf over those (at most 14) events in sequence
// order, starting from r0's own observed list, which reproduces capacity cut-off
// (MAX_OBSERVED_INTERFACES, flows.c:79) and the merge to OBSERVED_DIRECTION_BOTH exactly.
//
// Why seven candidates suffice: F (when non-zero) is the earliest interface of the
// flow, so six of the seven are side interfaces; an interface outside the earliest
// seven arrives after six other side interfaces, each of which is in the list
// already (from r0) or was appended — the list is full and flows.c:79 ignores it.
#include "nfagg_dedup.h"

namespace nfagg {

// ---- pass 1: c.entries[id] lookup-or-insert; first record; earliest interfaces
__global__ __launch_bounds__(256) void k_dedup_claim(TableView t, const void* __restrict__ recs, uint64_t n, uint64_t seq_base) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long skipped = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Rec r;
        uint64_t w[5], h;
        if (!record_prologue(t, recs, i, r, w, h)) { skipped++; continue; }
        dedup_claim_record(t, r, w, h, (uint32_t)(seq_base + i));
    }
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
}

// ---- pass 2: update_existing_flow for every record (the first record included: it
// contributes exactly what "stored whole" keeps, see the header of this file)
__global__ __launch_bounds__(256) void k_dedup_fold(TableView t, const void* __restrict__ recs, uint64_t n, uint64_t seq_base) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Rec r;
        uint64_t w[5], h;
        if (!record_prologue(t, recs, i, r, w, h)) continue;
        dedup_fold_record(t, r, w, h, (uint32_t)(seq_base + i));
    }
}

// ---- evict: rebuild the record; replay add_observed_intf over the recorded events
__global__ __launch_bounds__(256) void k_evict_dedup(TableView t, uint64_t n_live, uint64_t seq_limit, void* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_live; i += stride) {
        const uint32_t idx = t.live_list[i];
        SlotHot* H = &t.hot[idx];
        SlotCold* C = &t.cold[idx];
        SlotAux* A = &t.aux[idx];
        const SlotHot hv = *H;
        const SlotCold cv = *C;
        const SlotAux av = *A;
        const uint32_t first_inv = (uint32_t)(hv.id0 >> 32);
        const bool emit = first_inv != 0 && (uint64_t)(~first_inv) < seq_limit;
        if (emit) {
            uint32_t d[kRecordDwords];
#pragma unroll
            for (int k = 0; k < 5; k++) { d[2 * k] = (uint32_t)hv.key[k]; d[2 * k + 1] = (uint32_t)(hv.key[k] >> 32); }
            d[10] = (uint32_t)hv.start_inv; d[11] = (uint32_t)(hv.start_inv >> 32);
            d[12] = (uint32_t)av.endl_lo; d[13] = (uint32_t)av.endl_hi;
            d[14] = (uint32_t)hv.bytes; d[15] = (uint32_t)(hv.bytes >> 32);
            d[16] = hv.packets;
            d[17] = (uint32_t)(hv.eth_tag & 0xffffu) | ((hv.flags & 0xffffu) << 16);
            const uint64_t smac = (uint64_t)(uint32_t)hv.smac_lo | ((uint64_t)(cv.smac_hi & 0xffffu) << 32);
            const uint64_t dmac = (uint64_t)(uint32_t)hv.dmac_lo | ((uint64_t)(cv.dmac_hi & 0xffffu) << 32);
            d[18] = (uint32_t)smac;
            d[19] = (uint32_t)((smac >> 32) & 0xffffu) | (uint32_t)((dmac & 0xffffu) << 16);
            d[20] = (uint32_t)(dmac >> 16);
            d[21] = (uint32_t)hv.id0;
            d[22] = cv.id[0];
#pragma unroll
            for (int k = 1; k < 12; k++) d[23 + k] = cv.id[k];
            d[35] = 0;
            d[23] = (uint32_t)hv.samp_tag;
            d[24] = (d[24] & 0xff00ffffu) | ((uint32_t)(hv.dscp_tag & 0xffu) << 16);
            // tls / ssl (flows.c:112-125)
            const uint32_t ssl = (uint32_t)av.ssl_first & 0xffffu;
            uint32_t cs = d[33] >> 16, ks = d[34] & 0xffffu, misc = d[34] >> 24;
            if (av.cs_tag) cs = (uint32_t)av.cs_tag & 0xffffu;
            if (av.ks_tag) ks = (uint32_t)av.ks_tag & 0xffffu;
            if (av.ssl_max && (0x10000u - av.ssl_minv) != av.ssl_max) misc |= kMiscSslMismatch;
            d[33] = ssl | (cs << 16);
            d[34] = ks | (((hv.flags >> 16) & 0xffu) << 16) | (misc << 24);
            // observed interfaces: start from the first record's own list, replay the events
            uint32_t nb = d[24] >> 24;
            uint32_t oi[kObservedMax], od[kObservedMax];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                oi[k] = d[27 + k];
                od[k] = k < 4 ? (d[25] >> (8 * k)) & 0xffu : (d[26] >> (8 * (k - 4))) & 0xffu;
            }
            const uint32_t F = d[21];
            uint64_t ev[2 * kCand];      // (~seq)<<8 | dir : larger = earlier
            uint32_t ex[2 * kCand];
#pragma unroll
            for (int k = 0; k < kCand; k++) {
                const uint32_t ifx = (uint32_t)av.cand[k];
                const bool side = av.cand[k] != 0 && ifx != F;
                ev[2 * k] = side ? av.dir[k][0] : 0; ev[2 * k + 1] = side ? av.dir[k][1] : 0;
                ex[2 * k] = ifx; ex[2 * k + 1] = ifx;
            }
            for (int round = 0; round < 2 * kCand && nb < kObservedMax; round++) {
                int best = 0;
                uint64_t bv = ev[0]; uint32_t bx = ex[0];
#pragma unroll
                for (int k = 1; k < 2 * kCand; k++) { if (ev[k] > bv) { bv = ev[k]; bx = ex[k]; best = k; } }
                if (bv == 0) break;
#pragma unroll
                for (int k = 0; k < 2 * kCand; k++) if (k == best) ev[k] = 0;
                const uint32_t dirn = (uint32_t)bv & 0xffu;
                // add_observed_intf (flows.c:76-96); nb < 6 holds here
                bool found = false;
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    if (!found && (uint32_t)k < nb && oi[k] == bx) {
                        found = true;
                        if (od[k] != dirn && od[k] != kDirBoth) od[k] = kDirBoth;
                    }
                }
                if (!found) {
#pragma unroll
                    for (int k = 0; k < 6; k++) if ((uint32_t)k == nb) { oi[k] = bx; od[k] = dirn; }
                    nb++;
                }
            }
            d[24] = (d[24] & 0x00ffffffu) | (nb << 24);
            d[25] = od[0] | (od[1] << 8) | (od[2] << 16) | (od[3] << 24);
            d[26] = (d[26] & 0xffff0000u) | od[4] | (od[5] << 8);
#pragma unroll
            for (int k = 0; k < 6; k++) d[27 + k] = oi[k];
            const unsigned long long pos = aadd(&t.ctr->n_out, 1ull);
            uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + pos * kRecordBytes);
#pragma unroll
            for (int k = 0; k < 9; k++) o[k] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
        }
        // the table is left as it is: the API bumps the epoch, stale slots are re-initialised by whoever claims them
    }
}

static inline int grid_for(uint64_t n, int block, int max_blocks) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

hipError_t launch_ingest_dedup(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (!t.aux) return hipErrorInvalidValue;
    const int grid = grid_for(n, 256, 256 * 8);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_dedup_claim, dim3(grid), dim3(256), 0, s, t, d_records, n, seq_base);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_dedup_fold, dim3(grid), dim3(256), 0, s, t, d_records, n, seq_base);
    return hipGetLastError();
}

__global__ void k_reset_after_evict_dedup(DevCounters* c) {
    c->n_live = 0;
    c->n_finalized = 0;
    c->aborted = 0;
    c->max_probe = 0;
}

hipError_t launch_evict_dedup(const TableView& t, uint64_t n_live, uint64_t seq_limit, void* d_out, hipStream_t s) {
    if (n_live) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_evict_dedup, dim3(grid_for(n_live, 256, 256 * 8)), dim3(256), 0, s, t, n_live, seq_limit, d_out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_reset_after_evict_dedup, dim3(1), dim3(1), 0, s, t.ctr);
    return hipGetLastError();
}

}  // namespace nfagg
