// nfagg_kernels.hip — ingest / claim / evict kernels of the flow table (gfx950).
//
// Replaces the body of Accounter.Account's record arm and Accounter.evict
// (pkg/flow/account.go:81-96, 102-124). HBM-bound hash/scatter work: no MFMA.
#include <hipcub/hipcub.hpp>
#include "nfagg_device.h"

namespace nfagg {

// ------------------------------------------------------------------
// Variant 0 ("direct"): one record per lane, every record merged into the
// table with agent-scope atomics. Correct for any stream; hot keys serialise
// on their slot — the LDS-combining variants exist for that.
// ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ingest_direct(TableView t, const void* __restrict__ recs,
                                                       uint64_t n, uint64_t seq_base) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long skipped = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Rec r;
        load_record(recs, i, r);
        r.canonicalize();
        uint64_t w[5];
        r.key_words(w);
        const uint64_t h = key_hash(w);
        if (t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id) { skipped++; continue; }
        Partial p;
        partial_from_record(r, seq_base + i, p);
        upsert_partial(t, w, h, p);
    }
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
}

// ------------------------------------------------------------------
// Careful path (possible "full" eviction inside the batch, account.go:85-94).
// Phase A: claim slots only and plant first_inv, remember each record's slot.
// ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_claim(TableView t, const void* __restrict__ recs, uint64_t n,
                                               uint64_t seq_base, uint32_t* __restrict__ slot_idx) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Rec r;
        load_record(recs, i, r);
        r.canonicalize();
        uint64_t w[5];
        r.key_words(w);
        const uint64_t h = key_hash(w);
        if (t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id) { slot_idx[i] = kNoSlot; continue; }
        const uint32_t idx = find_or_claim(t, w, h);
        slot_idx[i] = idx;
        // plant the first-record tracker (tagged dword 21) so that phase B can find first occurrences
        if (idx != kNoSlot) amax(&t.hot[idx].id0, tagged(~(uint32_t)(seq_base + i), r.d[21]));
    }
}

// flags[i] = record i is the first record (in arrival order) of a key that was
// not in the table before this chunk: exactly the records at which
// len(c.entries) grows (account.go:95). A slot older than the chunk carries a
// first-record tag larger than every ~seq of the chunk, so equality identifies both.
__global__ __launch_bounds__(kFlagBlock) void k_first_flags(TableView t, const uint32_t* __restrict__ slot_idx,
                                                            uint64_t n, uint64_t seq_base,
                                                            uint8_t* __restrict__ flags,
                                                            uint32_t* __restrict__ block_counts) {
    __shared__ unsigned int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * kFlagBlock + threadIdx.x;
    unsigned f = 0;
    if (i < n) {
        const uint32_t idx = slot_idx[i];
        if (idx != kNoSlot) f = ((uint32_t)(t.hot[idx].id0 >> 32) == ~(uint32_t)(seq_base + i)) ? 1u : 0u;
        flags[i] = (uint8_t)f;
    }
    if (f) atomicAdd(&cnt, 1u);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = cnt;
}

// ------------------------------------------------------------------
// Evict: Accounter.evict (account.go:102-124) up to NewRecord. One lane per
// claimed slot: rebuild the 144-byte flow_record_t, write it densely, zero
// the slot (zero is every field's identity, so the next epoch needs no init).
// ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_evict(TableView t, uint64_t n_live, uint64_t seq_limit,
                                               void* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_live; i += stride) {
        const uint32_t idx = t.live_list[i];
        SlotHot* H = &t.hot[idx];
        SlotCold* C = &t.cold[idx];
        const SlotHot hv = *H;
        const SlotCold cv = *C;
        const uint32_t first_inv = (uint32_t)(hv.id0 >> 32);
        // A slot claimed by the careful path for a key that first appears at or
        // after the split point is not part of this epoch: drop it.
        const bool emit = first_inv != 0 && (uint64_t)(~first_inv) < seq_limit;
        if (emit) {
            uint32_t d[kRecordDwords];
#pragma unroll
            for (int k = 0; k < 5; k++) { d[2 * k] = (uint32_t)hv.key[k]; d[2 * k + 1] = (uint32_t)(hv.key[k] >> 32); }
            const uint64_t start = hv.start_inv ? ~hv.start_inv : 0ull;
            d[10] = (uint32_t)start; d[11] = (uint32_t)(start >> 32);
            d[12] = (uint32_t)hv.end; d[13] = (uint32_t)(hv.end >> 32);
            d[14] = (uint32_t)hv.bytes; d[15] = (uint32_t)(hv.bytes >> 32);
            d[16] = hv.packets;
            d[17] = (uint32_t)(hv.eth_tag & 0xffffu) | ((hv.flags & 0xffffu) << 16);
            const uint64_t smac = (uint64_t)(uint32_t)hv.smac_lo | ((uint64_t)(cv.smac_hi & 0xffffu) << 32);
            const uint64_t dmac = (uint64_t)(uint32_t)hv.dmac_lo | ((uint64_t)(cv.dmac_hi & 0xffffu) << 32);
            d[18] = (uint32_t)smac;
            d[19] = (uint32_t)((smac >> 32) & 0xffffu) | (uint32_t)((dmac & 0xffffu) << 16);
            d[20] = (uint32_t)(dmac >> 16);
            d[21] = (uint32_t)hv.id0;
#pragma unroll
            for (int k = 0; k < 14; k++) d[22 + k] = (uint32_t)cv.id[k];
            d[23] = (uint32_t)hv.samp_tag;                                   // sampling: last non-zero
            d[24] = (d[24] & 0xff00ffffu) | ((uint32_t)(hv.dscp_tag & 0xffu) << 16);  // dscp: last non-zero
            const unsigned long long pos = aadd(&t.ctr->n_out, 1ull);
            uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + pos * kRecordBytes);
#pragma unroll
            for (int k = 0; k < 9; k++) o[k] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
        }
        uint4* hz = reinterpret_cast<uint4*>(H);
        uint4* cz = reinterpret_cast<uint4*>(C);
#pragma unroll
        for (int k = 0; k < 8; k++) { hz[k] = make_uint4(0, 0, 0, 0); cz[k] = make_uint4(0, 0, 0, 0); }
    }
}

__global__ void k_reset_after_evict(DevCounters* c) {
    c->n_live = 0;
    c->aborted = 0;
    c->max_probe = 0;
}

// ------------------------------------------------------------------
// Optimistic fold (nfagg_api.hip): raw slot copies, discard, first sequence numbers.
// One 16-byte chunk per lane: consecutive lanes copy consecutive chunks of one slot's lines.
// ------------------------------------------------------------------
template <bool RESTORE>
__global__ __launch_bounds__(256) void k_snapshot(TableView t, uint64_t n, uint4* __restrict__ snap) {
    const int per = t.aux ? 32 : 16;                       // 16-byte chunks per slot: hot 8 + cold 8 (+ aux 16)
    const uint64_t total = n * (uint64_t)per, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint64_t i = g / per; const int c = (int)(g % per);
        const uint32_t idx = t.live_list[i];
        uint4* p; uint4* q;
        if (c < 8) { p = reinterpret_cast<uint4*>(&t.hot[idx]) + c; q = snap + i * 8 + c; }
        else if (c < 16) { p = reinterpret_cast<uint4*>(&t.cold[idx]) + (c - 8); q = snap + n * 8 + i * 8 + (c - 8); }
        else { p = reinterpret_cast<uint4*>(&t.aux[idx]) + (c - 16); q = snap + n * 16 + i * 16 + (c - 16); }
        if (RESTORE) *p = *q; else *q = *p;
    }
}

__global__ __launch_bounds__(256) void k_discard(TableView t, uint64_t from, uint64_t to) {
    const int per = t.aux ? 32 : 16;
    const uint64_t total = (to - from) * (uint64_t)per, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint64_t i = from + g / per; const int c = (int)(g % per);
        const uint32_t idx = t.live_list[i];
        uint4* p = c < 8 ? reinterpret_cast<uint4*>(&t.hot[idx]) + c
                 : c < 16 ? reinterpret_cast<uint4*>(&t.cold[idx]) + (c - 8) : reinterpret_cast<uint4*>(&t.aux[idx]) + (c - 16);
        *p = make_uint4(0, 0, 0, 0);
    }
}

// out[i - from] = epoch-relative sequence number of the first record of the flow in slot live_list[i]
// (the tag of id0, resolved by every fold kernel in both modes).
__global__ __launch_bounds__(256) void k_first_seqs(TableView t, uint64_t from, uint64_t to, uint32_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < to; i += stride)
        out[i - from] = ~(uint32_t)(t.hot[t.live_list[i]].id0 >> 32);
}

static inline int grid_for(uint64_t n, int block, int max_blocks) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

hipError_t launch_ingest_cached(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, uint64_t seq_base,
                                int variant, hipStream_t s);  // nfagg_ingest_cached.hip

// Default kernel by batch size, measured on configs[1]'s stream (profiles/r01e_batch_size_crossover.txt):
//   below 6 144 records the direct kernel (one record per lane, HBM atomics; no LDS cache to set up and flush): 14 / 24 us
//     per 1 024 / 4 096 records against 34 / 35 us for the cached kernel;
//   below 3 Mi records the single-pass LDS-cached kernel: 0.046 ms per 65 536 records against 0.098 ms for the three
//     launches of the two-pass fold, 0.46 against 0.51 ms at 2 Mi;
//   from there the two-pass partitioned fold: 0.74 against 0.79 ms at 4 Mi, 5.5 against 12.5 ms at 100 M.
constexpr uint64_t kDirectMaxBatch = 6144;
constexpr uint64_t kPartMinBatch = 3u << 20;
constexpr uint64_t kDedupCachedMinBatch = 1u << 16;
static bool takes_two_pass(int variant, uint64_t n) { return (variant >= 8 && variant <= 11) || (variant == 0 && n >= kPartMinBatch); }
// Shipping variants: 0 (by batch size), 1 direct, 3/4/5/7 geometries of the single-pass cached kernel, 10/11 two-pass
// (with / without the admission filter). 6/8/9 are the phase-timing builds and exist only in libnfagg_diag.so (-DNFAGG_DIAG).
bool ingest_variant_supported(int variant) {
#ifdef NFAGG_DIAG
    if (variant == 6 || variant == 8 || variant == 9) return true;
#endif
    return variant == 0 || variant == 1 || variant == 3 || variant == 4 || variant == 5 || variant == 7 || variant == 10 || variant == 11;
}
static bool takes_direct(int variant, uint64_t n, uint32_t sketch_flags) {
    return variant == 1 || (variant == 0 && n < kDirectMaxBatch && sketch_flags == 0);   // with sketches on, the cached kernel fuses them: one launch
}
static bool dedup_takes_cached(int variant, uint64_t n) { return !(variant == 1 || (variant != 10 && n < kDedupCachedMinBatch)); }
bool ingest_needs_spill(int mode, int variant, uint64_t n) { return mode == 0 ? takes_two_pass(variant, n) : dedup_takes_cached(variant, n); }
bool ingest_fuses_sketches(int mode, int variant, uint64_t n, uint32_t sketch_flags) {
    return mode == 0 && !takes_direct(variant, n, sketch_flags) && variant != 6 && variant != 8 && variant != 9;
}

hipError_t launch_ingest(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, uint64_t seq_base,
                         int mode, int variant, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (mode == 1) {   // NFAGG_MODE_KERNEL_DEDUP: LDS-cached passes; direct per-record passes for small batches (variant 1: always, 10: never)
        if (!dedup_takes_cached(variant, n)) return launch_ingest_dedup(t, d_records, n, seq_base, s);
        return launch_ingest_dedup_cached(t, d_records, n, seq_base, s);
    }
    // 0 (default): by batch size (see kDirectMaxBatch / kPartMinBatch above) — direct kernel, single-pass cached kernel
    // (what variant 7 always runs), two-pass partitioned fold (nfagg_ingest_part.hip; 8/9 = its phase-timing builds).
    // 3..5: other geometries of the cached kernel, 6: its phase-timing build (diag library only); 1: direct always.
    if (takes_two_pass(variant, n))   // 10: two-pass whatever the size; 11: same without the admission filter
        return launch_ingest_part(t, sk, t.spill, d_records, n, seq_base, variant, s);
    if (!takes_direct(variant, n, sk.flags)) return launch_ingest_cached(t, sk, d_records, n, seq_base, variant, s);
    (void)hipGetLastError(); hipLaunchKernelGGL(k_ingest_direct, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, s, t, d_records, n, seq_base);
    return hipGetLastError();
}

hipError_t launch_claim(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base,
                        uint32_t* d_slot_idx, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError(); hipLaunchKernelGGL(k_claim, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, s, t, d_records, n, seq_base, d_slot_idx);
    return hipGetLastError();
}

hipError_t launch_first_flags(const TableView& t, const uint32_t* d_slot_idx, uint64_t n, uint64_t seq_base,
                              uint8_t* d_flags, uint32_t* d_block_counts, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int blocks = (int)((n + kFlagBlock - 1) / kFlagBlock);
    (void)hipGetLastError(); hipLaunchKernelGGL(k_first_flags, dim3(blocks), dim3(kFlagBlock), 0, s, t, d_slot_idx, n, seq_base, d_flags, d_block_counts);
    return hipGetLastError();
}

// Radix sort of the live list by slot index (rocPRIM through hipCUB). Eviction visits the claimed slots of a table that
// can span tens of GiB; in claim order every access lands on another page and the kernel is bound by address
// translation (0.16 ms for 287 k flows of a 64 GiB table against 0.08 ms of a 2 GiB one); in slot order consecutive
// lanes walk the table front to back. temp == nullptr: size query.
hipError_t launch_sort_slots(const uint32_t* d_in, uint32_t* d_out, uint64_t n, int end_bit, void* d_temp, size_t* temp_bytes, hipStream_t s) {
    return hipcub::DeviceRadixSort::SortKeys(d_temp, *temp_bytes, d_in, d_out, (int)n, 0, end_bit, s);
}

size_t snapshot_bytes(const TableView& t, uint64_t n) { return (size_t)n * (t.aux ? 512 : 256); }

hipError_t launch_snapshot(const TableView& t, uint64_t n, void* d_snap, bool restore, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int grid = grid_for(n * (t.aux ? 32 : 16), 256, 256 * 16);
    (void)hipGetLastError();
    if (restore) hipLaunchKernelGGL(k_snapshot<true>, dim3(grid), dim3(256), 0, s, t, n, (uint4*)d_snap);
    else hipLaunchKernelGGL(k_snapshot<false>, dim3(grid), dim3(256), 0, s, t, n, (uint4*)d_snap);
    return hipGetLastError();
}

hipError_t launch_discard(const TableView& t, uint64_t from, uint64_t to, hipStream_t s) {
    if (to <= from) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_discard, dim3(grid_for((to - from) * (t.aux ? 32 : 16), 256, 256 * 16)), dim3(256), 0, s, t, from, to);
    return hipGetLastError();
}

hipError_t launch_first_seqs(const TableView& t, uint64_t from, uint64_t to, uint32_t* d_out, hipStream_t s) {
    if (to <= from) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_first_seqs, dim3(grid_for(to - from, 256, 256 * 8)), dim3(256), 0, s, t, from, to, d_out);
    return hipGetLastError();
}

hipError_t launch_sort_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n, void* d_temp, size_t* temp_bytes, hipStream_t s) {
    return hipcub::DeviceRadixSort::SortKeys(d_temp, *temp_bytes, d_in, d_out, (int)n, 0, 32, s);
}

hipError_t launch_evict(const TableView& t, uint64_t n_live, uint64_t seq_limit, void* d_out, hipStream_t s) {
    if (t.aux) return launch_evict_dedup(t, n_live, seq_limit, d_out, s);
    if (n_live) {
        (void)hipGetLastError(); hipLaunchKernelGGL(k_evict, dim3(grid_for(n_live, 256, 256 * 8)), dim3(256), 0, s, t, n_live, seq_limit, d_out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    (void)hipGetLastError(); hipLaunchKernelGGL(k_reset_after_evict, dim3(1), dim3(1), 0, s, t.ctr);
    return hipGetLastError();
}

}  // namespace nfagg
