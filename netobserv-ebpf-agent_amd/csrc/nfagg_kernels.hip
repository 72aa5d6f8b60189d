// nfagg_kernels.hip — ingest / claim / evict kernels of the flow table (gfx950).
//
// Replaces the body of Accounter.Account's record arm and Accounter.evict
// (pkg/flow/account.go:81-96, 102-124). HBM-bound hash/scatter work: no MFMA.
#include <hipcub/hipcub.hpp>
#include "nfagg_dedup.h"

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
        // sub-flow tables (kernel-dedup mode of a local-fold rank, nfagg_dedup.h): the key is (flow, if_index_first_seen)
        const uint32_t idx = t.subflow ? find_or_claim(t, w, sub_hash(t, h, r.d[21]), nullptr, nullptr, nullptr, sub_kx(t, r.d[21]))
                                       : find_or_claim(t, w, h);
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
// Evict: Accounter.evict (account.go:102-124) up to NewRecord. Reads one 128-byte hot line + one 64-byte cold half line
// per flow and writes one 144-byte record; nothing is zeroed (epoch tags, nfagg_internal.h). A wave takes 64 claimed
// slots at a time: eight lanes fetch one hot line (16 bytes each: every load instruction covers eight whole lines), four
// lanes one cold half line, through LDS; then one lane per slot rebuilds its record into the wave's LDS window at the
// rank a ballot gave it (one n_out atomic per wave, not per lane), and the window — the wave's records back to back —
// goes out with 16-byte stores, consecutive lanes on consecutive addresses.
// ------------------------------------------------------------------
#ifdef NFAGG_DIAG
// Experiment (libnfagg_diag.so only; round-2..5 reviews: "a dense 64-byte identity record per live-list position"): when set,
// k_finalize writes the first record's identity words to g_diag_dense[live-list position] and k_evict reads them from there —
// sequential 64-byte units instead of one random cold half line per flow (whose other half is a neighbour's: 128 bytes fetched
// for 64). The first-MAC high words stay in the cold line, which k_evict does NOT read in this mode: evicted MACs are WRONG (their
// high 16 bits zero) — a timing experiment for k_finalize + k_evict, not a layout. profiles/r06x_dense_identity.txt.
__device__ SlotCold* g_diag_dense = nullptr;
hipError_t diag_set_dense(void* p) { SlotCold* v = static_cast<SlotCold*>(p); return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_dense), &v, sizeof v); }
#define NF_DENSE_AT(pos, idx) (g_diag_dense ? &g_diag_dense[pos] : &t.cold[idx])
#else
#define NF_DENSE_AT(pos, idx) (&t.cold[idx])
#endif
constexpr int kEvictWaves = 2;                        // waves per workgroup
// Rows are padded to an odd number of 16-byte chunks (9 and 5): lane l reading chunk c of ITS row touches banks
// (36 l + 4 c) mod 32 resp. (20 l + 4 c) mod 32 — eight consecutive lanes cover all 32 banks, a 16-byte read per lane runs
// at the LDS's full rate. (Unpadded 128-byte rows put all 64 lanes on the same four banks.)
struct EvictWaveLds {
    uint4 hot[64][9];                                 // 9 KiB; reused as the output window: 64 records x 144 B
    uint4 cold[64][5];                                // 5 KiB
};
static_assert(sizeof(uint4) * 9 == kRecordBytes, "a padded hot row is exactly one output record");

// FILTER = false (no split pending: every claimed slot is a flow of this epoch): slot i of the live list goes to out[i],
// no counter at all — 16 k returning atomics on the one n_out word, one per wave, were what bounded this kernel
// (0.22 ms for 1 M flows whatever the traffic). FILTER = true: slots claimed for keys that first appear at or after the
// split point are dropped (careful path, nfagg_api.hip), positions come from n_out.
template <bool FILTER>
__global__ __launch_bounds__(64 * kEvictWaves) void k_evict(TableView t, uint64_t n_live, uint64_t seq_limit,
                                                            void* __restrict__ out) {
    __shared__ EvictWaveLds lds[kEvictWaves];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    EvictWaveLds& W = lds[wv];
    const uint64_t wave_id = (uint64_t)blockIdx.x * kEvictWaves + wv, n_waves = (uint64_t)gridDim.x * kEvictWaves;
    for (uint64_t base = wave_id * 64; base < n_live; base += n_waves * 64) {
        const uint64_t cnt = n_live - base < 64 ? n_live - base : 64;
        const uint32_t my_idx = (uint64_t)lane < cnt ? t.live_list[base + lane] : 0u;
        // ---- cooperative fetch: pass p covers slots 8p..8p+7 (hot), 16p..16p+15 (cold)
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int s = 8 * p + (lane >> 3);
            const uint32_t idx = __shfl(my_idx, s);
            if ((uint64_t)s < cnt) W.hot[s][lane & 7] = reinterpret_cast<const uint4*>(&t.hot[idx])[lane & 7];
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int s = 16 * p + (lane >> 2);
            const uint32_t idx = __shfl(my_idx, s);
            if ((uint64_t)s < cnt) W.cold[s][lane & 3] = reinterpret_cast<const uint4*>(NF_DENSE_AT(base + s, idx))[lane & 3];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- one lane per slot: rebuild the record in registers
        uint32_t d[kRecordDwords];
        bool emit = false;
        if ((uint64_t)lane < cnt) {
            uint64_t hq[16], cq[2];
            uint32_t ci[12];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint4 v = W.hot[lane][k];
                hq[2 * k] = (uint64_t)v.x | ((uint64_t)v.y << 32); hq[2 * k + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
            }
            {
                const uint4 v0 = W.cold[lane][0], v1 = W.cold[lane][1], v2 = W.cold[lane][2], v3 = W.cold[lane][3];
                cq[0] = (uint64_t)v0.x | ((uint64_t)v0.y << 32); cq[1] = (uint64_t)v0.z | ((uint64_t)v0.w << 32);
                ci[0] = v1.x; ci[1] = v1.y; ci[2] = v1.z; ci[3] = v1.w; ci[4] = v2.x; ci[5] = v2.y; ci[6] = v2.z; ci[7] = v2.w;
                ci[8] = v3.x; ci[9] = v3.y; ci[10] = v3.z; ci[11] = v3.w;
            }
            const uint64_t id0 = hq[13];
            const uint32_t first_inv = (uint32_t)(id0 >> 32);
            // A slot claimed by the careful path for a key that first appears at or after the split point is not part of
            // this epoch: drop it.
            emit = first_inv != 0 && (uint64_t)(~first_inv) < seq_limit;
            if (!FILTER && !emit) { atomicExch(&t.ctr->error, 6u); emit = true; }   // cannot happen: every claimer merges its record
            if (FILTER && emit && t.n_shards > 1) {
                // group local-fold mode: a table also holds flows of other shards (folded where they arrived, merged into
                // their owners at the tick); only the flows this shard owns leave through it. (A table that was fed through
                // the shard filter holds nothing else and passes unchanged.)
                uint64_t kw[5];
#pragma unroll
                for (int k = 0; k < 5; k++) kw[k] = hq[1 + k];
                emit = shard_of_hash(key_hash(kw), t.n_shards) == t.shard_id;
            }
#pragma unroll
            for (int k = 0; k < 5; k++) { d[2 * k] = (uint32_t)hq[1 + k]; d[2 * k + 1] = (uint32_t)(hq[1 + k] >> 32); }
            const uint64_t bytes = hq[6], end = hq[7], start_inv = hq[8], pf = hq[9], eth_tag = hq[10], dscp_tag = hq[11],
                           samp_tag = hq[12], smac_lo = hq[14], dmac_lo = hq[15];
            const uint64_t start = start_inv ? ~start_inv : 0ull;
            d[10] = (uint32_t)start; d[11] = (uint32_t)(start >> 32);
            d[12] = (uint32_t)end; d[13] = (uint32_t)(end >> 32);
            d[14] = (uint32_t)bytes; d[15] = (uint32_t)(bytes >> 32);
            d[16] = (uint32_t)pf;                                               // packets
            d[17] = (uint32_t)(eth_tag & 0xffffu) | (((uint32_t)(pf >> 32) & 0xffffu) << 16);   // eth (last non-zero) | flags
            const uint64_t smac = (uint64_t)(uint32_t)smac_lo | ((uint64_t)(cq[0] & 0xffffu) << 32);
            const uint64_t dmac = (uint64_t)(uint32_t)dmac_lo | ((uint64_t)(cq[1] & 0xffffu) << 32);
            d[18] = (uint32_t)smac;
            d[19] = (uint32_t)((smac >> 32) & 0xffffu) | (uint32_t)((dmac & 0xffffu) << 16);
            d[20] = (uint32_t)(dmac >> 16);
            d[21] = (uint32_t)id0;
            d[22] = ci[0];
            d[23] = (uint32_t)samp_tag;                                         // sampling: last non-zero
#pragma unroll
            for (int k = 1; k < 12; k++) d[23 + k] = ci[k];
            d[24] = (d[24] & 0xff00ffffu) | ((uint32_t)(dscp_tag & 0xffu) << 16);   // dscp: last non-zero
            d[35] = 0;
        }
        // ---- compaction: rank among the emitting lanes, one atomic per wave
        const unsigned long long mask = __ballot(emit);
        const uint32_t n_emit = (uint32_t)__popcll(mask);
        unsigned long long pos0 = base;
        if (FILTER) {
            if (lane == 0 && n_emit) pos0 = aadd(&t.ctr->n_out, (unsigned long long)n_emit);
            pos0 = __shfl(pos0, 0);
        }
        __builtin_amdgcn_wave_barrier();                                        // every lane has read its LDS rows
        if (emit) {
            const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            uint4* w = &W.hot[rank][0];
#pragma unroll
            for (int k = 0; k < 9; k++) w[k] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
        }
        __builtin_amdgcn_wave_barrier();
        uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + pos0 * kRecordBytes);
        const uint4* w = &W.hot[0][0];
        for (uint32_t c = lane; c < n_emit * 9; c += 64) o[c] = w[c];
        __builtin_amdgcn_wave_barrier();                                        // window free for the next 64 slots
    }
}

// ------------------------------------------------------------------
// Finalize: the last launch of every ingest call. For each slot claimed since the previous finalize, the tag of id0 is
// the sequence number of the flow's first record (account.go:95 stores it whole): copy that record's dwords 21..34
// from the batch into the slot — plain stores, this lane is the only writer, every fold kernel of the call is done.
// ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(TableView t, const void* __restrict__ recs, uint64_t n, uint32_t seq_base32) {
    const uint64_t from = ald(&t.ctr->n_finalized), to = t.ctr->n_live;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < to; i += stride) {
        const uint32_t idx = t.live_list[i];
        SlotHot* H = &t.hot[idx];
        const uint64_t id0 = H->id0;
        const uint64_t ri = (uint64_t)(uint32_t)(~(uint32_t)(id0 >> 32) - seq_base32);
        if ((uint32_t)(id0 >> 32) == 0 || ri >= n) continue;   // claimed, but no record of this batch's folded range is its first
        const uint4* rp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + ri * kRecordBytes);
        const uint4 c5 = rp[5], c6 = rp[6], c7 = rp[7], c8 = rp[8];     // record dwords 20..35
        uint4* cw = reinterpret_cast<uint4*>(NF_DENSE_AT(i, idx));
        cw[1] = make_uint4(c5.z, c6.x, c6.y, c6.z & 0x0000ffffu);       // dwords 22, 24, 25, 26 (pad2 cleared)
        cw[2] = make_uint4(c6.w, c7.x, c7.y, c7.z);                     // 27..30
        cw[3] = make_uint4(c7.w, c8.x, c8.y, c8.z);                     // 31..34
        reinterpret_cast<uint32_t*>(&H->id0)[0] = c5.y;                 // dword 21: if_index_first_seen
    }
    // The last block to get here publishes "everything up to n_live is finalized" (every block has read n_finalized by then:
    // its ticket comes after its loop) and resets the spill overflow tail of the two-pass fold. Round 2 spent two more
    // launches on this (k_finalize_done, a memset node): ~4 us of every small ingest call.
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int ticket = aadd(&t.ctr->fin_ticket, 1u);
        if (ticket == gridDim.x - 1) {
            ast(&t.ctr->n_finalized, (unsigned long long)to);
            ast(&t.ctr->fin_ticket, 0u);
            if (t.spill.ovf_tail) ast(t.spill.ovf_tail, 0u);
        }
    }
}

__global__ void k_reset_after_evict(DevCounters* c, int n_out_is_n_live) {
    if (n_out_is_n_live) c->n_out = c->n_live;
    c->n_live = 0;
    c->n_finalized = 0;
    c->aborted = 0;
    c->max_probe = 0;
}

// ------------------------------------------------------------------
// Optimistic fold (nfagg_api.hip): raw slot copies, discard, first sequence numbers.
// One 16-byte chunk per lane: consecutive lanes copy consecutive chunks of one slot's lines.
// ------------------------------------------------------------------
template <bool RESTORE>
__global__ __launch_bounds__(256) void k_snapshot(TableView t, uint64_t n, uint4* __restrict__ snap) {
    const int per = t.aux ? 28 : 12;                       // 16-byte chunks per slot: hot 8 + cold 4 (+ aux 16)
    const uint64_t total = n * (uint64_t)per, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint64_t i = g / per; const int c = (int)(g % per);
        const uint32_t idx = t.live_list[i];
        uint4* p; uint4* q;
        if (c < 8) { p = reinterpret_cast<uint4*>(&t.hot[idx]) + c; q = snap + i * 8 + c; }
        else if (c < 12) { p = reinterpret_cast<uint4*>(&t.cold[idx]) + (c - 8); q = snap + n * 8 + i * 4 + (c - 8); }
        else { p = reinterpret_cast<uint4*>(&t.aux[idx]) + (c - 12); q = snap + n * 12 + i * 16 + (c - 12); }
        if (RESTORE) *p = *q; else *q = *p;
    }
}

// Give the slots live_list[from..to) back: an empty tag is all it takes (whoever claims a slot re-initialises it).
__global__ __launch_bounds__(256) void k_discard(TableView t, uint64_t from, uint64_t to) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < to; i += stride)
        t.hot[t.live_list[i]].tag = 0;
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

// Default kernel by batch size, measured on configs[1]'s stream, per call (round 3: profiles/r03_batch_size_sweep.txt; round 2:
// profiles/r02_batch_size_sweep.txt; round 1: profiles/r01e_batch_size_crossover.txt):
//   below 6 144 records the direct kernel (one record per lane, HBM atomics; no LDS cache to set up and flush);
//   below 384 Ki records (768 Ki in round 2) the single-pass LDS-cached kernel: 0.049 ms per 65 536 records against 0.063 ms for the launches
//     of the two-pass fold, 0.130 against 0.120 ms at 256 Ki;
//   from there the two-pass partitioned fold: 0.17 against 0.24 ms at 512 Ki, 0.26 against 0.41 ms at 1 Mi, 0.70 against 1.29 ms at
//     4 Mi (partitions scaled to the batch and a cheaper flush moved the crossover from 768 Ki to ~300 Ki this round).
constexpr uint64_t kDirectMaxBatch = 6144;
constexpr uint64_t kPartMinBatch = 3u << 17;   // 384 Ki (round 3: 0.114 against 0.128 ms at 256 Ki, 0.18 against 0.23 at 512 Ki)
constexpr uint64_t kDedupCachedMinBatch = 1u << 16;
static bool takes_two_pass(int variant, uint64_t n) { return (variant >= 8 && variant <= 11) || variant == 17 || (variant >= 20 && variant <= 28) || ((variant == 0 || variant == 30) && n >= kPartMinBatch); }
// Shipping variants: 0 (by batch size), 1 direct, 3/4/5/7 geometries of the single-pass cached kernel, 10/11 two-pass
// (with / without the admission filter). 6/8/9 are the phase-timing builds and exist only in libnfagg_diag.so (-DNFAGG_DIAG).
bool ingest_variant_supported(int variant) {
#ifdef NFAGG_DIAG
    if (variant == 6 || variant == 8 || variant == 9 || (variant >= 13 && variant <= 15) || (variant >= 20 && variant <= 28)) return true;
#endif
    return variant == 0 || variant == 1 || variant == 3 || variant == 4 || variant == 5 || variant == 7 || variant == 10 || variant == 11 || variant == 12 ||
           variant == 17 ||   // 17: the two-pass fold always, pass 1 without its barriers (nfagg_ingest_part.hip k_pass1_free)
           variant == 16 ||   // 16 (kernel-dedup mode, tests): the cached passes always, the partition pass always sorts its items first
           variant == 30;     // 30 (tests): everything as 0, but nfagg_account always takes its kernel chain — the fallback of the epochs-found-first path
}
static bool takes_direct(int variant, uint64_t n, uint32_t sketch_flags) {
    return variant == 1 || ((variant == 0 || variant == 30) && n < kDirectMaxBatch && sketch_flags == 0);   // with sketches on, the cached kernel fuses them: one launch
}
static bool dedup_takes_cached(int variant, uint64_t n) { return !(variant == 1 || (variant != 10 && (variant < 12 || variant > 16) && n < kDedupCachedMinBatch)); }
bool ingest_needs_spill(int mode, int variant, uint64_t n) { return mode == 0 ? takes_two_pass(variant, n) : dedup_takes_cached(variant, n); }
bool ingest_fuses_sketches(int mode, int variant, uint64_t n, uint32_t sketch_flags) {
    if (mode == 1) return dedup_takes_cached(variant, n) && !(variant >= 13 && variant <= 15);   // the partition pass's flushes feed them (nfagg_dedup_cached.hip); 13-15: its timing ablations
    return mode == 0 && !takes_direct(variant, n, sketch_flags) && variant != 6 && variant != 8 && variant != 9 && (variant < 20 || variant == 30);
}

hipError_t launch_ingest(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, uint64_t seq_base,
                         int mode, int variant, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (mode == 1) {   // NFAGG_MODE_KERNEL_DEDUP: LDS-cached passes; direct per-record passes for small batches (variant 1: always, 10: never)
        if (!dedup_takes_cached(variant, n)) return launch_ingest_dedup(t, d_records, n, seq_base, s);
        return launch_ingest_dedup_cached(t, sk, d_records, n, seq_base, variant, s);
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

size_t snapshot_bytes(const TableView& t, uint64_t n) { return (size_t)n * (t.aux ? 448 : 192); }

hipError_t launch_snapshot(const TableView& t, uint64_t n, void* d_snap, bool restore, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int grid = grid_for(n * (t.aux ? 28 : 12), 256, 256 * 16);
    (void)hipGetLastError();
    if (restore) hipLaunchKernelGGL(k_snapshot<true>, dim3(grid), dim3(256), 0, s, t, n, (uint4*)d_snap);
    else hipLaunchKernelGGL(k_snapshot<false>, dim3(grid), dim3(256), 0, s, t, n, (uint4*)d_snap);
    return hipGetLastError();
}

hipError_t launch_discard(const TableView& t, uint64_t from, uint64_t to, hipStream_t s) {
    if (to <= from) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_discard, dim3(grid_for(to - from, 256, 256 * 8)), dim3(256), 0, s, t, from, to);
    return hipGetLastError();
}

hipError_t launch_first_seqs(const TableView& t, uint64_t from, uint64_t to, uint32_t* d_out, hipStream_t s) {
    if (to <= from) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_first_seqs, dim3(grid_for(to - from, 256, 256 * 8)), dim3(256), 0, s, t, from, to, d_out);
    return hipGetLastError();
}

hipError_t launch_finalize(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base, hipStream_t s) {
    (void)hipGetLastError();
    // the number of new slots is only known on the device: a fixed grid strides over [n_finalized, n_live)
    const int grid = n >= (1u << 20) ? 1024 : (n >= (1u << 14) ? 64 : 4);
    hipLaunchKernelGGL(k_finalize, dim3(grid), dim3(256), 0, s, t, d_records, n, (uint32_t)seq_base);
    return hipGetLastError();
}

hipError_t launch_reset_counters(const TableView& t, hipStream_t s) {
    (void)hipGetLastError(); hipLaunchKernelGGL(k_reset_after_evict, dim3(1), dim3(1), 0, s, t.ctr, 0);
    return hipGetLastError();
}

hipError_t launch_sort_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n, void* d_temp, size_t* temp_bytes, hipStream_t s) {
    return hipcub::DeviceRadixSort::SortKeys(d_temp, *temp_bytes, d_in, d_out, (int)n, 0, 32, s);
}

// force_filter: take the filtered kernel (positions from n_out) although no split is pending — the group's local-fold mode
// evicts only the flows the shard owns
hipError_t launch_evict_filtered(const TableView& t, uint64_t n_live, uint64_t seq_limit, void* d_out, hipStream_t s) {
    if (n_live) {
        (void)hipGetLastError();
        const dim3 grid(grid_for(n_live, 64 * kEvictWaves, 256 * 16)), block(64 * kEvictWaves);
        hipLaunchKernelGGL(k_evict<true>, grid, block, 0, s, t, n_live, seq_limit, d_out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    (void)hipGetLastError(); hipLaunchKernelGGL(k_reset_after_evict, dim3(1), dim3(1), 0, s, t.ctr, 0);
    return hipGetLastError();
}

hipError_t launch_evict(const TableView& t, uint64_t n_live, uint64_t seq_limit, void* d_out, hipStream_t s) {
    if (t.aux) return launch_evict_dedup(t, n_live, seq_limit, d_out, s);
    if (n_live) {
        (void)hipGetLastError();
        const dim3 grid(grid_for(n_live, 64 * kEvictWaves, 256 * 16)), block(64 * kEvictWaves);
        if (seq_limit == ~0ull) hipLaunchKernelGGL(k_evict<false>, grid, block, 0, s, t, n_live, seq_limit, d_out);
        else hipLaunchKernelGGL(k_evict<true>, grid, block, 0, s, t, n_live, seq_limit, d_out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    (void)hipGetLastError(); hipLaunchKernelGGL(k_reset_after_evict, dim3(1), dim3(1), 0, s, t.ctr, seq_limit == ~0ull ? 1 : 0);
    return hipGetLastError();
}

}  // namespace nfagg
