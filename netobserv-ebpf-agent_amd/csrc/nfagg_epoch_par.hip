// nfagg_epoch_par.hip — the evict-on-full loop of Accounter.Account (pkg/flow/account.go:81-96) WITHOUT its sequential chain: the
// epochs of a call are found first, then every complete epoch is folded on its own, all of them at once, with no flow table at all.
// What nfagg_account[_device] runs for calls of more than a few epochs (host side: nfagg_account_par.inc; the rule
// is pinned on the CPU by tests/test_epoch_boundaries.py).
//
// The loop is sequential only in WHERE its epochs end. With prev(i) = the index of the previous record of record i's flow in the
// call (-1: none), record i >= s starts a new flow in the epoch that began at record s exactly when prev(i) < s (first epoch of a
// call: and the flow is not live in the table), so the epoch ends at the record where the count of such records reaches
// max_entries + 1 — a prefix count over prev[], no table involved. Given the cuts, an epoch's eviction is the fold of every flow's
// records between two cuts (AccumulateBase, pkg/model/flow_content.go:28-61, over the records in arrival order), and ONE sort
// already brings them together: the call's records sorted by (key hash, index) hold each flow's records of one epoch as a
// contiguous SEGMENT in arrival order. So:
//
//   k_par_hash      sort key per record: (top 40 bits of the key hash) << 24 | index (a launch takes at most 2^24 records)
//   rocPRIM radix sort of the keys on their hash bits (stable: equal hash bits stay in index order — the array is sorted as 64-bit
//                   numbers)
//   k_par_links     prev(i) from the neighbours to the left in sorted order, FULL keys compared: flows that share their 40 hash
//                   bits (one pair in two calls at a million flows; any number when somebody crafts them) are told apart here and
//                   in the fold. Only when more than 4096 records of OTHER flows lie between a record and its previous occurrence
//                   (a cold flow sharing its hash bits with one of the hottest) does the call go to the kernel chain
//   k_par_live      first occurrences whose flow is live in the table: prev = -2 (not new in the first epoch of the call)
//   k_par_cuts      ONE workgroup streams prev[] once in blocks of 16 Ki records and walks the epochs over it (per step: 16 compares per wave,
//                   scalar mask counts, one 16-entry LDS exchange, one barrier); resumable: run in parts, the folds of a part's epochs beside the next part
//   k_par_rank_*    the rank of every new flow among its epoch's new flows in arrival order = its position in the epoch's eviction:
//                   one running count over tiles of the launch's records (exactly max_entries per epoch: checked)
//   k_par_segfold   one lane per segment of at most kSegShort records: the records gathered in arrival order and folded
//                   SEQUENTIALLY, literally as the reference does; the folded record goes straight to its place in the caller's buffer
//   k_par_segfold_long   one wave per longer segment (the hot flows): an order-free partial per lane, combined across the wave
//   k_par_huge_chunks / k_par_huge_combine   segments of more than 4096 records (long epochs): a wave per chunk of 2048, then the chunks' partials
//
// The first epoch of the call (it continues what the table holds) and the last, incomplete one (it stays live) go through the
// ordinary table fold; everything in between never touches a hash table.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "nfagg_device.h"

namespace nfagg {

constexpr int kParBlock = 256;
constexpr uint32_t kParNone = 0xffffffffu;
constexpr int kLinkSearch = 4096;                                     // records of OTHER flows with the same hash bits that k_par_links walks over before it gives up
constexpr int kIdxBits = 24;                                          // a launch takes at most 2^24 records: the index's share of a sort key
constexpr uint64_t kIdxMask = (1ull << kIdxBits) - 1ull, kHashMask = ~kIdxMask;   // ... and the key hash's: its top 40 bits
constexpr int kSegUnroll = 4;                                         // record gathers a lane of the wave folds keeps in flight
#ifndef NF_SEG_SHORT                                                  // (the three thresholds can be set on the compiler's command line: tools/gpu/r06_seg_sweep.sh)
#define NF_SEG_SHORT 16
#endif
#ifndef NF_SEG_HUGE
#define NF_SEG_HUGE 4096
#endif
#ifndef NF_HUGE_CHUNK
#define NF_HUGE_CHUNK 2048
#endif
constexpr uint32_t kSegShort = NF_SEG_SHORT;                                    // records per segment the one-lane fold takes
constexpr uint32_t kSegHuge = NF_SEG_HUGE;                                   // positions per segment beyond which it is folded in CHUNKS, a wave per chunk (one wave for the whole segment up to here)
constexpr uint32_t kHugeChunk = NF_HUGE_CHUNK;                                 // positions per chunk
constexpr uint32_t kHugeCap = (1u << 24) / (kSegHuge + 1u) + 4097u;                                   // entries of the list of such segments (disjoint runs of > kSegHuge positions: <= 2^24 / 4097 per launch)
constexpr uint32_t kHugeChunkCap = (1u << 24) / kHugeChunk + kHugeCap;   // ... and of their chunks: every segment ends with one partly filled chunk

static inline int par_grid(uint64_t n, int per_block = kParBlock, int cap = 1 << 20) {
    uint64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > (uint64_t)cap) g = cap;
    return (int)g;
}

// The index a sort key carries. The empty asm is not decoration: hipcc (ROCm 7.2) sees `(key & 0xFFFFFF) * 144` as a 24-bit
// multiply, drops the mask as "not demanded", and then forms v_mad_u64_u32 — which DOES read the upper eight bits: the record
// address was computed from the key's low 32 bits, hash bits included (k_par_links faulted on its first launch; ISA and the
// stand-alone reproduction: profiles/r05_mul24_miscompile.txt, tools/gpu/links_probe.hip). An opaque register keeps the mask.
NF_DEV uint32_t key_index(uint64_t key) {
    uint32_t i = (uint32_t)(key & kIdxMask);
    asm volatile("" : "+v"(i));
    return i;
}

NF_DEV void par_key(const void* recs, uint64_t i, uint64_t w[5]) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + i * kRecordBytes);
    const uint4 a = p[0], b = p[1], c = p[2];
    w[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); w[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    w[2] = (uint64_t)b.x | ((uint64_t)b.y << 32); w[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
    w[4] = ((uint64_t)c.x | ((uint64_t)c.y << 32)) & 0x00FFFFFFFFFFFFFFull;      // key byte 39: Go's blank field, not part of the key
}
NF_DEV bool par_same_key(const uint64_t a[5], const uint64_t b[5]) {
    return ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3]) | (a[4] ^ b[4])) == 0;
}

__global__ __launch_bounds__(kParBlock) void k_par_hash(const void* __restrict__ recs, uint64_t n, uint64_t* __restrict__ keys) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t w[5];
        par_key(recs, i, w);
        keys[i] = (key_hash(w) & kHashMask) | (uint64_t)i;
    }
}

// Sorted position p holds record i. Its previous occurrence is the nearest position to the left with the same hash bits AND the
// same key (equal hash bits are in index order). Nearly always that is position p - 1 or nothing; flows that share their hash bits
// interleave, and the search walks over the other flows' records (at most kLinkSearch of them: beyond that *overflow is raised and
// the call takes the kernel chain).
// Every lane gathers ITS record's key once; the key of position p - 1 comes from the lane to the left (lane 0 of a wave loads
// it): one gather per record instead of two — the second one was not served from the cache (299 B fetched per record for two
// 48-byte keys, profiles/r05_cache_max_flows_5000_pmc_summary.md of the round's first evidence pass).
NF_DEV uint64_t shfl_up1_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, 1), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__global__ __launch_bounds__(kParBlock) void k_par_links(const void* __restrict__ recs, const uint64_t* __restrict__ ks, uint64_t n,
                                                         int32_t* __restrict__ prev, uint32_t* __restrict__ overflow) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += stride) {      // (every lane of a wave goes round together: shuffles)
        const uint64_t p = base + threadIdx.x;
        const bool valid = p < n;
        const uint64_t key = valid ? ks[p] : 0ull;
        const uint32_t i = key_index(key);
        const uint64_t hb = key & kHashMask;
        uint64_t a[5] = {0, 0, 0, 0, 0};
        if (valid) par_key(recs, i, a);
        // position p - 1: hash bits, index and key from the lane to the left
        uint64_t hb_l = shfl_up1_u64(hb);
        uint32_t j_l = (uint32_t)__shfl_up((int)i, 1);
        uint64_t b[5];
#pragma unroll
        for (int k = 0; k < 5; k++) b[k] = shfl_up1_u64(a[k]);
        if (lane == 0 && valid && p > 0) {                            // the wave's first position: its left neighbour belongs to another wave
            const uint64_t k2 = ks[p - 1];
            hb_l = k2 & kHashMask; j_l = key_index(k2);
            if (hb_l == hb) par_key(recs, j_l, b);
        }
        if (!valid) continue;
        int32_t pv = -1;
        if (p > 0 && hb_l == hb) {
            if (par_same_key(a, b)) {
                pv = (int32_t)j_l;
            } else {                                                  // another flow with these hash bits: walk on to the left
                uint64_t q = p - 1;
                int looked = 1;
                while (q > 0) {
                    const uint64_t k2 = ks[--q];
                    if ((k2 & kHashMask) != hb) break;
                    if (++looked > kLinkSearch) { atomicExch(overflow, 1u); break; }
                    const uint32_t j = key_index(k2);
                    par_key(recs, j, b);
                    if (par_same_key(a, b)) { pv = (int32_t)j; break; }
                }
            }
        }
        prev[i] = pv;
    }
}

// c.entries[key] without inserting (the table is quiescent: no fold is running). Probing as find_or_claim does: a slot whose tag
// is not of this epoch ends the chain.
NF_DEV bool par_is_live(const TableView& t, const uint64_t w[5], uint64_t h) {
    const uint64_t ready = tag_ready(t, h);
    uint64_t idx = h & t.mask;
    for (uint64_t probes = 0; probes <= t.mask; probes++) {
        const SlotHot* s = &t.hot[idx];
        const uint64_t tag = ald(&s->tag);
        if (tag_is_free(t, tag)) return false;
        if (tag == ready) {
            bool eq = true;
#pragma unroll
            for (int k = 0; k < 5; k++) eq &= (ald(&s->key[k]) == w[k]);
            if (eq) return true;
        }
        idx = (idx + 1) & t.mask;
    }
    return false;
}

__global__ __launch_bounds__(kParBlock) void k_par_live(TableView t, const void* __restrict__ recs, int32_t* __restrict__ prev, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (prev[i] != -1) continue;
        uint64_t w[5];
        par_key(recs, i, w);
        if (par_is_live(t, w, key_hash(w))) prev[i] = -2;
    }
}

// ---- the cut walk ---------------------------------------------------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes of a wave with DPP: inside each row of 16 (row_shr 1, 2, 4, 8, zeroes shifted in), then
// row 0's total into row 1 and row 2's into row 3 (row_bcast:15), then the total of rows 0-1 into rows 2-3 (row_bcast:31). Every
// lane must be active.
template <int CTRL, int ROW_MASK>
NF_DEV uint32_t dpp_add_u32(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
NF_DEV uint32_t wave_scan_u32(uint32_t v) {
    v = dpp_add_u32<0x111, 0xf>(v);
    v = dpp_add_u32<0x112, 0xf>(v);
    v = dpp_add_u32<0x114, 0xf>(v);
    v = dpp_add_u32<0x118, 0xf>(v);
    v = dpp_add_u32<0x142, 0xa>(v);
    v = dpp_add_u32<0x143, 0xc>(v);
    return v;
}

// ONE workgroup. prev[] is streamed ONCE in aligned blocks of kCutSpan records (a lane holds kCutPer consecutive ones in
// registers; FOUR blocks rotate through four register sets, so the loads of the next three are in flight while one is worked on —
// a rotation by register copies would wait for the youngest load at every block), and the epochs are walked over the resident
// block: from the epoch's first record s on, the records that start a new flow in it — prev < s; in the first epoch of the call:
// prev == -1 (a flow the table holds, prev == -2, is no new entry) — are counted; the one that sees exactly `budget` new flows
// before it (max_entries; less what is live, for the first epoch) finds the map full (account.go:85): it ends the epoch and opens
// the next, which is walked over the SAME registers. A step is a block that is counted through or an epoch that ends: one
// vector compare per register and scalar counting of the lane masks (cut_block), the 16 wave totals through LDS, one barrier — and a
// second one only when the epoch ends in the block (the lane that holds the cut tells the others). Records before s are dead for
// good (s only grows): waves wholly before it skip their counting, the lane that holds s overwrites its dead records with "never new".
// One CU evaluates 16 Ki records per step: the walk is bound by the instructions of a step, not by the 4 bytes per record.
// (Round 4's walk re-read every epoch from its first record on, 16 Ki records per step with a 256-entry scan by one lane: 5.8 us
// per epoch, 3.2 ms per 8 M-record call with 557 epochs; this kernel's first form — six instructions per record, the blocks rotated
// by register copies — 2.67 ms: profiles/r05_account_par_kernel_stats_first.csv.)
// cuts[k] = the record that ends epoch k; at most max_cuts of them. ctl[0] = how many were found; ctl[3] = the new flows of the
// epoch in progress when the records ended (what the table holds once that epoch's records are folded; meaningless when the walk
// stopped at max_cuts).
constexpr int kCutBlock = 1024;
constexpr int kCutPer = 16;
constexpr uint64_t kCutSpan = (uint64_t)kCutBlock * kCutPer;
constexpr int32_t kCutNever = 0x7fffffff;

// A wave's share of a block: 1024 consecutive records as four ROWS of 256; lane l holds records 4 l .. 4 l + 3 of every row (one
// 16-byte load per row: an instruction covers 1 KiB of consecutive addresses — with 16 consecutive records per lane an instruction
// touched 32 lines for 16 bytes each and every line four times: 1.48 ms per 8 M-record call against this layout's figure in §4.11b).
constexpr int kCutRows = kCutPer / 4;
// The loads are UNCONDITIONAL: prev[] is padded with kCutNever up to four blocks beyond the block that holds the call's last
// record (par_prev_entries(); the host fills the pad). With a bounds branch around them the compiler closed every load with
// s_waitcnt vmcnt(0) — nothing was in flight while a block was walked, and the walk was bound by four memory round trips per
// block (1.32 ms per 8 M-record call, whatever its arithmetic: profiles/r05_cut_walk_latency.txt).
NF_DEV void cut_load(const int32_t* __restrict__ prev, uint64_t wave_base, int lane, int32_t v[kCutPer]) {
#pragma unroll
    for (int r = 0; r < kCutRows; r++) {
        const int4 x = *reinterpret_cast<const int4*>(prev + wave_base + (uint64_t)r * 256 + (uint64_t)lane * 4);
        v[4 * r] = x.x; v[4 * r + 1] = x.y; v[4 * r + 2] = x.z; v[4 * r + 3] = x.w;
    }
}

NF_DEV void cut_arrived(const int32_t v[kCutPer]) {
    asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                       "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
}

struct CutState {
    uint32_t s, k, before, par, budget, max_entries, max_cuts;
    bool first;
#ifdef NFAGG_DIAG
    unsigned long long ph[8], tp;     // libnfagg_diag.so: cycles of wave 0 per phase (ctl[16..21], in Ki cycles): waiting for a block's loads,
                                      // counting, barrier + totals, finding the cut + second barrier, bookkeeping of a cut, issuing loads
#endif
};
#ifdef NFAGG_DIAG
#define NF_CUT_TICK(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); st.ph[k] += tn_ - st.tp; st.tp = tn_; } while (0)
#else
#define NF_CUT_TICK(k)
#endif

// The epochs over one resident block (cur[]: this lane's records wave_base + 256 r + 4 lane + c). Returns when the block is
// counted through or max_cuts are found. Record order inside the wave is row-major (row, lane, component).
// The counting is SCALAR arithmetic on lane masks: one vector compare per register (16 per wave and step), its 64-bit result ANDed
// with the row's live lanes (the groups of four wholly before s are dead) and counted by the scalar unit — the wave's total never
// enters a vector register. (The first forms of this walk counted per lane — compare, add — and scanned the counts with DPP in every
// step, ~135 vector instructions per wave and step on one compute unit: 1.32 ms per 8 M-record call, profiles/r05_cut_walk_counters.txt.)
// Only the step that finds the epoch's end looks for a position, and only in the one wave and row that hold it.
template <bool FIRST>
NF_DEV bool cut_is_new(int32_t v, int32_t s32) { return FIRST ? v == -1 : v < s32; }

// the new records of the wave's four rows, counted by the scalar unit from the compare masks (s: a scalar)
template <bool FIRST>
NF_DEV void cut_count(const int32_t cur[kCutPer], uint32_t wave_base, uint32_t s, uint32_t row_tot[kCutRows], uint64_t row_live[kCutRows]) {
    const int32_t s32 = (int32_t)s;
    if (s <= wave_base) {                                             // the usual case: the epoch began before this wave's records, all of them count
        // (all sixteen compares first, each into scalar registers of its own: with one compare -> count pair after the other
        // every pair waits for the vector pipeline to hand over VCC)
        uint64_t m[kCutPer];
#pragma unroll
        for (int k = 0; k < kCutPer; k++) m[k] = __ballot(cut_is_new<FIRST>(cur[k], s32));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < kCutRows; r++) {
            row_tot[r] = (uint32_t)(__popcll(m[4 * r]) + __popcll(m[4 * r + 1]) + __popcll(m[4 * r + 2]) + __popcll(m[4 * r + 3]));
            row_live[r] = ~0ull;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < kCutRows; r++) {
        // lane l's four records of row r lie wholly before s when 4 l + 4 <= s - row_base (the four that hold s have their dead
        // records overwritten when the cut is made)
        const uint32_t row_base = wave_base + (uint32_t)r * 256;
        const uint32_t dead = s > row_base ? (s - row_base) >> 2 : 0u;
        const uint64_t livem = dead >= 64u ? 0ull : ~0ull << dead;
        row_live[r] = livem;
        uint32_t rc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) rc += (uint32_t)__popcll(__ballot(cut_is_new<FIRST>(cur[4 * r + q], s32)) & livem);
        row_tot[r] = rc;
    }
}

// the wave and row that hold new record number `target` (from 0, in record order) of the row: its index in the call
template <bool FIRST>
NF_DEV void cut_find(const int32_t cur[kCutPer], int r, uint32_t row_first, uint32_t s, uint64_t livem, uint32_t target, int lane, uint32_t* out) {
    const int32_t s32 = (int32_t)s;
    uint32_t cr = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) cr += cut_is_new<FIRST>(cur[4 * r + q], s32) ? 1u : 0u;
    if (!((livem >> lane) & 1ull)) cr = 0;
    const uint32_t incl = wave_scan_u32(cr);
    const uint32_t need = target - (incl - cr);                       // which of this lane's new records of the row it is (wraps when it is elsewhere)
    if (need < cr) {
        uint32_t seen = 0, at = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (cut_is_new<FIRST>(cur[4 * r + q], s32)) { if (seen == need) at = (uint32_t)q; seen++; }
        }
        *out = row_first + (uint32_t)lane * 4 + at;
    }
}

NF_DEV void cut_block(int32_t cur[kCutPer], uint32_t wave_base_v, CutState& st, uint32_t (*wtot)[kCutBlock / 64], uint32_t* fnd,
                      uint32_t* __restrict__ cuts) {
    constexpr int kWaves = kCutBlock / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // everything the steps decide on is the same in every lane of a wave: kept in scalar registers, so that the masks are counted
    // and the branches taken by the scalar unit
    const uint32_t wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_base_v);
    const uint32_t lane_base = wave_base + (uint32_t)lane * 4;        // first record of this lane's row 0
    const int wv_u = __builtin_amdgcn_readfirstlane(wv);
    while (st.k < st.max_cuts) {
        uint32_t row_tot[kCutRows] = {0, 0, 0, 0};
        uint64_t row_live[kCutRows] = {0, 0, 0, 0};
        const uint32_t s_u = st.s;
        const bool first_u = st.first;
        if (wave_base + 64u * kCutPer > s_u) {                        // some record of this wave is at or after s
            if (first_u) cut_count<true>(cur, wave_base, s_u, row_tot, row_live);
            else cut_count<false>(cur, wave_base, s_u, row_tot, row_live);
        }
        const uint32_t wave_total = row_tot[0] + row_tot[1] + row_tot[2] + row_tot[3];
        NF_CUT_TICK(1);
        if (lane == 0) wtot[st.par][wv] = wave_total;
        __syncthreads();
        NF_CUT_TICK(6);
        // the 16 wave totals: one LDS read per lane and a scan inside the first row of 16 lanes
        uint32_t pre = lane < kWaves ? wtot[st.par][lane] : 0u;
        pre = dpp_add_u32<0x111, 0xf>(pre); pre = dpp_add_u32<0x112, 0xf>(pre); pre = dpp_add_u32<0x114, 0xf>(pre); pre = dpp_add_u32<0x118, 0xf>(pre);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)pre, kWaves - 1);
        const uint32_t woff = wv_u ? (uint32_t)__builtin_amdgcn_readlane((int)pre, wv_u - 1) : 0u;
        NF_CUT_TICK(2);
        if (st.before + total <= st.budget) {                         // the epoch goes on beyond this block
            st.before += total;
            st.par ^= 1u;
            return;
        }
        // entry number budget + 1 is in this block: new record number `target` (from 0) of exactly one wave — the others see a
        // number beyond their total (before it: not reached; after it: the subtraction wraps)
        uint32_t target = st.budget - st.before - woff;
        if (target < wave_total) {
#pragma unroll
            for (int r = 0; r < kCutRows; r++) {
                if (target < row_tot[r]) {                            // ... of exactly one row of it
                    if (first_u) cut_find<true>(cur, r, wave_base + (uint32_t)r * 256, s_u, row_live[r], target, lane, &fnd[st.par]);
                    else cut_find<false>(cur, r, wave_base + (uint32_t)r * 256, s_u, row_live[r], target, lane, &fnd[st.par]);
                    target = 0xffffffffu;                             // found: no later row
                } else {
                    target -= row_tot[r];
                }
            }
        }
        __syncthreads();
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)fnd[st.par]);   // (the same in every lane: kept scalar, like all of st)
        NF_CUT_TICK(3);
        if (tid == 0) cuts[st.k] = f;
        st.k++; st.s = f; st.before = 0; st.first = false; st.budget = st.max_entries;   // the next epoch is walked over the same block
        st.par ^= 1u;
        if (f - wave_base < 64u * kCutPer) {                          // (scalar) this wave holds the new s ...
#pragma unroll
            for (int r = 0; r < kCutRows; r++) {
                if ((f - wave_base) >> 8 == (uint32_t)r) {            // (scalar) ... in this row: of the four records around it, those before it never count again
                    const uint32_t b4 = lane_base + (uint32_t)r * 256;
                    if (f >= b4 && f < b4 + 4) {
#pragma unroll
                        for (int q = 0; q < 4; q++) if (b4 + (uint32_t)q < f) cur[4 * r + q] = kCutNever;
                    }
                }
            }
        }
        NF_CUT_TICK(4);
    }
}

// The walk is RESUMABLE: a launch covers the blocks [b_begin, b_end) and leaves its state (the epoch in progress: where it began,
// how many new flows it has seen, which cut comes next) in ctl[8..12]; a launch with b_begin > 0 picks it up. The host runs the walk
// in a few such parts and hands the epochs each part has completed to the segment folds while the next part walks on: the walk
// keeps ONE compute unit busy, the folds the other 255 (nfagg_account_par.inc). Nothing else changes: a block boundary is an
// ordinary place for an epoch to go on across (the records of a block before the epoch's start were dead anyway).
__global__ __launch_bounds__(kCutBlock) void k_par_cuts(const int32_t* __restrict__ prev, uint64_t n, uint32_t max_entries, uint32_t live0,
                                                        uint32_t* __restrict__ cuts, uint32_t max_cuts, uint32_t* __restrict__ ctl,
                                                        uint64_t b_begin, uint64_t b_end, uint32_t* __restrict__ host_cuts, uint32_t* __restrict__ host_state) {
    __shared__ uint32_t wtot[2][kCutBlock / 64];                      // double-buffered by step parity: one barrier per step
    __shared__ uint32_t fnd[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the walk is what the rest of the launch waits for, and waves of the segment folds may share this compute unit: its
    // instructions go first
    __builtin_amdgcn_s_setprio(3);
    CutState st;
    st.par = 0;
    if (b_begin == 0) {
        st.s = 0; st.k = 0; st.before = 0; st.first = true;           // the first epoch: the one the table's live flows belong to (it may end at record 0)
        st.budget = live0 >= max_entries ? 0u : max_entries - live0;
    } else {                                                          // (read by every lane before lane 0 rewrites them behind the walk's barriers)
        st.s = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[8]); st.k = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[9]);
        st.before = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[10]); st.first = __builtin_amdgcn_readfirstlane((int)ctl[11]) != 0;
        st.budget = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[12]);
    }
    st.max_entries = max_entries; st.max_cuts = max_cuts;
    const uint32_t k_begin = st.k;                                    // the cuts this part finds: [k_begin, st.k)
#ifdef NFAGG_DIAG
    for (int k = 0; k < 8; k++) st.ph[k] = 0;
    st.tp = __builtin_readcyclecounter();
#endif
    __syncthreads();
    const uint64_t wave_off = (uint64_t)wv * 64u * kCutPer;          // this wave's first record, relative to the block
    int32_t A[kCutPer], B[kCutPer], C[kCutPer], D[kCutPer];
    // b_end - b_begin is a multiple of four (the host cuts the parts so; prev[] is padded with "never new" to a multiple of four
    // blocks and four more): the loop body has no branch around a load, and the loads are issued in the order their registers are
    // used in — what the compiler needs to let twelve of them fly while the oldest four are worked on.
    cut_load(prev, b_begin * kCutSpan + wave_off, lane, A);
    __builtin_amdgcn_sched_barrier(0);
    cut_load(prev, (b_begin + 1) * kCutSpan + wave_off, lane, B);
    __builtin_amdgcn_sched_barrier(0);
    cut_load(prev, (b_begin + 2) * kCutSpan + wave_off, lane, C);
    __builtin_amdgcn_sched_barrier(0);
    cut_load(prev, (b_begin + 3) * kCutSpan + wave_off, lane, D);
    __builtin_amdgcn_sched_barrier(0);
    // cut_arrived: an empty statement that READS the block's registers, outside any loop: the compiler waits there for exactly
    // these sixteen (vmcnt(12): the twelve younger loads stay in flight). Without it the first use is inside cut_block's loop,
    // and the compiler drains every load before a loop that reads loaded registers (vmcnt(0) in the loop's preheader).
#define NF_CUT_BLOCK(BUF, BIDX)                                                                                                      \
    cut_arrived(BUF);                                                                                                                \
    NF_CUT_TICK(0);                                                                                                                  \
    cut_block(BUF, (uint32_t)((BIDX) * kCutSpan + wave_off), st, wtot, fnd, cuts);                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                               \
    cut_load(prev, ((BIDX) + 4) * kCutSpan + wave_off, lane, BUF);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                               \
    NF_CUT_TICK(5);
    for (uint64_t b0 = b_begin; b0 < b_end && st.k < st.max_cuts; b0 += 4) {
        NF_CUT_BLOCK(A, b0)
        NF_CUT_BLOCK(B, b0 + 1)
        NF_CUT_BLOCK(C, b0 + 2)
        NF_CUT_BLOCK(D, b0 + 3)
    }
#undef NF_CUT_BLOCK
#ifdef NFAGG_DIAG
    if (tid == 0) for (int k = 0; k < 8; k++) atomicAdd(&ctl[16 + k], (uint32_t)(st.ph[k] >> 10));
    if (tid == kCutBlock - 64) for (int k = 0; k < 8; k++) atomicAdd(&ctl[24 + k], (uint32_t)(st.ph[k] >> 10));
#endif
    if (tid == 0) {
        ctl[0] = st.k; ctl[3] = st.before;
        ctl[8] = st.s; ctl[9] = st.k; ctl[10] = st.before; ctl[11] = st.first ? 1u : 0u; ctl[12] = st.budget;
    }
    // What the HOST needs of this part — the cuts it found, how far the walk is — goes straight into the host's pinned list: the
    // parts of a walk are enqueued back to back with an event behind each, the host waits for the events and hands every part's
    // epochs to the folds while the walk goes on; nothing of the walk waits for the host (it was a copy and a round trip per part:
    // ~60 us of an idle walk each, profiles/r06x_walk_ungated.txt). host_state: a block of its own per part — the host reads part
    // p's while part p + 1 runs.
    if (host_cuts) {
        __threadfence();                                              // lane 0's cuts[] stores, for the other lanes (read back past the vector cache)
        __syncthreads();
        for (uint32_t k = k_begin + (uint32_t)tid; k < st.k; k += kCutBlock)
            host_cuts[k] = __atomic_load_n(&cuts[k], __ATOMIC_RELAXED);
        if (tid == 0) { host_state[0] = st.k; host_state[1] = st.before; host_state[2] = __atomic_load_n(&ctl[1], __ATOMIC_RELAXED); host_state[3] = k_begin; }
    }
}

// The positions of the middle epochs' new flows. Epoch t of the middle = records [cuts[t], cuts[t + 1]); a record i of it starts a new
// flow in it when prev[i] < cuts[t] (a HEAD). pos[i] = t * max_entries + the number of heads of the epoch before record i, for the
// heads; kParNone for the other records. The eviction of the epoch is what the heads' segments fold to, in that order (Go's map
// order is random: any order is the reference's). Every complete epoch holds exactly max_entries heads — that is how its end was
// found — so the position is ONE running count over all the records of the launch: t_lo * max_entries + the heads in
// [cuts[t_lo], i). Three launches over tiles of kRankTile records, whatever the epochs' lengths (round 5 ranked every epoch with one
// workgroup of its own: right for the 14 k-record epochs of CACHE_MAX_FLOWS = 5000, a serial walk of 2400 steps per epoch at
// 100 000 — pkg/flow/tracer_map_bench_test.go:64-111 brackets 1 k / 10 k / 100 k):
//   k_par_rank_count   heads per tile
//   k_par_rank_scan    one workgroup: exclusive prefix of the tile counts; the total must be (t_hi - t_lo) * max_entries
//   k_par_rank_write   pos[]; and the check that makes "exactly max_entries per epoch" a fact and not an assumption: the first
//                      record of every epoch t is a head (prev[i] < i always) and its position must be exactly t * max_entries — by
//                      induction every epoch before it holds max_entries heads; the total covers the last one. *bad otherwise.
constexpr int kRankPer = 8;
constexpr uint32_t kRankTile = kParBlock * kRankPer;

// the middle epoch that holds record i: the largest t in [t_lo, t_hi) with cuts[t] <= i (cuts[t_lo] <= i < cuts[t_hi])
NF_DEV uint32_t par_epoch_of(const uint32_t* __restrict__ cuts, uint32_t t_lo, uint32_t t_hi, uint32_t i) {
    uint32_t lo = t_lo, hi = t_hi;
    while (hi - lo > 1u) { const uint32_t mid = lo + ((hi - lo) >> 1); if (cuts[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// Is record i a head, and of which epoch? The tile's first and last epochs bracket the search (most tiles lie inside one epoch).
struct RankTile { uint32_t t_first, t_last; };
NF_DEV RankTile rank_tile_epochs(const uint32_t* __restrict__ cuts, uint32_t t_lo, uint32_t t_hi, uint32_t base, uint32_t i_hi, uint32_t* sh) {
    if (threadIdx.x == 0) {
        const uint32_t last = base + kRankTile - 1u < i_hi ? base + kRankTile - 1u : i_hi - 1u;
        sh[0] = par_epoch_of(cuts, t_lo, t_hi, base);
        sh[1] = par_epoch_of(cuts, sh[0], t_hi, last);
    }
    __syncthreads();
    RankTile r; r.t_first = sh[0]; r.t_last = sh[1];
    return r;
}
NF_DEV bool rank_is_head(const int32_t* __restrict__ prev, const uint32_t* __restrict__ cuts, const RankTile& rt, uint32_t s_first, uint32_t i, uint32_t* t_out) {
    uint32_t t = rt.t_first, s = s_first;
    if (rt.t_last != rt.t_first) { t = par_epoch_of(cuts, rt.t_first, rt.t_last + 1u, i); s = cuts[t]; }
    *t_out = t;
    return (int64_t)prev[i] < (int64_t)s;
}

__global__ __launch_bounds__(kParBlock) void k_par_rank_count(const int32_t* __restrict__ prev, const uint32_t* __restrict__ cuts, uint32_t t_lo, uint32_t t_hi,
                                                              uint32_t i_lo, uint32_t i_hi, uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t sh[2], wsum[kParBlock / 64];
    const uint32_t base = i_lo + blockIdx.x * kRankTile;
    const RankTile rt = rank_tile_epochs(cuts, t_lo, t_hi, base, i_hi, sh);
    const uint32_t s_first = cuts[rt.t_first];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kRankPer; k++) {
        const uint32_t i = base + (uint32_t)k * kParBlock + threadIdx.x;
        uint32_t t;
        if (i < i_hi && rank_is_head(prev, cuts, rt, s_first, i, &t)) mine++;
    }
    mine = wave_scan_u32(mine);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t tot = 0; for (int q = 0; q < kParBlock / 64; q++) tot += wsum[q]; tile_cnt[blockIdx.x] = tot; }
}

// One workgroup of 1024 lanes: tile_cnt[0 .. n_tiles) -> its exclusive prefix, in place; n_tiles <= 2^24 / kRankTile = 8192.
__global__ __launch_bounds__(1024) void k_par_rank_scan(uint32_t* __restrict__ tile_cnt, uint32_t n_tiles, uint32_t expect_total, uint32_t* __restrict__ bad) {
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += 1024u) {
        const uint32_t j = base + (uint32_t)tid;
        const uint32_t v = j < n_tiles ? tile_cnt[j] : 0u;
        const uint32_t incl = wave_scan_u32(v);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) { const uint32_t x = wsum[q]; total += x; woff += q < wv ? x : 0u; }
        if (j < n_tiles) tile_cnt[j] = carry + woff + incl - v;
        carry += total;
        __syncthreads();
    }
    if (tid == 0 && carry != expect_total) atomicExch(bad, 2u);
}

__global__ __launch_bounds__(kParBlock) void k_par_rank_write(const int32_t* __restrict__ prev, const uint32_t* __restrict__ cuts, uint32_t t_lo, uint32_t t_hi,
                                                              uint32_t i_lo, uint32_t i_hi, uint32_t max_entries, const uint32_t* __restrict__ tile_off,
                                                              uint32_t* __restrict__ pos, uint32_t* __restrict__ bad) {
    constexpr int kWaves = kParBlock / 64;
    __shared__ uint32_t sh[2], wcnt[kRankPer][kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t base = i_lo + blockIdx.x * kRankTile;
    const RankTile rt = rank_tile_epochs(cuts, t_lo, t_hi, base, i_hi, sh);
    const uint32_t s_first = cuts[rt.t_first];
    unsigned long long m[kRankPer];
    uint32_t ep[kRankPer];
#pragma unroll
    for (int k = 0; k < kRankPer; k++) {
        const uint32_t i = base + (uint32_t)k * kParBlock + (uint32_t)tid;
        uint32_t t = 0;
        const bool head = i < i_hi && rank_is_head(prev, cuts, rt, s_first, i, &t);
        ep[k] = t;
        m[k] = __ballot(head);
        if (lane == 0) wcnt[k][wv] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    uint32_t running = t_lo * max_entries + tile_off[blockIdx.x];          // heads before this tile, counted from the first middle epoch
#pragma unroll
    for (int k = 0; k < kRankPer; k++) {
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int q = 0; q < kWaves; q++) { const uint32_t x = wcnt[k][q]; total += x; woff += q < wv ? x : 0u; }
        const uint32_t i = base + (uint32_t)k * kParBlock + (uint32_t)tid;
        if (i < i_hi) {
            const bool head = (m[k] >> lane) & 1ull;
            const uint32_t p = running + woff + (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull));
            pos[i] = head ? p : kParNone;
            // the epoch's first record: a head whose position is the epoch's first — or an earlier epoch does not hold max_entries
            if (i == (rt.t_last != rt.t_first ? cuts[ep[k]] : s_first) && (!head || p != ep[k] * max_entries)) atomicExch(bad, 1u);
        }
        running += total;
    }
}

// ---- the fold of a segment --------------------------------------------------------------------------------------------------------
// model.AccumulateBase(p, other) (pkg/model/flow_content.go:28-61) on whole records, p = the stored first record (account.go:95).
NF_DEV void accumulate_base(Rec& p, const Rec& o) {
    const uint64_t ps = p.start(), os = o.start();
    if (ps == 0 || (ps > os && os != 0)) { p.d[10] = o.d[10]; p.d[11] = o.d[11]; }                  // :36-38
    const uint64_t pe = p.end(), oe = o.end();
    if (pe == 0 || pe < oe) { p.d[12] = o.d[12]; p.d[13] = o.d[13]; }                               // :39-41
    const uint64_t by = p.bytes() + o.bytes();                                                      // :42
    p.d[14] = (uint32_t)by; p.d[15] = (uint32_t)(by >> 32);
    p.d[16] += o.d[16];                                                                             // :43
    p.d[17] |= o.d[17] & 0xffff0000u;                                                               // :44 flags
    if (o.eth()) p.d[17] = (p.d[17] & 0xffff0000u) | o.eth();                                       // :45-47
    if (p.smac() == 0) { p.d[18] = o.d[18]; p.d[19] = (p.d[19] & 0xffff0000u) | (o.d[19] & 0x0000ffffu); }   // :48-50
    if (p.dmac() == 0) { p.d[19] = (p.d[19] & 0x0000ffffu) | (o.d[19] & 0xffff0000u); p.d[20] = o.d[20]; }   // :51-53
    if (o.dscp()) p.d[24] = (p.d[24] & 0xff00ffffu) | (o.d[24] & 0x00ff0000u);                      // :54-56
    if (o.sampling()) p.d[23] = o.d[23];                                                            // :57-59
}

// bytes 0..111 of a record: key and every field AccumulateBase reads (the rest of `r` stays as it is)
NF_DEV void load_record_head(const void* base, uint64_t i, Rec& r) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + i * kRecordBytes);
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const uint4 v = p[k];
        r.d[4 * k] = v.x; r.d[4 * k + 1] = v.y; r.d[4 * k + 2] = v.z; r.d[4 * k + 3] = v.w;
    }
}
NF_DEV void store_record(void* base, uint64_t i, const Rec& r) {
    uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + i * kRecordBytes);
#pragma unroll
    for (int k = 0; k < 9; k++) p[k] = make_uint4(r.d[4 * k], r.d[4 * k + 1], r.d[4 * k + 2], r.d[4 * k + 3]);
}

// the end of the segment that begins at sorted position p: the first position in (p, n] whose sort key reaches `limit` = (hash
// bits, end of the epoch) — a binary search, the array is sorted
NF_DEV uint64_t seg_end(const uint64_t* __restrict__ ks, uint64_t n, uint64_t p, uint64_t limit) {
    uint64_t lo = p + 1, hi = n;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (ks[mid] < limit) lo = mid + 1; else hi = mid; }
    return lo;
}

// One lane per sorted position p whose record i starts a flow in a middle epoch (pos[i] != kParNone). Its segment: the positions
// from p up to (not including) the first whose sort key reaches (hash bits, first record of the next epoch) — the array is sorted
// by (hash bits, index), so that is where the flow's records of this epoch end — less the records of other flows with the same
// hash bits (full keys compared). Up to kSegShort positions: folded here, in arrival order, exactly as account.go:82-95 does;
// longer ones are listed for k_par_segfold_long. A launch folds the epochs whose records are [i_lo, i_hi) (cut to cut): the ones
// k_par_rank has just given positions.
// pos[] counts from the first middle epoch, whose cut is cuts[0].
template <bool SKETCH>
__global__ __launch_bounds__(kParBlock) void k_par_segfold(const void* __restrict__ recs, const uint64_t* __restrict__ ks, uint64_t n,
                                                           const uint32_t* __restrict__ pos, const uint32_t* __restrict__ cuts,
                                                           uint32_t max_entries, SketchView sk, void* __restrict__ out,
                                                           uint32_t* __restrict__ long_list, uint32_t* __restrict__ n_long, uint32_t long_cap,
                                                           uint32_t* __restrict__ huge_list, uint32_t* __restrict__ n_huge,
                                                           uint32_t i_lo, uint32_t i_hi) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t key = ks[p];
    const uint32_t i = key_index(key);
    if (i < i_lo || i >= i_hi) return;                                // not a record of the epochs this launch folds (pos[] holds nothing for it, or an earlier launch's)
    const uint32_t ps = pos[i];
    if (ps == kParNone) return;
    const uint64_t limit = (key & kHashMask) | (uint64_t)cuts[ps / max_entries + 1];     // (hash bits, end of the epoch)
    if (p + kSegShort < n && ks[p + kSegShort] < limit) {             // more than kSegShort positions: a wave takes it
        // The list is bounded. Segments of flows that do not share their 40 hash bits are disjoint runs of more than kSegShort
        // positions: at most n / (kSegShort + 1) of them. Flows that DO share them (crafted: key_hash has a fixed seed) interleave
        // in one run, and every head with kSegShort positions of the run behind it lands here — G such flows list ~G segments
        // from G records. Beyond the list's room this lane folds its segment itself, however long the run is: slower, never out
        // of bounds (tests/test_account_par_gpu.py::test_many_runs_of_flows_that_share_their_key_hash).
        // ... and a segment of more than kSegHuge positions goes in chunks of kHugeChunk, a wave per chunk, the chunks' partials
        // combined afterwards (k_par_huge_chunks / k_par_huge_combine): the hot flows of a LONG epoch (CACHE_MAX_FLOWS = 100 000:
        // ~600 k records per epoch, 42 k of them the hottest flow's — 656 dependent gathers for one wave, 1.3 ms on the critical
        // path of every launch; a workgroup of 1024 lanes per segment still took 0.73 ms: profiles/r06_account_regimes.txt).
        // This lane finds the segment's end (a binary search: these segments are few) and lists the chunks.
        if (p + kSegHuge < n && ks[p + kSegHuge] < limit) {
            const uint64_t end = seg_end(ks, n, p, limit);
            const uint32_t chunks = (uint32_t)((end - p + kHugeChunk - 1) / kHugeChunk);
            const uint32_t ah = atomicAdd(&n_huge[0], 1u);
            if (ah < kHugeCap) {
                const uint32_t c0 = atomicAdd(&n_huge[1], chunks);
                if (c0 + chunks <= kHugeChunkCap) {
                    huge_list[3 * ah] = (uint32_t)p; huge_list[3 * ah + 1] = c0; huge_list[3 * ah + 2] = chunks;
                    uint32_t* cl = huge_list + 3 * kHugeCap;
                    for (uint32_t c = 0; c < chunks; c++) cl[c0 + c] = ah;
                    return;
                }
                huge_list[3 * ah] = (uint32_t)p; huge_list[3 * ah + 1] = 0; huge_list[3 * ah + 2] = 0;    // (no room for its chunks — cannot happen within the
                                                                                                         // caps above — : listed empty, folded as a long segment)
            }
        }
        const uint32_t at = atomicAdd(n_long, 1u);
        if (at < long_cap) { long_list[at] = (uint32_t)p; return; }
    }
    Rec acc;
    load_record(recs, i, acc);
    acc.canonicalize();
    uint64_t w[5];
    acc.key_words(w);
    for (uint64_t q = p + 1; q < n; q++) {
        const uint64_t k2 = ks[q];
        if (k2 >= limit) break;
        Rec r;
        load_record_head(recs, key_index(k2), r);
        r.d[9] &= 0x00ffffffu;
        uint64_t w2[5];
        r.key_words(w2);
        if (!par_same_key(w, w2)) continue;                           // another flow with these hash bits
        accumulate_base(acc, r);
    }
    if (SKETCH) sketch_add(sk, w, acc.bytes());                       // Count-Min is linear, HyperLogLog idempotent: once per segment
    store_record(out, ps, acc);
}

// What a run of records of one flow contributes, in a form that combines in any order: the index of the record decides what
// "first" and "last" mean (the tagged words of the table, nfagg_device.h, in registers).
struct SegAcc {
    uint64_t bytes, end, start_inv, eth_tag, dscp_tag, samp_tag, smac, dmac;
    uint32_t packets, flags, smac_at, dmac_at;
    NF_DEV void clear() {
        bytes = end = start_inv = eth_tag = dscp_tag = samp_tag = smac = dmac = 0;
        packets = flags = 0; smac_at = dmac_at = kParNone;
    }
    NF_DEV void add(const Rec& r, uint32_t idx) {
        bytes += r.bytes(); packets += r.packets(); flags |= r.flags();
        if (r.end() > end) end = r.end();
        const uint64_t si = r.start() ? ~r.start() : 0ull;
        if (si > start_inv) start_inv = si;
        const uint64_t s1 = (uint64_t)idx + 1;
        if (r.eth()) { const uint64_t v = (s1 << 16) | r.eth(); if (v > eth_tag) eth_tag = v; }
        if (r.dscp()) { const uint64_t v = (s1 << 8) | r.dscp(); if (v > dscp_tag) dscp_tag = v; }
        if (r.sampling()) { const uint64_t v = (s1 << 32) | r.sampling(); if (v > samp_tag) samp_tag = v; }
        if (r.smac() && idx < smac_at) { smac_at = idx; smac = r.smac(); }
        if (r.dmac() && idx < dmac_at) { dmac_at = idx; dmac = r.dmac(); }
    }
    NF_DEV void combine(const SegAcc& o) {
        bytes += o.bytes; packets += o.packets; flags |= o.flags;
        if (o.end > end) end = o.end;
        if (o.start_inv > start_inv) start_inv = o.start_inv;
        if (o.eth_tag > eth_tag) eth_tag = o.eth_tag;
        if (o.dscp_tag > dscp_tag) dscp_tag = o.dscp_tag;
        if (o.samp_tag > samp_tag) samp_tag = o.samp_tag;
        if (o.smac_at < smac_at) { smac_at = o.smac_at; smac = o.smac; }
        if (o.dmac_at < dmac_at) { dmac_at = o.dmac_at; dmac = o.dmac; }
    }
    // the flow's first record (it is part of the run) becomes the eviction: every field AccumulateBase touches from the run
    NF_DEV void apply(Rec& head) const {
        const uint64_t st = start_inv ? ~start_inv : 0ull;
        head.d[10] = (uint32_t)st; head.d[11] = (uint32_t)(st >> 32);
        head.d[12] = (uint32_t)end; head.d[13] = (uint32_t)(end >> 32);
        head.d[14] = (uint32_t)bytes; head.d[15] = (uint32_t)(bytes >> 32);
        head.d[16] = packets;
        head.d[17] = (flags << 16) | (uint32_t)(eth_tag & 0xffffu);
        head.d[18] = (uint32_t)smac;
        head.d[19] = (uint32_t)((smac >> 32) & 0xffffu) | ((uint32_t)(dmac & 0xffffu) << 16);
        head.d[20] = (uint32_t)(dmac >> 16);
        head.d[23] = (uint32_t)samp_tag;
        head.d[24] = (head.d[24] & 0xff00ffffu) | ((uint32_t)(dscp_tag & 0xffu) << 16);
    }
};

NF_DEV uint64_t shfl_xor_u64(uint64_t v, int m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// What the lanes of one wave hold after they have taken the positions [first + lane, end) `step` apart, combined over the wave
// (xor-shuffles: every lane ends up with the wave's partial).
NF_DEV void seg_gather(const void* __restrict__ recs, const uint64_t* __restrict__ ks, uint64_t first, uint64_t end, uint64_t step,
                       const uint64_t w[5], SegAcc& a) {
    a.clear();
    // kSegUnroll positions per lane and round, their loads issued together: with one gather in flight per lane a segment of the
    // hottest flows was a chain of memory round trips (k_par_segfold_long 0.35 ms per part at 100 000 entries:
    // profiles/r06x_walk_ungated.txt)
    for (uint64_t q0 = first; q0 < end; q0 += kSegUnroll * step) {
        uint32_t i2[kSegUnroll];
        Rec r[kSegUnroll];
#pragma unroll
        for (int u = 0; u < kSegUnroll; u++) {
            const uint64_t q = q0 + (uint64_t)u * step;
            i2[u] = q < end ? key_index(ks[q]) : kParNone;
        }
#pragma unroll
        for (int u = 0; u < kSegUnroll; u++) if (i2[u] != kParNone) load_record_head(recs, i2[u], r[u]);
#pragma unroll
        for (int u = 0; u < kSegUnroll; u++) {
            if (i2[u] == kParNone) continue;
            r[u].d[9] &= 0x00ffffffu;
            uint64_t w2[5];
            r[u].key_words(w2);
            if (par_same_key(w, w2)) a.add(r[u], i2[u]);
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        SegAcc o;
        o.bytes = shfl_xor_u64(a.bytes, m); o.end = shfl_xor_u64(a.end, m); o.start_inv = shfl_xor_u64(a.start_inv, m);
        o.eth_tag = shfl_xor_u64(a.eth_tag, m); o.dscp_tag = shfl_xor_u64(a.dscp_tag, m); o.samp_tag = shfl_xor_u64(a.samp_tag, m);
        o.smac = shfl_xor_u64(a.smac, m); o.dmac = shfl_xor_u64(a.dmac, m);
        o.packets = (uint32_t)__shfl_xor((int)a.packets, m); o.flags = (uint32_t)__shfl_xor((int)a.flags, m);
        o.smac_at = (uint32_t)__shfl_xor((int)a.smac_at, m); o.dmac_at = (uint32_t)__shfl_xor((int)a.dmac_at, m);
        a.combine(o);
    }
}

// One wave per listed segment. The lanes take the positions 64 at a time; what they hold is combined with xor-shuffles; lane 0
// applies it to the first record and stores.
template <bool SKETCH>
__global__ __launch_bounds__(kParBlock) void k_par_segfold_long(const void* __restrict__ recs, const uint64_t* __restrict__ ks, uint64_t n,
                                                                const uint32_t* __restrict__ pos, const uint32_t* __restrict__ cuts,
                                                                uint32_t max_entries, SketchView sk, void* __restrict__ out,
                                                                const uint32_t* __restrict__ long_list, const uint32_t* __restrict__ n_long, uint32_t long_cap) {
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * (kParBlock / 64);
    const uint32_t count = *n_long < long_cap ? *n_long : long_cap;  // (what did not fit was folded by k_par_segfold's own lanes)
    for (uint32_t e = blockIdx.x * (kParBlock / 64) + (threadIdx.x >> 6); e < count; e += waves) {
        const uint64_t p = long_list[e];
        const uint64_t key = ks[p];
        const uint32_t i = key_index(key);
        const uint32_t ps = pos[i];
        const uint64_t limit = (key & kHashMask) | (uint64_t)cuts[ps / max_entries + 1];
        const uint64_t end = seg_end(ks, n, p, limit);
        Rec head;
        load_record(recs, i, head);
        head.canonicalize();
        uint64_t w[5];
        head.key_words(w);
        SegAcc a;
        seg_gather(recs, ks, p + lane, end, 64, w, a);
        if (lane == 0) {
            a.apply(head);
            if (SKETCH) sketch_add(sk, w, a.bytes);
            store_record(out, ps, head);
        }
    }
}

// The segments of more than kSegHuge positions, in two steps. huge_list: {p, first chunk, chunks} per segment, then (from
// 3 * kHugeCap on) the segment of every chunk; n_huge[0] segments, n_huge[1] chunks.
//   k_par_huge_chunks    one wave per chunk of kHugeChunk positions: its order-free partial (SegAcc) to `partials`
//   k_par_huge_combine   one wave per segment: the chunks' partials combined (order-free), applied to the first record, stored
struct alignas(16) SegPartial { uint64_t w[12]; };                    // a SegAcc, 96 bytes
NF_DEV void seg_pack(const SegAcc& a, SegPartial& o) {
    o.w[0] = a.bytes; o.w[1] = a.end; o.w[2] = a.start_inv; o.w[3] = a.eth_tag; o.w[4] = a.dscp_tag; o.w[5] = a.samp_tag; o.w[6] = a.smac; o.w[7] = a.dmac;
    o.w[8] = (uint64_t)a.packets | ((uint64_t)a.flags << 32); o.w[9] = (uint64_t)a.smac_at | ((uint64_t)a.dmac_at << 32); o.w[10] = 0; o.w[11] = 0;
}
NF_DEV void seg_unpack(const SegPartial& o, SegAcc& a) {
    a.bytes = o.w[0]; a.end = o.w[1]; a.start_inv = o.w[2]; a.eth_tag = o.w[3]; a.dscp_tag = o.w[4]; a.samp_tag = o.w[5]; a.smac = o.w[6]; a.dmac = o.w[7];
    a.packets = (uint32_t)o.w[8]; a.flags = (uint32_t)(o.w[8] >> 32); a.smac_at = (uint32_t)o.w[9]; a.dmac_at = (uint32_t)(o.w[9] >> 32);
}
NF_DEV void seg_wave_combine(SegAcc& a) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        SegAcc o;
        o.bytes = shfl_xor_u64(a.bytes, m); o.end = shfl_xor_u64(a.end, m); o.start_inv = shfl_xor_u64(a.start_inv, m);
        o.eth_tag = shfl_xor_u64(a.eth_tag, m); o.dscp_tag = shfl_xor_u64(a.dscp_tag, m); o.samp_tag = shfl_xor_u64(a.samp_tag, m);
        o.smac = shfl_xor_u64(a.smac, m); o.dmac = shfl_xor_u64(a.dmac, m);
        o.packets = (uint32_t)__shfl_xor((int)a.packets, m); o.flags = (uint32_t)__shfl_xor((int)a.flags, m);
        o.smac_at = (uint32_t)__shfl_xor((int)a.smac_at, m); o.dmac_at = (uint32_t)__shfl_xor((int)a.dmac_at, m);
        a.combine(o);
    }
}

__global__ __launch_bounds__(kParBlock) void k_par_huge_chunks(const void* __restrict__ recs, const uint64_t* __restrict__ ks, uint64_t n,
                                                               const uint32_t* __restrict__ pos, const uint32_t* __restrict__ cuts, uint32_t max_entries,
                                                               const uint32_t* __restrict__ huge_list, const uint32_t* __restrict__ n_huge,
                                                               SegPartial* __restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * (kParBlock / 64);
    const uint32_t n_seg = n_huge[0] < kHugeCap ? n_huge[0] : kHugeCap;
    const uint32_t count = n_huge[1] < kHugeChunkCap ? n_huge[1] : kHugeChunkCap;
    const uint32_t* chunk_seg = huge_list + 3 * kHugeCap;
    for (uint32_t c = blockIdx.x * (kParBlock / 64) + (threadIdx.x >> 6); c < count; c += waves) {
        const uint32_t sg = chunk_seg[c];
        if (sg >= n_seg) continue;                                    // (wave-uniform)
        const uint64_t p = huge_list[3 * sg];
        const uint32_t c0 = huge_list[3 * sg + 1], chunks = huge_list[3 * sg + 2];
        if (c - c0 >= chunks) continue;                               // a chunk entry of a segment that was listed empty
        const uint64_t key = ks[p];
        const uint32_t i = key_index(key);
        const uint64_t limit = (key & kHashMask) | (uint64_t)cuts[pos[i] / max_entries + 1];
        uint64_t w[5];
        par_key(recs, i, w);
        const uint64_t lo = p + (uint64_t)(c - c0) * kHugeChunk;
        uint64_t hi = lo + kHugeChunk;
        if (hi > n) hi = n;
        // the chunk ends where the segment does: the last chunk of a segment stops at the first key that reaches `limit`
        // (positions are sorted: once a key reaches it, every later one does)
        SegAcc a;
        a.clear();
        for (uint64_t q0 = lo + lane; q0 < hi; q0 += 64u * kSegUnroll) {      // (kSegUnroll gathers in flight per lane: seg_gather)
            uint32_t i2[kSegUnroll];
            Rec r[kSegUnroll];
#pragma unroll
            for (int u = 0; u < kSegUnroll; u++) {
                const uint64_t q = q0 + 64u * (uint64_t)u;
                const uint64_t k2 = q < hi ? ks[q] : ~0ull;
                i2[u] = k2 < limit ? key_index(k2) : kParNone;               // (sorted: once a key reaches the limit, every later one does)
            }
#pragma unroll
            for (int u = 0; u < kSegUnroll; u++) if (i2[u] != kParNone) load_record_head(recs, i2[u], r[u]);
#pragma unroll
            for (int u = 0; u < kSegUnroll; u++) {
                if (i2[u] == kParNone) continue;
                r[u].d[9] &= 0x00ffffffu;
                uint64_t w2[5];
                r[u].key_words(w2);
                if (par_same_key(w, w2)) a.add(r[u], i2[u]);
            }
            if (i2[kSegUnroll - 1] == kParNone) break;
        }
        seg_wave_combine(a);
        if (lane == 0) seg_pack(a, partials[c]);
    }
}

template <bool SKETCH>
__global__ __launch_bounds__(kParBlock) void k_par_huge_combine(const void* __restrict__ recs, const uint64_t* __restrict__ ks,
                                                                const uint32_t* __restrict__ pos, SketchView sk, void* __restrict__ out,
                                                                const uint32_t* __restrict__ huge_list, const uint32_t* __restrict__ n_huge,
                                                                const SegPartial* __restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * (kParBlock / 64);
    const uint32_t n_seg = n_huge[0] < kHugeCap ? n_huge[0] : kHugeCap;
    for (uint32_t sg = blockIdx.x * (kParBlock / 64) + (threadIdx.x >> 6); sg < n_seg; sg += waves) {
        const uint64_t p = huge_list[3 * sg];
        const uint32_t c0 = huge_list[3 * sg + 1], chunks = huge_list[3 * sg + 2];
        if (chunks == 0) continue;
        const uint32_t i = key_index(ks[p]);
        SegAcc a;
        a.clear();
        for (uint32_t c = lane; c < chunks; c += 64) { SegAcc o; seg_unpack(partials[c0 + c], o); a.combine(o); }
        seg_wave_combine(a);
        if (lane == 0) {
            Rec head;
            load_record(recs, i, head);
            head.canonicalize();
            uint64_t w[5];
            head.key_words(w);
            a.apply(head);
            if (SKETCH) sketch_add(sk, w, a.bytes);
            store_record(out, pos[i], head);
        }
    }
}

// ---- launch wrappers ----------------------------------------------------------------------------------------------------------
hipError_t launch_par_hash(const void* d_records, uint64_t n, uint64_t* d_keys, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_hash, dim3(par_grid(n, kParBlock, 8192)), dim3(kParBlock), 0, s, d_records, n, d_keys);
    return hipGetLastError();
}

// temp == nullptr: only *temp_bytes is written. Sorted on the hash bits only: the indices below them are in order already, and the
// radix passes are stable. ALWAYS the one-sweep radix passes (MergeSortLimit 0): up to 2^20 keys rocPRIM would take its block-sort +
// merge path, and with a partial bit range that path does not deliver the stable sort on those bits (tools/gpu/sort_probe.hip: at
// 600 000 keys, bits [32, 64) or [24, 64), every position differs from std::stable_sort; the one-sweep passes are right for every
// range and size tried: profiles/r05_sort_merge_path.txt).
using ParSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
hipError_t launch_par_sort(void* temp, size_t* temp_bytes, const uint64_t* k_in, uint64_t* k_out, uint64_t n, hipStream_t s) {
    return rocprim::radix_sort_keys<ParSortConfig>(temp, *temp_bytes, k_in, k_out, (size_t)n, kIdxBits, 64, s);
}

hipError_t launch_par_links(const void* d_records, const uint64_t* d_keys_sorted, uint64_t n, int32_t* d_prev, uint32_t* d_overflow, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_links, dim3(par_grid(n, kParBlock, 8192)), dim3(kParBlock), 0, s, d_records, d_keys_sorted, n, d_prev, d_overflow);
    return hipGetLastError();
}

hipError_t launch_par_live(const TableView& t, const void* d_records, int32_t* d_prev, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_live, dim3(par_grid(n, kParBlock, 8192)), dim3(kParBlock), 0, s, t, d_records, d_prev, n);
    return hipGetLastError();
}

// The walk over the blocks [b_begin, b_end) of par_cut_span() records each (k_par_cuts: b_begin > 0 resumes from the state in d_ctl);
// b_end - b_begin: a multiple of four, b_end <= par_walk_blocks(n).
// host_cuts / host_state: the host's pinned cut list and this part's four state words there ([0] cuts so far [1] new flows of the
// epoch in progress [2] the links' overflow flag [3] cuts before this part), or null.
hipError_t launch_par_cuts(const int32_t* d_prev, uint64_t n, uint32_t max_entries, uint32_t live0, uint32_t* d_cuts, uint32_t max_cuts,
                           uint32_t* d_ctl, uint64_t b_begin, uint64_t b_end, uint32_t* host_cuts, uint32_t* host_state, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_par_cuts, dim3(1), dim3(kCutBlock), 0, s, d_prev, n, max_entries, live0, d_cuts, max_cuts, d_ctl, b_begin, b_end, host_cuts, host_state);
    return hipGetLastError();
}
uint64_t par_cut_span() { return kCutSpan; }
// entries of prev[] for a launch of n records: its blocks (a multiple of four) and the four the walk requests ahead (cut_load);
// [n, this) = kCutNever
uint64_t par_walk_blocks(uint64_t n) { return (((n + kCutSpan - 1) / kCutSpan + 3) / 4) * 4; }     // the walk takes its blocks four at a time
uint64_t par_prev_entries(uint64_t n) { return (par_walk_blocks(n) + 4) * kCutSpan; }
int32_t par_prev_pad_value() { return kCutNever; }

uint64_t par_rank_tiles(uint64_t records) { return (records + kRankTile - 1) / kRankTile; }

// The complete epochs [t_lo, t_hi) of the middle (epoch t = records [cuts[t], cuts[t + 1]); i_lo = cuts[t_lo], i_hi = cuts[t_hi]):
// positions, then both folds. d_out: where the FIRST middle epoch's eviction begins (epoch t goes to d_out + t * max_entries records).
// d_long: room for long_cap positions, d_huge: par_huge_cap() words; d_n_long: three counters; d_tiles: par_rank_tiles(i_hi - i_lo) counters; *d_n_long is zeroed here (in stream order);
// *d_bad accumulates.
hipError_t launch_par_middle(const void* d_records, const uint64_t* d_keys_sorted, uint64_t n, const int32_t* d_prev, const uint32_t* d_cuts,
                             uint32_t t_lo, uint32_t t_hi, uint32_t i_lo, uint32_t i_hi, uint32_t max_entries, const SketchView& sk, uint32_t* d_pos,
                             void* d_out, uint32_t* d_long, uint32_t long_cap, uint32_t* d_huge, uint32_t* d_tiles, uint32_t* d_n_long, uint32_t* d_bad, hipStream_t s) {
    if (t_hi <= t_lo || i_hi <= i_lo) return hipSuccess;
    const uint32_t n_mid = t_hi - t_lo;
    hipError_t e = hipMemsetAsync(d_n_long, 0, 3 * sizeof(uint32_t), s);    // [0] long segments listed, [1] huge ones, [2] their chunks
    uint32_t* d_n_huge = d_n_long + 1;
    SegPartial* d_partials = reinterpret_cast<SegPartial*>(((uintptr_t)(d_huge + 3 * kHugeCap + kHugeChunkCap) + 15u) & ~(uintptr_t)15u);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    const uint32_t n_tiles = (uint32_t)par_rank_tiles(i_hi - i_lo);
    hipLaunchKernelGGL(k_par_rank_count, dim3(n_tiles), dim3(kParBlock), 0, s, d_prev, d_cuts, t_lo, t_hi, i_lo, i_hi, d_tiles);
    hipLaunchKernelGGL(k_par_rank_scan, dim3(1), dim3(1024), 0, s, d_tiles, n_tiles, n_mid * max_entries, d_bad);
    hipLaunchKernelGGL(k_par_rank_write, dim3(n_tiles), dim3(kParBlock), 0, s, d_prev, d_cuts, t_lo, t_hi, i_lo, i_hi, max_entries, (const uint32_t*)d_tiles, d_pos, d_bad);
    if (sk.flags) {
        hipLaunchKernelGGL(k_par_segfold<true>, dim3(par_grid(n)), dim3(kParBlock), 0, s, d_records, d_keys_sorted, n, (const uint32_t*)d_pos, d_cuts, max_entries, sk, d_out, d_long, d_n_long, long_cap, d_huge, d_n_huge, i_lo, i_hi);
        hipLaunchKernelGGL(k_par_huge_chunks, dim3(1024), dim3(kParBlock), 0, s, d_records, d_keys_sorted, n, (const uint32_t*)d_pos, d_cuts, max_entries, (const uint32_t*)d_huge, (const uint32_t*)d_n_huge, d_partials);
        hipLaunchKernelGGL(k_par_huge_combine<true>, dim3(64), dim3(kParBlock), 0, s, d_records, d_keys_sorted, (const uint32_t*)d_pos, sk, d_out, (const uint32_t*)d_huge, (const uint32_t*)d_n_huge, (const SegPartial*)d_partials);
        hipLaunchKernelGGL(k_par_segfold_long<true>, dim3(2048), dim3(kParBlock), 0, s, d_records, d_keys_sorted, n, (const uint32_t*)d_pos, d_cuts, max_entries, sk, d_out, (const uint32_t*)d_long, (const uint32_t*)d_n_long, long_cap);
    } else {
        hipLaunchKernelGGL(k_par_segfold<false>, dim3(par_grid(n)), dim3(kParBlock), 0, s, d_records, d_keys_sorted, n, (const uint32_t*)d_pos, d_cuts, max_entries, sk, d_out, d_long, d_n_long, long_cap, d_huge, d_n_huge, i_lo, i_hi);
        hipLaunchKernelGGL(k_par_huge_chunks, dim3(1024), dim3(kParBlock), 0, s, d_records, d_keys_sorted, n, (const uint32_t*)d_pos, d_cuts, max_entries, (const uint32_t*)d_huge, (const uint32_t*)d_n_huge, d_partials);
        hipLaunchKernelGGL(k_par_huge_combine<false>, dim3(64), dim3(kParBlock), 0, s, d_records, d_keys_sorted, (const uint32_t*)d_pos, sk, d_out, (const uint32_t*)d_huge, (const uint32_t*)d_n_huge, (const SegPartial*)d_partials);
        hipLaunchKernelGGL(k_par_segfold_long<false>, dim3(2048), dim3(kParBlock), 0, s, d_records, d_keys_sorted, n, (const uint32_t*)d_pos, d_cuts, max_entries, sk, d_out, (const uint32_t*)d_long, (const uint32_t*)d_n_long, long_cap);
    }
    return hipGetLastError();
}
uint32_t par_seg_short() { return kSegShort; }
uint32_t par_huge_cap() { return 3 * kHugeCap + kHugeChunkCap + (uint32_t)(kHugeChunkCap * sizeof(SegPartial) / sizeof(uint32_t)) + 8; }   // words: segment list, chunk list, chunk partials

}  // namespace nfagg
