// nfagg_epochs.hip — Accounter.Account's record arm INCLUDING its eviction on "full" (pkg/flow/account.go:81-96) as ONE
// persistent kernel, for small CACHE_MAX_FLOWS (the reference ships 5000, pkg/config/config.go:146).
//
// With a small map the stream stops on "full" every few thousand records: an epoch is ~14 k records of the configs[1] stream
// at 5000 entries. Driven from the host that is six small launches and two round trips per epoch (130 us: 0.11 G records/s
// device-resident, 36 M/s through nfagg_ingest — round 2). Here the whole loop runs on the device: a cooperative grid of
// 64 x 256 lanes takes the batch window by window (one record per lane), and per window
//   P1  claims: find-or-claim the record's slot, plant its sequence number as a candidate "first record" (tag of id0)
//   P2  flags the records that are the first of a key new to the map (their tag won) and counts them per block
//   P3  every block prefixes the 64 block counts: the map is full at the (room+1)-th such record — the block that holds it
//       finds its position (ballot ranks): the SPLIT, exactly where account.go:85 finds len(entries) >= maxEntries
//   P4  folds the records before the split (model.AccumulateBase, flow_content.go:28-61: the operators of nfagg_device.h)
//   P5  on full: every live flow whose first record precedes the split is written to the output (Accounter.evict,
//       account.go:102-124; identity dwords straight from the first record in the batch), the epoch tag is bumped — nothing
//       is cleared — and the next window starts AT the split: that record is inserted into the empty map, as :95 does.
// Grid-wide synchronisation (cooperative groups) separates the phases: five per evicted epoch, two per window that does not
// fill the map. Control state (position, sequence, len(entries), output position, epoch tag) is REPLICATED in every lane's
// registers — all lanes see the same block counts and the same split — so no phase waits for a "manager" lane. The host
// reads the control block once per call.
//
// Coherence inside ONE kernel across the 8 XCDs (no kernel boundary between an epoch's fold and its eviction, or between an
// eviction and the re-use of its slots): every word of a slot, the live-list ring and the control block are accessed with
// agent-scope atomics / atomic loads and stores only. Nothing here takes the plain-load shortcuts of the fold kernels
// (probe_home, hints): a line cached during an earlier epoch of this very launch would be stale.
#include <hip/hip_cooperative_groups.h>
#include <string.h>
#include "nfagg_device.h"

namespace cg = cooperative_groups;

namespace nfagg {

constexpr int kEpBlock = 256, kEpGrid = 64;
constexpr uint32_t kEpWindow = kEpBlock * kEpGrid;      // records examined per window: one per lane

struct EpochCtl {
    // in/out (host writes before the launch, lane 0 writes back at the end)
    unsigned long long pos;         // records of the batch consumed
    unsigned long long seq;         // epoch-relative sequence number of record `pos`
    unsigned long long live;        // len(entries)
    unsigned long long list_base;   // live-list ring position of the epoch's first claimed slot
    unsigned long long list_fin;    // ring positions below it were claimed before this launch: identity dwords in the cold line
    unsigned long long out_pos;     // records written to the output
    unsigned long long epoch_bits;  // current epoch tag << 48
    uint32_t n_epochs;              // evictions performed by this launch
    uint32_t stop;                  // 1 batch consumed, 2 no room for another eviction in the output / epoch list, 3 epoch tags wrap next
    // scratch of the running kernel
    uint32_t split, pad0;
    unsigned long long out_cursor;  // absolute write position of the evict phase
    unsigned long long bar;         // grid barrier: arrivals, never reset (barrier k is passed at k x blocks)
    uint32_t abort, pad1;           // a block gave up waiting at a barrier: everybody leaves, the API reports NFAGG_EDEVICE
    uint32_t block_count[kEpGrid];
    unsigned long long phase[8];    // diagnostics (lane 0): 100 MHz ticks in P1 claim, sync, P2, sync, P3 (+ sync), P4, sync + P5, sync
};

struct EpochArgs {
    const void* recs; uint64_t n;
    void* out; uint64_t out_cap;          // evicted records, appended; capacity in records
    uint64_t* epoch_end; uint32_t max_epochs;
    uint64_t max_entries;
    EpochCtl* ctl;
};

// c.entries[id] lookup-or-insert with coherent accesses only; the live list is a ring (positions never restart inside a launch).
// The lanes of a wave that win a slot in the same trip reserve their live-list positions with ONE atomic (ballot + rank): a
// returning atomic on one address retires every ~14 ns however many lanes issue it — one per claim was 70-260 us per epoch.
NF_DEV uint32_t ep_find_or_claim(const TableView& t, const uint64_t w[5], uint64_t h, bool active) {
    const uint64_t ready = tag_ready(t, h), locked = tag_locked(t, h);
    const int lane = threadIdx.x & 63;
    uint64_t idx = h & t.mask, probes = 0;
    uint32_t result = kNoSlot, trips = 0;
    bool done = !active;
    while (__ballot(!done)) {                                    // wave-uniform loop: the ballots below see every lane
        if (++trips > kSpinLimit) { if (!done) atomicExch(&t.ctr->error, 2u); break; }
        SlotHot* s = &t.hot[idx];
        bool won = false;
        uint64_t tag = 0, kw[5] = {0, 0, 0, 0, 0};
        if (!done) {
            // tag and key in ONE round trip. The key words may be older than the tag (they are separate loads): a MATCH under a
            // ready tag of this epoch and fingerprint is the flow (its claimer wrote exactly these words before publishing); a
            // mismatch is looked at again, key after tag, before the probe moves on.
            tag = ald(&s->tag);
#pragma unroll
            for (int k = 0; k < 5; k++) kw[k] = ald(&s->key[k]);
            if (tag_is_free(t, tag)) won = acas(&s->tag, tag, locked) == tag;     // lost: somebody else took it, look again next trip
        }
        const unsigned long long wm = __ballot(won);
        if (wm) {
            const int leader = __ffsll((long long)wm) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = aadd(&t.ctr->n_live, (unsigned long long)__popcll(wm));
            uint64_t* hw = reinterpret_cast<uint64_t*>(s);
            if (won) {                                             // key and identities go out while the reservation is in flight
#pragma unroll
                for (int k = 0; k < 5; k++) ast(&hw[1 + k], w[k]);
#pragma unroll
                for (int k = 6; k < 16; k++) ast(&hw[k], (uint64_t)0);
                ast(&t.cold[idx].smac_hi, (uint64_t)0);
                ast(&t.cold[idx].dmac_hi, (uint64_t)0);
            }
            base = __shfl(base, leader);
            if (won) {
                const unsigned long long pos = base + (unsigned long long)__popcll(wm & ((1ull << lane) - 1ull));
                ast(&t.live_list[pos & t.mask], (uint32_t)idx);
                drain_stores();
                ast(&s->tag, ready);
                result = (uint32_t)idx; done = true;
            }
        }
        if (!done && !won && !tag_is_free(t, tag)) {
            if (tag == ready) {
                bool eq = true;
#pragma unroll
                for (int k = 0; k < 5; k++) eq &= (kw[k] == w[k]);
                if (!eq) {                                       // possibly words older than the tag: once more, after it
                    eq = true;
#pragma unroll
                    for (int k = 0; k < 5; k++) eq &= (ald(&s->key[k]) == w[k]);
                }
                if (eq) { result = (uint32_t)idx; done = true; }
                else { idx = (idx + 1) & t.mask; probes++; }
            } else if (tag != locked) {                          // locked = same fingerprint, key not yet published: look again next trip
                idx = (idx + 1) & t.mask; probes++;
            }
            if (probes > t.mask) { atomicExch(&t.ctr->error, 1u); done = true; }
        }
    }
    return result;
}

// model.AccumulateBase + "first record stored whole" for one record: every operator unconditionally (no hints, see above)
NF_DEV void ep_merge_record(const TableView& t, uint32_t idx, const Rec& r, uint64_t seq) {
    Partial p;
    partial_from_record(r, seq, p);
    SlotHot* H = &t.hot[idx];
    SlotCold* C = &t.cold[idx];
    if (p.bytes) aadd(&H->bytes, p.bytes);
    if (p.packets) aadd(&H->packets, p.packets);
    if (p.flags) aor(&H->flags, p.flags);
    if (p.end) amax(&H->end, p.end);
    if (p.start_inv) amax(&H->start_inv, p.start_inv);
    if (p.eth_tag) amax(&H->eth_tag, p.eth_tag);
    if (p.dscp_tag) amax(&H->dscp_tag, p.dscp_tag);
    if (p.samp_tag) amax(&H->samp_tag, p.samp_tag);
    if (p.smac_inv) { amax(&H->smac_lo, tagged(p.smac_inv, (uint32_t)p.smac)); amax(&C->smac_hi, tagged(p.smac_inv, (uint32_t)(p.smac >> 32))); }
    if (p.dmac_inv) { amax(&H->dmac_lo, tagged(p.dmac_inv, (uint32_t)p.dmac)); amax(&C->dmac_hi, tagged(p.dmac_inv, (uint32_t)(p.dmac >> 32))); }
}

// Grid barrier on one monotone counter (the launch is cooperative: all 64 blocks are resident). cooperative_groups' grid.sync()
// measured ~10 us here; this one is an atomic add and a polling load (~3 us). The wait is bounded: a block that gives up raises
// `abort`, everybody leaves the kernel and the API reports the failure — never a hung GPU.
constexpr uint32_t kBarSpinLimit = 1u << 22;
NF_DEV bool grid_barrier(EpochCtl* c, unsigned long long& passed, uint32_t* lds_fail) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");             // this lane's atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        passed++;
        const unsigned long long target = passed * (unsigned long long)gridDim.x;
        aadd(&c->bar, 1ull);
        uint32_t spins = 0, fail_ = 0;
        while (ald(&c->bar) < target) {
            if (ald(&c->abort) || ++spins > kBarSpinLimit) { fail_ = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (fail_) ast(&c->abort, 1u);
        *lds_fail = fail_;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return *lds_fail == 0;
}

// Per-block LDS: the records of a block that share a key are represented by ONE of them towards the table. A Zipf stream sends
// ~12 % of a window to its hottest flow: 2000 lanes of the grid loading, CAS-ing and max-ing the same slot serialise at the
// memory side (measured: claim 65 us, fold 53 us + 90 us of draining per 14 k-record epoch). With the table below the hottest
// flow reaches HBM once per block and phase.
//   P1  an entry per key (64-bit hash, full key verified); its earliest lane claims the slot and plants ITS sequence number,
//       the others take the slot index from it
//   P4  the entry is the partial of its flow (LDS atomics, same operators as the table); one lane per entry merges it
constexpr int kEpEntries = 256, kEpProbe = 8;
struct EpLds {
    unsigned long long kh[kEpEntries];      // key hash | 1; 0 = free
    uint64_t key[5][kEpEntries];
    uint32_t rep[kEpEntries];               // lowest lane of the block with this key
    uint32_t sidx[kEpEntries];              // its slot
    unsigned long long bytes[kEpEntries], end[kEpEntries], start_inv[kEpEntries], eth[kEpEntries], dscp[kEpEntries], samp[kEpEntries];
    unsigned long long smac_lo[kEpEntries], smac_hi[kEpEntries], dmac_lo[kEpEntries], dmac_hi[kEpEntries];
    uint32_t packets[kEpEntries], flags[kEpEntries], touched[kEpEntries];
};

template <bool SKETCH>
__global__ __launch_bounds__(kEpBlock) void k_account_epochs(TableView t, SketchView sk, EpochArgs a) {
    __shared__ uint32_t wave_cnt[kEpBlock / 64];
    __shared__ uint32_t blk_base_s, total_s, bar_fail, split_s;
    __shared__ EpLds E;
    unsigned long long bars = 0;
#define EP_BARRIER() do { if (!grid_barrier(c, bars, &bar_fail)) return; } while (0)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t gid = blockIdx.x * kEpBlock + tid;
    EpochCtl* c = a.ctl;
    // replicated control state
    uint64_t pos = c->pos, seq0 = c->seq, live = c->live, list_base = c->list_base, list_fin = c->list_fin, out_pos = c->out_pos;
    uint32_t n_epochs = c->n_epochs, stop = 0;
    t.epoch_bits = c->epoch_bits;
    unsigned long long skipped = 0;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = wall_clock64();
#define EP_TICK(k) do { const unsigned long long tn_ = wall_clock64(); ph[k] += tn_ - tp; tp = tn_; } while (0)
    uint32_t win_cap = kEpWindow;
    uint64_t epoch_len = 0;                                        // records the epoch in progress has consumed inside this launch
    for (;;) {
        if (pos >= a.n) { stop = 1; break; }
        if (a.out_cap - out_pos < a.max_entries || n_epochs >= a.max_epochs) { stop = 2; break; }   // an eviction in this window could not be delivered
        if ((t.epoch_bits >> 48) >= 0xFFFFull) { stop = 3; break; }
        // records examined by this window: not more than an epoch has been taking (what lies beyond the split is claimed for nothing)
        const uint32_t window = (uint32_t)((a.n - pos) < (uint64_t)win_cap ? (a.n - pos) : (uint64_t)win_cap);
        const uint64_t room = a.max_entries > live ? a.max_entries - live : 0;
        // ---- P1: claim
        Rec r; uint64_t w[5]; uint64_t h = 0;
        uint32_t idx = kNoSlot;
        const bool in = gid < window;
        bool mine = false;
        if (in) {
            load_record(a.recs, pos + gid, r);
            r.canonicalize();
            r.key_words(w);
            h = key_hash(w);
            mine = !(t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id);
            if (!mine) skipped++;
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) w[k] = 0;
        }
        // the block's records with one key elect their earliest lane; only that lane goes to the table
        static_assert(kEpEntries == kEpBlock, "one LDS entry per lane");
        E.kh[tid] = 0; E.rep[tid] = 0xffffffffu; E.sidx[tid] = kNoSlot;
        E.bytes[tid] = 0; E.end[tid] = 0; E.start_inv[tid] = 0; E.eth[tid] = 0; E.dscp[tid] = 0; E.samp[tid] = 0;
        E.smac_lo[tid] = 0; E.smac_hi[tid] = 0; E.dmac_lo[tid] = 0; E.dmac_hi[tid] = 0; E.packets[tid] = 0; E.flags[tid] = 0; E.touched[tid] = 0;
        __syncthreads();
        int ent = -1;
        if (mine) {
            const unsigned long long hk = h | 1ull;
            uint32_t e = (uint32_t)(h >> 24) & (kEpEntries - 1);
#pragma unroll 1
            for (int p = 0; p < kEpProbe; p++) {
                unsigned long long cur = E.kh[e];
                if (cur == 0) {
                    cur = atomicCAS(&E.kh[e], 0ull, hk);
                    if (cur == 0) {
#pragma unroll
                        for (int k = 0; k < 5; k++) E.key[k][e] = w[k];
                        ent = (int)e; break;
                    }
                }
                if (cur == hk) { ent = (int)e; break; }
                e = (e + 1) & (kEpEntries - 1);
            }
        }
        __syncthreads();
        if (ent >= 0) {
            bool same = true;
#pragma unroll
            for (int k = 0; k < 5; k++) same &= (E.key[k][ent] == w[k]);
            if (same) atomicMin(&E.rep[ent], (uint32_t)tid);
            else ent = -1;                                          // another key with the same 64-bit hash: this lane goes alone
        }
        __syncthreads();
        const bool lead = mine && (ent < 0 || E.rep[ent] == (uint32_t)tid);
        idx = ep_find_or_claim(t, w, h, lead);                      // whole waves enter: the claim loop ballots
        if (lead && idx != kNoSlot) {
            amax(&t.hot[idx].id0, tagged(~(uint32_t)(seq0 + gid), r.d[21]));
            if (ent >= 0) E.sidx[ent] = idx;
        }
        __syncthreads();
        if (mine && !lead) idx = E.sidx[ent];
        EP_TICK(0);
        EP_BARRIER();
        EP_TICK(1);
        // ---- P2: first records of keys new to the map, counted per block
        const bool flag = idx != kNoSlot && (uint32_t)(ald(&t.hot[idx].id0) >> 32) == ~(uint32_t)(seq0 + gid);
        const unsigned long long fm = __ballot(flag);
        if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(fm);
        __syncthreads();
        uint32_t rank = (uint32_t)__popcll(fm & ((1ull << lane) - 1ull)), blk = 0;   // rank of a flagged lane inside its block
#pragma unroll
        for (int k = 0; k < kEpBlock / 64; k++) { if (k < wv) rank += wave_cnt[k]; blk += wave_cnt[k]; }
        if (tid == 0) ast(&c->block_count[blockIdx.x], blk);
        EP_TICK(2);
        EP_BARRIER();
        EP_TICK(3);
        // ---- P3: where does the map fill up? (every block computes the same prefix over the 64 counts)
        if (wv == 0) {
            const uint32_t cnt = ald(&c->block_count[lane]);
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            const uint32_t base_of_mine = __shfl(incl - cnt, blockIdx.x), tot = __shfl(incl, 63);
            if (lane == 0) { blk_base_s = base_of_mine; total_s = tot; }
        }
        __syncthreads();
        const uint32_t blk_base = blk_base_s, total_new = total_s;
        const bool full = (uint64_t)total_new > room;              // a record's NEW key finds len(entries) >= maxEntries (account.go:85)
        // The (room+1)-th new key of the window is the split. Only its own block needs the position now (blocks before it fold
        // all their records, blocks after it none); everybody else reads it after the barrier in front of the eviction.
        uint32_t fold_end = window;                                // this block folds its records with gid < fold_end
        if (full) {
            const bool split_here = (uint64_t)blk_base <= room && room < (uint64_t)blk_base + blk;
            if (flag && (uint64_t)(blk_base + rank) == room) { ast(&c->split, gid); split_s = gid; }   // exactly one lane of the grid
            __syncthreads();
            fold_end = split_here ? split_s : ((uint64_t)blk_base + blk <= room ? window : 0u);
        }
        EP_TICK(4);
        // ---- P4: fold the records before the split: into the block's entry of the flow, one merge per entry into the table
        if (in && gid < fold_end && idx != kNoSlot) {
            if (ent >= 0) {
                Partial p;
                partial_from_record(r, seq0 + gid, p);
                if (p.bytes) atomicAdd(&E.bytes[ent], (unsigned long long)p.bytes);
                if (p.packets) atomicAdd(&E.packets[ent], p.packets);
                if (p.flags) atomicOr(&E.flags[ent], p.flags);
                if (p.end) atomicMax(&E.end[ent], (unsigned long long)p.end);
                if (p.start_inv) atomicMax(&E.start_inv[ent], (unsigned long long)p.start_inv);
                if (p.eth_tag) atomicMax(&E.eth[ent], (unsigned long long)p.eth_tag);
                if (p.dscp_tag) atomicMax(&E.dscp[ent], (unsigned long long)p.dscp_tag);
                if (p.samp_tag) atomicMax(&E.samp[ent], (unsigned long long)p.samp_tag);
                if (p.smac_inv) { atomicMax(&E.smac_lo[ent], (unsigned long long)tagged(p.smac_inv, (uint32_t)p.smac));
                                  atomicMax(&E.smac_hi[ent], (unsigned long long)tagged(p.smac_inv, (uint32_t)(p.smac >> 32))); }
                if (p.dmac_inv) { atomicMax(&E.dmac_lo[ent], (unsigned long long)tagged(p.dmac_inv, (uint32_t)p.dmac));
                                  atomicMax(&E.dmac_hi[ent], (unsigned long long)tagged(p.dmac_inv, (uint32_t)(p.dmac >> 32))); }
                E.touched[ent] = 1;
            } else {
                ep_merge_record(t, idx, r, seq0 + gid);
            }
            if (SKETCH) sketch_add(sk, w, r.bytes());
        }
        __syncthreads();
        if (E.touched[tid]) {                                       // lane e merges entry e
            const uint32_t si = E.sidx[tid];
            SlotHot* H = &t.hot[si];
            SlotCold* Cc = &t.cold[si];
            if (E.bytes[tid]) aadd(&H->bytes, (uint64_t)E.bytes[tid]);
            if (E.packets[tid]) aadd(&H->packets, E.packets[tid]);
            if (E.flags[tid]) aor(&H->flags, E.flags[tid]);
            if (E.end[tid]) amax(&H->end, (uint64_t)E.end[tid]);
            if (E.start_inv[tid]) amax(&H->start_inv, (uint64_t)E.start_inv[tid]);
            if (E.eth[tid]) amax(&H->eth_tag, (uint64_t)E.eth[tid]);
            if (E.dscp[tid]) amax(&H->dscp_tag, (uint64_t)E.dscp[tid]);
            if (E.samp[tid]) amax(&H->samp_tag, (uint64_t)E.samp[tid]);
            if (E.smac_lo[tid]) { amax(&H->smac_lo, (uint64_t)E.smac_lo[tid]); amax(&Cc->smac_hi, (uint64_t)E.smac_hi[tid]); }
            if (E.dmac_lo[tid]) { amax(&H->dmac_lo, (uint64_t)E.dmac_lo[tid]); amax(&Cc->dmac_hi, (uint64_t)E.dmac_hi[tid]); }
        }
        EP_TICK(5);
        if (!full) {                                               // the epoch goes on
            live += total_new; pos += window; seq0 += window; epoch_len += window;
            continue;
        }
        EP_BARRIER();
        // ---- P5: Accounter.evict — every live flow whose first record precedes the split
        const uint32_t split = ald(&c->split);
        const uint64_t n_abs = ald(&t.ctr->n_live), split_seq = seq0 + split;
        for (uint64_t p = list_base + gid; ; p += kEpWindow) {      // wave-uniform trip count: the ballot below needs whole waves
            const bool have = p < n_abs;
            if (!__ballot(have)) break;
            uint32_t d[kRecordDwords];
            bool emit = false;
            if (have) {
                const uint32_t si = ald(&t.live_list[p & t.mask]);
                const uint64_t* hw = reinterpret_cast<const uint64_t*>(&t.hot[si]);
                uint64_t hq[16];
#pragma unroll
                for (int k = 1; k < 16; k++) hq[k] = ald(&hw[k]);
                const uint64_t smac_hi = ald(&t.cold[si].smac_hi), dmac_hi = ald(&t.cold[si].dmac_hi);   // with the hot words: one round trip
                const uint64_t id0 = hq[13];
                const uint32_t first_inv = (uint32_t)(id0 >> 32);
                emit = first_inv != 0 && (uint64_t)(~first_inv) < split_seq;   // slots claimed for keys first seen at or after the split die with the epoch
                if (emit) {
                    uint32_t ci[12];
                    if (p < list_fin) {                            // claimed and finalized before this launch
                        const uint4* cw = reinterpret_cast<const uint4*>(&t.cold[si]);
                        const uint4 v1 = cw[1], v2 = cw[2], v3 = cw[3];
                        ci[0] = v1.x; ci[1] = v1.y; ci[2] = v1.z; ci[3] = v1.w; ci[4] = v2.x; ci[5] = v2.y; ci[6] = v2.z; ci[7] = v2.w;
                        ci[8] = v3.x; ci[9] = v3.y; ci[10] = v3.z; ci[11] = v3.w;
                    } else {                                       // claimed in this launch: its first record is in the batch
                        const uint64_t ri = pos + (uint64_t)(uint32_t)(~first_inv) - seq0;   // first_seq may lie before seq0 (an earlier window of this epoch)
                        const uint4* rp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.recs) + ri * kRecordBytes);
                        const uint4 c5 = rp[5], c6 = rp[6], c7 = rp[7], c8 = rp[8];
                        ci[0] = c5.z; ci[1] = c6.x; ci[2] = c6.y; ci[3] = c6.z & 0x0000ffffu; ci[4] = c6.w; ci[5] = c7.x; ci[6] = c7.y; ci[7] = c7.z;
                        ci[8] = c7.w; ci[9] = c8.x; ci[10] = c8.y; ci[11] = c8.z;
                    }
#pragma unroll
                    for (int k = 0; k < 5; k++) { d[2 * k] = (uint32_t)hq[1 + k]; d[2 * k + 1] = (uint32_t)(hq[1 + k] >> 32); }
                    const uint64_t bytes = hq[6], end = hq[7], start_inv = hq[8], pf = hq[9], eth_tag = hq[10], dscp_tag = hq[11],
                                   samp_tag = hq[12], smac_lo = hq[14], dmac_lo = hq[15];
                    const uint64_t start = start_inv ? ~start_inv : 0ull;
                    d[10] = (uint32_t)start; d[11] = (uint32_t)(start >> 32);
                    d[12] = (uint32_t)end; d[13] = (uint32_t)(end >> 32);
                    d[14] = (uint32_t)bytes; d[15] = (uint32_t)(bytes >> 32);
                    d[16] = (uint32_t)pf;
                    d[17] = (uint32_t)(eth_tag & 0xffffu) | (((uint32_t)(pf >> 32) & 0xffffu) << 16);
                    const uint64_t smac = (uint64_t)(uint32_t)smac_lo | ((uint64_t)(smac_hi & 0xffffu) << 32);
                    const uint64_t dmac = (uint64_t)(uint32_t)dmac_lo | ((uint64_t)(dmac_hi & 0xffffu) << 32);
                    d[18] = (uint32_t)smac;
                    d[19] = (uint32_t)((smac >> 32) & 0xffffu) | (uint32_t)((dmac & 0xffffu) << 16);
                    d[20] = (uint32_t)(dmac >> 16);
                    d[21] = (uint32_t)id0;
                    d[22] = ci[0];
                    d[23] = (uint32_t)samp_tag;
#pragma unroll
                    for (int k = 1; k < 12; k++) d[23 + k] = ci[k];
                    d[24] = (d[24] & 0xff00ffffu) | ((uint32_t)(dscp_tag & 0xffu) << 16);
                    d[35] = 0;
                }
            }
            const unsigned long long em = __ballot(emit);
            unsigned long long at = 0;
            if (lane == 0 && em) at = aadd(&c->out_cursor, (unsigned long long)__popcll(em));
            at = __shfl(at, 0);
            if (emit) {
                uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.out) + (at + (unsigned long long)__popcll(em & ((1ull << lane) - 1ull))) * kRecordBytes);
#pragma unroll
                for (int k = 0; k < 9; k++) o[k] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
            }
        }
        EP_TICK(6);
        EP_BARRIER();                                              // the evicted slots may be claimed again from here on
        EP_TICK(7);
        // ---- the next epoch starts AT the split: that record is inserted into the empty map (account.go:95)
        out_pos += live + room;                                    // = max_entries: len(entries) when the map was found full
        if (gid == 0) a.epoch_end[n_epochs] = out_pos;
        n_epochs++;
        list_base = n_abs; list_fin = n_abs;
        t.epoch_bits += 1ull << 48;
        live = 0; pos += split; seq0 = 0;
        epoch_len += split;
        {   // next epoch: a window of about 1.25 x what this one took, whole waves, at least 512 records
            uint64_t wc = epoch_len + epoch_len / 4 + 64;
            wc = (wc + 63) & ~63ull;
            if (wc < 512) wc = 512;
            win_cap = wc > (uint64_t)kEpWindow ? kEpWindow : (uint32_t)wc;
        }
        epoch_len = 0;
    }
    if (gid == 0) {
        c->pos = pos; c->seq = seq0; c->live = live; c->list_base = list_base; c->list_fin = list_fin; c->out_pos = out_pos;
        c->epoch_bits = t.epoch_bits; c->n_epochs = n_epochs; c->stop = stop;
        for (int k = 0; k < 8; k++) c->phase[k] = ph[k];
    }
#undef EP_TICK
#undef EP_BARRIER
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
}

// After the launch: the epoch in progress occupies ring positions [base, base + cnt): move it to the front of the live list so
// that every other kernel finds it where it expects it. n_finalized: how many of these slots have their identity dwords.
__global__ __launch_bounds__(256) void k_ring_to_front(TableView t, uint64_t base, uint64_t cnt, uint32_t* __restrict__ tmp, int phase) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += stride) {
        if (phase == 0) tmp[k] = t.live_list[(base + k) & t.mask];
        else t.live_list[k] = tmp[k];
    }
}
__global__ void k_ring_counters(DevCounters* c, unsigned long long n_live, unsigned long long n_finalized) { c->n_live = n_live; c->n_finalized = n_finalized; }

size_t epoch_ctl_bytes() { return sizeof(EpochCtl); }
uint32_t epoch_window() { return kEpWindow; }

// ctl fields as plain numbers for the API (nfagg_api.hip does not see the struct)
void epoch_ctl_fill(void* h_ctl, uint64_t seq, uint64_t live, uint64_t list_base, uint64_t list_fin, uint64_t epoch_bits) {
    EpochCtl* c = static_cast<EpochCtl*>(h_ctl);
    memset(c, 0, sizeof *c);
    c->seq = seq; c->live = live; c->list_base = list_base; c->list_fin = list_fin; c->epoch_bits = epoch_bits;
}
void epoch_ctl_phases(const void* h_ctl, uint64_t out[8]) {
    const EpochCtl* c = static_cast<const EpochCtl*>(h_ctl);
    for (int k = 0; k < 8; k++) out[k] = c->phase[k];
}
void epoch_ctl_read(const void* h_ctl, uint64_t out[9]) {
    const EpochCtl* c = static_cast<const EpochCtl*>(h_ctl);
    out[0] = c->pos; out[1] = c->seq; out[2] = c->live; out[3] = c->list_base; out[4] = c->list_fin; out[5] = c->out_pos;
    out[6] = c->epoch_bits; out[7] = c->n_epochs; out[8] = c->stop;
}

hipError_t launch_account_epochs(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, void* d_out, uint64_t out_cap,
                                 uint64_t* d_epoch_end, uint32_t max_epochs, uint64_t max_entries, void* d_ctl, hipStream_t s) {
    EpochArgs a{d_records, n, d_out, out_cap, d_epoch_end, max_epochs, max_entries, static_cast<EpochCtl*>(d_ctl)};
    TableView tv = t;
    SketchView skv = sk;
    void* args[3] = {&tv, &skv, &a};
    const void* fn = sk.flags ? reinterpret_cast<const void*>(&k_account_epochs<true>) : reinterpret_cast<const void*>(&k_account_epochs<false>);
    return hipLaunchCooperativeKernel(fn, dim3(kEpGrid), dim3(kEpBlock), args, 0, s);
}

hipError_t launch_ring_to_front(const TableView& t, uint64_t base, uint64_t cnt, uint64_t n_finalized, uint32_t* d_tmp, hipStream_t s) {
    (void)hipGetLastError();
    if (cnt && (base & t.mask) != 0) {
        const int grid = (int)((cnt + 255) / 256 > 1024 ? 1024 : (cnt + 255) / 256);
        hipLaunchKernelGGL(k_ring_to_front, dim3(grid), dim3(256), 0, s, t, base, cnt, d_tmp, 0);
        hipLaunchKernelGGL(k_ring_to_front, dim3(grid), dim3(256), 0, s, t, base, cnt, d_tmp, 1);
    }
    hipLaunchKernelGGL(k_ring_counters, dim3(1), dim3(1), 0, s, t.ctr, (unsigned long long)cnt, (unsigned long long)n_finalized);
    return hipGetLastError();
}

}  // namespace nfagg
