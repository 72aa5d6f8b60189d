// nfagg_epoch_chain.hip — the evict-on-full loop of Accounter.Account (pkg/flow/account.go:81-96) as a CHAIN of four small
// kernels per window that read their arguments from a control block in device memory, enqueued by the host many windows at
// a time with no round trip in between (nfagg_account[_device] for small CACHE_MAX_FLOWS; the default form).
//
// Same phases as the persistent kernel of nfagg_epochs.hip (which stays as ingest_variant 30, for comparison):
//   K1 k_ep_claim   one record per lane of the window; the block's records with one key elect their earliest lane (LDS), that lane
//                   finds or claims the slot and plants its sequence number as candidate first record; slot index per record
//   K2 k_ep_flags   first records of keys new to the map, counted per block; the LAST block to finish prefixes the 64 counts,
//                   decides "full" (account.go:85) and finds the split inside the block that holds the (room+1)-th new key
//   K3 k_ep_fold    records before the split, pre-combined per slot in LDS, one merge per (block, flow)
//   K4 k_ep_evict   on full: Accounter.evict (account.go:102-124) of every flow whose first record precedes the split; its last
//                   block bumps the epoch tag, resets the live list and moves the control block to the next window (which
//                   starts AT the split: that record is inserted into the empty map, account.go:95)
// Why a chain beats the one persistent kernel here: a dependent kernel boundary costs ~1.5 us on this chip, a grid-wide
// barrier across the 8 XCDs ~5 us (measured; MI355X guide: boundary 1.45, barrier-counter 7.4), and behind a kernel boundary
// the table may be read with plain, cached 16-byte loads (probe_home, hints) where the persistent kernel must use 8-byte
// agent-scope accesses for every shared word: 71 -> ~30 us per 14 k-record epoch. The host enqueues kChainBatch windows
// (4 launches each, ~3.5 us of host time per launch, asynchronous) and then reads the control block once; kernels of windows
// behind the end of the batch find `stop` set and return at once.
#include <string.h>
#include "nfagg_device.h"

namespace nfagg {

constexpr int kCkBlock = 256, kCkGrid = 64;
constexpr uint32_t kCkWindow = kCkBlock * kCkGrid;

struct ChainCtl {
    // written by the host before the first launch, advanced by K4's last block, read back by the host
    unsigned long long pos, n;          // records of the batch consumed / in the batch
    unsigned long long seq;             // window-relative sequence number of record `pos`
    unsigned long long live;            // len(entries)
    unsigned long long list_fin;        // live-list positions below it were finalized before this call (identity in the cold line)
    unsigned long long out_pos, out_cap;// records written to / room in the output
    unsigned long long epoch_bits;
    unsigned long long max_entries;
    unsigned long long epoch_first;     // batch index of the first record of the epoch in progress (if it began inside this call)
    const void* recs;                   // the batch and the output: in the control block so that the kernels' ARGUMENTS never change
    void* out;                          // from call to call and the chain can be replayed as one captured hipGraph
    uint32_t n_epochs, max_epochs;
    uint32_t stop;                      // 1 batch consumed, 2 no room for another eviction / epoch list full, 3 epoch tags wrap next
    uint32_t window;                    // records of the current window
    uint32_t win_cap;                   // adaptive: about 1.25 x the last epoch
    uint32_t epoch_began_here;          // the epoch in progress began inside this call
    // per window
    uint32_t full, split, total_new, pad0;
    unsigned long long epoch_len;       // records the epoch in progress has consumed inside this call
    unsigned long long n_out;           // evict: positions handed out
    uint32_t ticket[4];                 // last-block tickets of K2 / K4
    uint32_t block_count[kCkGrid];
};

NF_DEV uint32_t ck_next_window(const ChainCtl* c) {
    const unsigned long long left = c->n - c->pos;
    return (uint32_t)(left < (unsigned long long)c->win_cap ? left : (unsigned long long)c->win_cap);
}

// set `stop` when the next window must not start: nothing left, no room to deliver an eviction, the epoch tags would wrap
NF_DEV void ck_check_stop(ChainCtl* c) {
    if (c->pos >= c->n) c->stop = 1;
    else if (c->out_cap - c->out_pos < c->max_entries || c->n_epochs >= c->max_epochs) c->stop = 2;
    else if ((c->epoch_bits >> 48) >= 0xFFFFull) c->stop = 3;
    c->window = c->stop ? 0u : ck_next_window(c);
}

struct CkLds {
    unsigned long long kh[kCkBlock];
    uint64_t key[5][kCkBlock];
    uint32_t rep[kCkBlock], sidx[kCkBlock];
};

// ---- K1
__global__ __launch_bounds__(kCkBlock) void k_ep_claim(TableView t, ChainCtl* c, uint32_t* __restrict__ slot_idx) {
    if (c->stop) return;
    const void* recs = c->recs;
    __shared__ CkLds E;
    const int tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * kCkBlock + tid, window = c->window;
    const uint64_t pos = c->pos, seq0 = c->seq;
    t.epoch_bits = c->epoch_bits;
    Rec r; uint64_t w[5]; uint64_t h = 0;
    bool mine = false;
    if (gid < window) {
        load_record(recs, pos + gid, r);
        r.canonicalize();
        r.key_words(w);
        h = key_hash(w);
        mine = !(t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id);
        if (!mine) aadd(&t.ctr->n_skipped, 1ull);
    }
    E.kh[tid] = 0; E.rep[tid] = 0xffffffffu; E.sidx[tid] = kNoSlot;
    __syncthreads();
    int ent = -1;
    if (mine) {
        const unsigned long long hk = h | 1ull;
        uint32_t e = (uint32_t)(h >> 24) & (kCkBlock - 1);
#pragma unroll 1
        for (int p = 0; p < 8; p++) {
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
            e = (e + 1) & (kCkBlock - 1);
        }
    }
    __syncthreads();
    if (ent >= 0) {
        bool same = true;
#pragma unroll
        for (int k = 0; k < 5; k++) same &= (E.key[k][ent] == w[k]);
        if (same) atomicMin(&E.rep[ent], (uint32_t)tid);
        else ent = -1;                                            // another key with the same 64-bit hash: this lane goes alone
    }
    __syncthreads();
    const bool lead = mine && (ent < 0 || E.rep[ent] == (uint32_t)tid);
    uint32_t idx = kNoSlot;
    if (lead) {
        Hints x;
        idx = probe_home(t, w, h, x);
        if (idx == kNoSlot) { idx = find_or_claim(t, w, h); x.id0 = 0; }
        if (idx != kNoSlot) {
            const uint64_t my0 = tagged(~(uint32_t)(seq0 + gid), r.d[21]);
            if (x.id0 < my0) amax(&t.hot[idx].id0, my0);          // a stale hint is a lower bound: at worst one atomic too many
            if (ent >= 0) E.sidx[ent] = idx;
        }
    }
    __syncthreads();
    if (mine && !lead) idx = E.sidx[ent];
    if (gid < window) slot_idx[gid] = idx;
}

// ---- K2
__global__ __launch_bounds__(kCkBlock) void k_ep_flags(TableView t, ChainCtl* c, const uint32_t* __restrict__ slot_idx) {
    if (c->stop) return;
    __shared__ uint32_t wave_cnt[kCkBlock / 64], last_s, split_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t gid = blockIdx.x * kCkBlock + tid, window = c->window;
    const uint64_t seq0 = c->seq;
    auto flag_of = [&](uint32_t g) -> bool {
        if (g >= window) return false;
        const uint32_t idx = slot_idx[g];
        return idx != kNoSlot && (uint32_t)(t.hot[idx].id0 >> 32) == ~(uint32_t)(seq0 + g);
    };
    const bool flag = flag_of(gid);
    const unsigned long long fm = __ballot(flag);
    if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(fm);
    __syncthreads();
    if (tid == 0) {
        uint32_t blk = 0;
        for (int k = 0; k < kCkBlock / 64; k++) blk += wave_cnt[k];
        ast(&c->block_count[blockIdx.x], blk);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        last_s = aadd(&c->ticket[0], 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last_s) return;
    // ---- the last block: where does the map fill up?
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint64_t room = c->max_entries > c->live ? c->max_entries - c->live : 0;
    uint32_t cnt = 0, incl = 0;
    if (wv == 0) {
        cnt = ald(&c->block_count[lane]);
        incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    }
    __shared__ uint32_t total_s, sb_s, sbase_s;
    if (wv == 0) {
        const uint32_t total = __shfl(incl, 63);
        // the block that holds the (room+1)-th new key
        const bool here = (uint64_t)(incl - cnt) <= room && room < (uint64_t)incl;
        const unsigned long long hm = __ballot(here);
        if (lane == 0) { total_s = total; sb_s = hm ? (uint32_t)(__ffsll((long long)hm) - 1) : 0xffffffffu; }
        if (here) sbase_s = incl - cnt;
    }
    __syncthreads();
    const bool full = (uint64_t)total_s > room;
    if (full) {                                                   // rank the flagged lanes of that block again: the split position
        const uint32_t g = sb_s * kCkBlock + tid;
        const bool f = flag_of(g);
        const unsigned long long m2 = __ballot(f);
        if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m2);
        __syncthreads();
        uint32_t rank = (uint32_t)__popcll(m2 & ((1ull << lane) - 1ull));
        for (int k = 0; k < wv; k++) rank += wave_cnt[k];
        if (f && (uint64_t)(sbase_s + rank) == room) split_s = g;
        __syncthreads();
    }
    if (tid == 0) {
        c->full = full ? 1u : 0u;
        c->split = full ? split_s : window;
        c->total_new = total_s;
        c->ticket[0] = 0;
    }
}

// ---- K3: fold the records before the split; per block one merge per slot
struct CkFold {
    uint32_t sidx[kCkBlock];                 // slot of the entry, kNoSlot = free
    unsigned long long bytes[kCkBlock], end[kCkBlock], start_inv[kCkBlock], eth[kCkBlock], dscp[kCkBlock], samp[kCkBlock];
    unsigned long long smac_lo[kCkBlock], smac_hi[kCkBlock], dmac_lo[kCkBlock], dmac_hi[kCkBlock];
    uint32_t packets[kCkBlock], flags[kCkBlock];
};

template <bool SKETCH>
__global__ __launch_bounds__(kCkBlock) void k_ep_fold(TableView t, SketchView sk, ChainCtl* c, const uint32_t* __restrict__ slot_idx) {
    if (c->stop) return;
    const void* recs = c->recs;
    __shared__ CkFold F;
    const int tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * kCkBlock + tid, split = c->split;
    const uint64_t pos = c->pos, seq0 = c->seq;
    F.sidx[tid] = kNoSlot;
    F.bytes[tid] = 0; F.end[tid] = 0; F.start_inv[tid] = 0; F.eth[tid] = 0; F.dscp[tid] = 0; F.samp[tid] = 0;
    F.smac_lo[tid] = 0; F.smac_hi[tid] = 0; F.dmac_lo[tid] = 0; F.dmac_hi[tid] = 0; F.packets[tid] = 0; F.flags[tid] = 0;
    __syncthreads();
    if (gid < split) {
        const uint32_t idx = slot_idx[gid];
        if (idx != kNoSlot) {
            Rec r;
            load_record(recs, pos + gid, r);
            r.canonicalize();
            Partial p;
            partial_from_record(r, seq0 + gid, p);
            int ent = -1;
            uint32_t e = (idx * 2654435761u >> 24) & (kCkBlock - 1);
#pragma unroll 1
            for (int q = 0; q < 8; q++) {
                uint32_t cur = F.sidx[e];
                if (cur == kNoSlot) cur = atomicCAS(&F.sidx[e], kNoSlot, idx);
                if (cur == kNoSlot || cur == idx) { ent = (int)e; break; }
                e = (e + 1) & (kCkBlock - 1);
            }
            if (ent >= 0) {
                if (p.bytes) atomicAdd(&F.bytes[ent], (unsigned long long)p.bytes);
                if (p.packets) atomicAdd(&F.packets[ent], p.packets);
                if (p.flags) atomicOr(&F.flags[ent], p.flags);
                if (p.end) atomicMax(&F.end[ent], (unsigned long long)p.end);
                if (p.start_inv) atomicMax(&F.start_inv[ent], (unsigned long long)p.start_inv);
                if (p.eth_tag) atomicMax(&F.eth[ent], (unsigned long long)p.eth_tag);
                if (p.dscp_tag) atomicMax(&F.dscp[ent], (unsigned long long)p.dscp_tag);
                if (p.samp_tag) atomicMax(&F.samp[ent], (unsigned long long)p.samp_tag);
                if (p.smac_inv) { atomicMax(&F.smac_lo[ent], (unsigned long long)tagged(p.smac_inv, (uint32_t)p.smac));
                                  atomicMax(&F.smac_hi[ent], (unsigned long long)tagged(p.smac_inv, (uint32_t)(p.smac >> 32))); }
                if (p.dmac_inv) { atomicMax(&F.dmac_lo[ent], (unsigned long long)tagged(p.dmac_inv, (uint32_t)p.dmac));
                                  atomicMax(&F.dmac_hi[ent], (unsigned long long)tagged(p.dmac_inv, (uint32_t)(p.dmac >> 32))); }
            } else {                                              // the block's table is full for this slot: merge the record itself
                Hints x;
                load_hints(&t.hot[idx], x);
                p.first_inv = 0;                                  // id0 was planted by K1
                merge_partial(t, idx, p, x);
            }
            if (SKETCH) { uint64_t w[5]; r.key_words(w); sketch_add(sk, w, r.bytes()); }
        }
    }
    __syncthreads();
    const uint32_t si = F.sidx[tid];
    if (si != kNoSlot) {                                          // lane e merges entry e
        SlotHot* H = &t.hot[si];
        SlotCold* Cc = &t.cold[si];
        Hints x;
        load_hints(H, x);
        if (F.bytes[tid]) aadd(&H->bytes, (uint64_t)F.bytes[tid]);
        if (F.packets[tid]) aadd(&H->packets, F.packets[tid]);
        if (F.flags[tid] & ~x.flags) aor(&H->flags, F.flags[tid]);
        if (F.end[tid] > x.end) amax(&H->end, (uint64_t)F.end[tid]);
        if (F.start_inv[tid] > x.start_inv) amax(&H->start_inv, (uint64_t)F.start_inv[tid]);
        if (F.eth[tid]) amax(&H->eth_tag, (uint64_t)F.eth[tid]);
        if (F.dscp[tid]) amax(&H->dscp_tag, (uint64_t)F.dscp[tid]);
        if (F.samp[tid]) amax(&H->samp_tag, (uint64_t)F.samp[tid]);
        if (F.smac_lo[tid] && x.smac_lo <= F.smac_lo[tid]) { amax(&H->smac_lo, (uint64_t)F.smac_lo[tid]); amax(&Cc->smac_hi, (uint64_t)F.smac_hi[tid]); }
        if (F.dmac_lo[tid] && x.dmac_lo <= F.dmac_lo[tid]) { amax(&H->dmac_lo, (uint64_t)F.dmac_lo[tid]); amax(&Cc->dmac_hi, (uint64_t)F.dmac_hi[tid]); }
    }
}

// ---- K4: on full Accounter.evict; always: advance the control block (last block)
__global__ __launch_bounds__(kCkBlock) void k_ep_evict(TableView t, ChainCtl* c, uint64_t* __restrict__ epoch_end) {
    if (c->stop) return;
    const void* recs = c->recs;
    void* out = c->out;
    __shared__ uint32_t last_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t gid = blockIdx.x * kCkBlock + tid;
    const bool full = c->full != 0;
    const uint64_t pos = c->pos, seq0 = c->seq;
    const uint64_t n_live = t.ctr->n_live;
    if (full) {
        const uint64_t split_seq = seq0 + c->split, list_fin = c->list_fin, out_pos = c->out_pos;
        for (uint64_t p = gid; ; p += kCkWindow) {
            const bool have = p < n_live;
            if (!__ballot(have)) break;
            uint32_t d[kRecordDwords];
            bool emit = false;
            if (have) {
                const uint32_t si = t.live_list[p];
                const uint4* L = reinterpret_cast<const uint4*>(&t.hot[si]);
                uint4 hv[8];
#pragma unroll
                for (int k = 0; k < 8; k++) hv[k] = L[k];
                const uint4* cw = reinterpret_cast<const uint4*>(&t.cold[si]);
                const uint4 c0 = cw[0];
                const uint32_t first_inv = hv[6].w;
                emit = first_inv != 0 && (uint64_t)(~first_inv) < split_seq;   // slots claimed for keys first seen at or after the split die with the epoch
                if (emit) {
                    uint32_t ci[12];
                    if (p < list_fin) {                            // finalized before this call
                        const uint4 v1 = cw[1], v2 = cw[2], v3 = cw[3];
                        ci[0] = v1.x; ci[1] = v1.y; ci[2] = v1.z; ci[3] = v1.w; ci[4] = v2.x; ci[5] = v2.y; ci[6] = v2.z; ci[7] = v2.w;
                        ci[8] = v3.x; ci[9] = v3.y; ci[10] = v3.z; ci[11] = v3.w;
                    } else {                                       // claimed in this call: its first record is in the batch
                        const uint64_t ri = pos + (uint64_t)(uint32_t)(~first_inv) - seq0;
                        const uint4* rp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + ri * kRecordBytes);
                        const uint4 c5 = rp[5], c6 = rp[6], c7 = rp[7], c8 = rp[8];
                        ci[0] = c5.z; ci[1] = c6.x; ci[2] = c6.y; ci[3] = c6.z & 0x0000ffffu; ci[4] = c6.w; ci[5] = c7.x; ci[6] = c7.y; ci[7] = c7.z;
                        ci[8] = c7.w; ci[9] = c8.x; ci[10] = c8.y; ci[11] = c8.z;
                    }
                    auto q64 = [](const uint4& v, int hi) -> uint64_t { return hi ? ((uint64_t)v.z | ((uint64_t)v.w << 32)) : ((uint64_t)v.x | ((uint64_t)v.y << 32)); };
                    d[0] = hv[0].z; d[1] = hv[0].w; d[2] = hv[1].x; d[3] = hv[1].y; d[4] = hv[1].z; d[5] = hv[1].w;
                    d[6] = hv[2].x; d[7] = hv[2].y; d[8] = hv[2].z; d[9] = hv[2].w;
                    const uint64_t bytes = q64(hv[3], 0), end = q64(hv[3], 1), start_inv = q64(hv[4], 0), eth_tag = q64(hv[5], 0),
                                   dscp_tag = q64(hv[5], 1), samp_tag = q64(hv[6], 0), id0 = q64(hv[6], 1), smac_lo = q64(hv[7], 0), dmac_lo = q64(hv[7], 1);
                    const uint64_t start = start_inv ? ~start_inv : 0ull;
                    d[10] = (uint32_t)start; d[11] = (uint32_t)(start >> 32);
                    d[12] = (uint32_t)end; d[13] = (uint32_t)(end >> 32);
                    d[14] = (uint32_t)bytes; d[15] = (uint32_t)(bytes >> 32);
                    d[16] = hv[4].z;
                    d[17] = (uint32_t)(eth_tag & 0xffffu) | ((hv[4].w & 0xffffu) << 16);
                    const uint64_t smac = (uint64_t)(uint32_t)smac_lo | ((uint64_t)(c0.x & 0xffffu) << 32);
                    const uint64_t dmac = (uint64_t)(uint32_t)dmac_lo | ((uint64_t)(c0.z & 0xffffu) << 32);
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
            if (lane == 0 && em) at = aadd(&c->n_out, (unsigned long long)__popcll(em));
            at = __shfl(at, 0);
            if (emit) {
                uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + (out_pos + at + (unsigned long long)__popcll(em & ((1ull << lane) - 1ull))) * kRecordBytes);
#pragma unroll
                for (int k = 0; k < 9; k++) o[k] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
            }
        }
    }
    // ---- the last block moves on to the next window
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        last_s = aadd(&c->ticket[1], 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last_s || tid != 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    c->ticket[1] = 0;
    if (full) {
        const unsigned long long wrote = ald(&c->n_out);
        c->out_pos += wrote; c->n_out = 0;
        epoch_end[c->n_epochs] = c->out_pos;
        c->n_epochs++;
        c->epoch_bits += 1ull << 48;
        t.ctr->n_live = 0; t.ctr->n_finalized = 0; t.ctr->aborted = 0; t.ctr->max_probe = 0;
        c->list_fin = 0; c->live = 0;
        c->epoch_len += c->split;
        {   // next epoch: a window of about 1.25 x what this one took, whole waves, at least 512 records
            unsigned long long wc = c->epoch_len + c->epoch_len / 4 + 64;
            wc = (wc + 63) & ~63ull;
            if (wc < 512) wc = 512;
            c->win_cap = wc > (unsigned long long)kCkWindow ? kCkWindow : (uint32_t)wc;
        }
        c->epoch_len = 0;
        c->pos += c->split; c->seq = 0;
        c->epoch_first = c->pos; c->epoch_began_here = 1;
    } else {
        c->live += c->total_new; c->pos += c->window; c->seq += c->window; c->epoch_len += c->window;
    }
    ck_check_stop(c);
}

size_t chain_ctl_bytes() { return sizeof(ChainCtl); }
uint32_t chain_window() { return kCkWindow; }

void chain_ctl_fill(void* h_ctl, const void* d_records, void* d_out, uint64_t n, uint64_t seq, uint64_t live, uint64_t list_fin, uint64_t out_cap,
                    uint64_t epoch_bits, uint64_t max_entries, uint32_t max_epochs) {
    ChainCtl* c = static_cast<ChainCtl*>(h_ctl);
    memset(c, 0, sizeof *c);
    c->recs = d_records; c->out = d_out;
    c->n = n; c->seq = seq; c->live = live; c->list_fin = list_fin; c->out_cap = out_cap; c->epoch_bits = epoch_bits;
    c->max_entries = max_entries; c->max_epochs = max_epochs; c->win_cap = kCkWindow;
    // the first window (what ck_check_stop does on the device)
    if (c->pos >= c->n) c->stop = 1;
    else if (c->out_cap - c->out_pos < c->max_entries || c->n_epochs >= c->max_epochs) c->stop = 2;
    else if ((c->epoch_bits >> 48) >= 0xFFFFull) c->stop = 3;
    c->window = c->stop ? 0u : (uint32_t)(n < (uint64_t)kCkWindow ? n : (uint64_t)kCkWindow);
}
// -> {pos, seq, live, out_pos, epoch_bits, n_epochs, stop, epoch_first, epoch_began_here}
void chain_ctl_read(const void* h_ctl, uint64_t out[9]) {
    const ChainCtl* c = static_cast<const ChainCtl*>(h_ctl);
    out[0] = c->pos; out[1] = c->seq; out[2] = c->live; out[3] = c->out_pos; out[4] = c->epoch_bits; out[5] = c->n_epochs; out[6] = c->stop;
    out[7] = c->epoch_first; out[8] = c->epoch_began_here;
}

// One window = four launches; every argument is the same from call to call (batch, output and all positions live in the control
// block), so the caller may capture a run of windows into a hipGraph once and replay it.
hipError_t launch_epoch_chain_window(const TableView& t, const SketchView& sk, void* d_ctl, uint32_t* d_slot_idx, uint64_t* d_epoch_end, hipStream_t s) {
    ChainCtl* c = static_cast<ChainCtl*>(d_ctl);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_ep_claim, dim3(kCkGrid), dim3(kCkBlock), 0, s, t, c, d_slot_idx);
    hipLaunchKernelGGL(k_ep_flags, dim3(kCkGrid), dim3(kCkBlock), 0, s, t, c, (const uint32_t*)d_slot_idx);
    if (sk.flags) hipLaunchKernelGGL(k_ep_fold<true>, dim3(kCkGrid), dim3(kCkBlock), 0, s, t, sk, c, (const uint32_t*)d_slot_idx);
    else hipLaunchKernelGGL(k_ep_fold<false>, dim3(kCkGrid), dim3(kCkBlock), 0, s, t, sk, c, (const uint32_t*)d_slot_idx);
    hipLaunchKernelGGL(k_ep_evict, dim3(kCkGrid), dim3(kCkBlock), 0, s, t, c, d_epoch_end);
    return hipGetLastError();
}

}  // namespace nfagg
