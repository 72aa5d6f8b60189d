// nfagg_ingest_part.hip — two-pass partitioned ingest (the default from 384 Ki records per call; ingest_variant 10 forces it).
//
// The single-pass cached kernel (nfagg_ingest_cached.hip) folds the hot head of a
// Zipf stream in LDS, but every record of the cold tail (40 % of configs[1]) goes to
// the HBM table one by one: 4-5 agent-scope atomics each, and the chip retires only
// ~24 G random atomics/s (profiles/r01_ubench_atomics.txt) — that, not HBM
// bandwidth, bounds the kernel. Here the tail is not merged record by record:
//
//   pass 1  k_pass1            streams the batch once. Hot flows are folded in the
//           workgroup's LDS flow cache exactly as before; a record whose flow gets no
//           cache entry is SPILLED: its 32-bit index is appended to the queue of its
//           partition (2048 partitions = the top bits of its home slot index), staged four at a time in LDS so a
//           spill costs one 16-byte store and a quarter of an atomic.
//   pass 2  k_pass2            one workgroup per partition gathers the spilled
//           records by index and folds them in ITS LDS cache. A partition holds
//           ~1/2048 of the flows, so (almost) every flow of it gets an entry.
//   Both passes end by merging each cache entry into the table ONCE.
//
// Exactness (pkg/model/flow_content.go:28-61, pkg/flow/account.go:95) is unchanged:
// a cache entry is a partial (nfagg_device.h) carrying sequence-tagged fields, and
// merging partials is associative and commutative. "First record" data (and the first
// non-zero MACs) never enter the cache: an entry only tracks the smallest sequence
// numbers; the flush gathers the MACs from the batch (still in HBM — the caller owns it
// until the call returns) and publishes the first record's sequence number, whose
// identity dwords k_finalize (nfagg_kernels.hip) copies from the batch afterwards.
#include <atomic>
#include "nfagg_device.h"

namespace nfagg {
namespace part {

constexpr int kBlock = 1024;
constexpr int kEntries = 1024;
#ifndef NF_PART_PROBE                // (settable on the compiler's command line: tools/gpu/r06_probe_sweep.sh)
#define NF_PART_PROBE 8
#endif
constexpr int kProbe = NF_PART_PROBE;   // cache probe window
constexpr int kStage = 4;         // staged spills per partition = one 16-byte store

// 116 bytes per entry, laid out as arrays of 16-BYTE UNITS so that phase B reads an entry's key with three ds_read_b128 and its
// guard words (what decides which atomics can change anything) with two, instead of eleven 8- and 4-byte reads (round 2's
// field-per-array layout): the fold loops are bound by LDS instructions issued, not by LDS bytes.
struct Cache {
    uint4 k0[kEntries];           // x,y: key hash | 1 (0 = free)   z,w: key word 0
    uint4 k1[kEntries];           // key words 1, 2
    uint4 k2[kEntries];           // key words 3, 4
    uint4 t[kEntries];            // x,y: end (max)                 z,w: start_inv (max of ~start)
    uint4 q[kEntries];            // x: flags (OR)  y: first_seq  z: smac_seq  w: dmac_seq — min seq32 of the records folded into the entry
                                  //    (all of them / those with a non-zero src_mac / dst_mac)
    uint4 v[kEntries];            // x,y: bytes (sum)               z,w: samp_tag
    uint4 w[kEntries];            // x,y: eth_tag                   z,w: dscp_tag
    uint32_t packets[kEntries];
};
static_assert(sizeof(Cache) == 116 * kEntries, "cache entry");
NF_DEV uint64_t u64lo(const uint4& a) { return (uint64_t)a.x | ((uint64_t)a.y << 32); }
NF_DEV uint64_t u64hi(const uint4& a) { return (uint64_t)a.z | ((uint64_t)a.w << 32); }
NF_DEV uint4 mk4(uint64_t lo, uint64_t hi) { return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)); }
NF_DEV unsigned long long* lo64(uint4* a) { return reinterpret_cast<unsigned long long*>(a); }
NF_DEV unsigned long long* hi64(uint4* a) { return reinterpret_cast<unsigned long long*>(a) + 1; }

struct Stage {
    uint32_t buf[kSpillParts][kStage];
    uint32_t cnt[kSpillParts];
};

// Pass-1 admission filter ("doorkeeper"): entries are never evicted, so WHICH flows get the 1024 entries
// decides the hit rate. First come, first served hands about half of them to one-off flows of the cold tail
// (40 % of a Zipf(1.1) stream are tail records, each a distinct flow). With the filter a flow is admitted on its
// SECOND appearance in this workgroup's share: bit (hash) unset -> set it and spill the record. Hot flows come
// back within a few tiles; a tail flow rarely does before the cache is full. 32 Ki bits = the 4 KiB of LDS
// the cache and the spill staging leave free.
constexpr int kDoorBits = 32768;
struct Door { uint32_t bits[kDoorBits / 32]; };

NF_DEV uint32_t part_of(uint64_t h, const SpillView& q) { return (uint32_t)(h >> q.part_shift) & (q.n_parts - 1); }

NF_DEV void cache_init(Cache& L, int tid) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int e = tid; e < kEntries; e += kBlock) {
        L.k0[e] = z; L.t[e] = z; L.v[e] = z; L.w[e] = z; L.packets[e] = 0;
        L.q[e] = make_uint4(0, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    }
}

// phase A: find or claim the entry of hash h (the creator writes the key); -1 = window full (or not admitted yet)
// fill (pass 2's retry rounds): counts the entries created, so that the rounds can be sized by what the cache holds
template <bool DOOR>
NF_DEV int cache_claim(Cache& L, uint32_t* door, uint64_t h, const uint64_t w[5], uint32_t* fill = nullptr) {
    const uint64_t hk = h | 1ull;
    uint32_t e = (uint32_t)(h >> 40) & (kEntries - 1);
#pragma unroll 1
    for (int p = 0; p < kProbe; p++) {
        uint64_t cur = *lo64(&L.k0[e]);
        if (cur == 0) {
            if (DOOR) {
                const uint32_t b = (uint32_t)(h >> 14) & (kDoorBits - 1), m = 1u << (b & 31);
                // the atomic's own return value decides: of the lanes that meet a new flow in the same tile exactly one
                // is turned away (a hot flow would otherwise spill a whole burst into one partition's staging group)
                if (!(door[b >> 5] & m) && !(atomicOr(&door[b >> 5], m) & m)) return -1;
            }
            cur = atomicCAS(lo64(&L.k0[e]), 0ull, (unsigned long long)hk);
            if (cur == 0) {
                *hi64(&L.k0[e]) = w[0];
                L.k1[e] = mk4(w[1], w[2]);
                L.k2[e] = mk4(w[3], w[4]);
                if (fill) atomicAdd(fill, 1u);
                return (int)e;
            }
        }
        if (cur == hk) return (int)e;
        e = (e + 1) & (kEntries - 1);
    }
    return -1;
}

// phase B (after a barrier): full-key check, then AccumulateBase into the entry
NF_DEV int cache_fold(Cache& L, int ent, const Rec& r, const uint64_t w[5], uint32_t seq32) {
    const uint4 a = L.k0[ent], b = L.k1[ent], c = L.k2[ent];
    const bool same = ((u64hi(a) ^ w[0]) | (u64lo(b) ^ w[1]) | (u64hi(b) ^ w[2]) | (u64lo(c) ^ w[3]) | (u64hi(c) ^ w[4])) == 0;
    if (!same) return -1;         // two flows with one 64-bit hash: the later one is spilled / merged directly
    const uint4 tt = L.t[ent], qq = L.q[ent];
    if (r.bytes()) atomicAdd(lo64(&L.v[ent]), (unsigned long long)r.bytes());
    if (r.packets()) atomicAdd(&L.packets[ent], r.packets());
    if (r.flags() & ~qq.x) atomicOr(reinterpret_cast<uint32_t*>(&L.q[ent]), r.flags());
    if (r.end() > u64lo(tt)) atomicMax(lo64(&L.t[ent]), (unsigned long long)r.end());
    if (r.start() && ~r.start() > u64hi(tt)) atomicMax(hi64(&L.t[ent]), (unsigned long long)~r.start());
    const uint64_t s1 = (uint64_t)seq32 + 1;
    if (r.eth()) atomicMax(lo64(&L.w[ent]), (unsigned long long)((s1 << 16) | r.eth()));
    if (r.dscp()) atomicMax(hi64(&L.w[ent]), (unsigned long long)((s1 << 8) | r.dscp()));
    if (r.sampling()) atomicMax(hi64(&L.v[ent]), (unsigned long long)((s1 << 32) | r.sampling()));
    uint32_t* qw = reinterpret_cast<uint32_t*>(&L.q[ent]);
    if (qq.y > seq32) atomicMin(qw + 1, seq32);
    if (r.smac() && qq.z > seq32) atomicMin(qw + 2, seq32);
    if (r.dmac() && qq.w > seq32) atomicMin(qw + 3, seq32);
    return ent;
}

#ifdef NFAGG_DIAG
// ---- experiment (libnfagg_diag.so, ingest_variant 24; results RIGHT): a wave's duplicates combined BEFORE the LDS atomics ----------
// The round-2/3/4 reviews asked for this to be built and measured instead of estimated (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// is 56 % in pass 1: 64 lanes hit a hash-indexed cache, and ~5 lanes of every wave fold into the hottest flow). One group per
// wave and tile: the entries of lanes 0, 16, 32 and 48 are candidates, the one most lanes share wins; its lanes' contributions
// are reduced across the wave with DPP (row shifts + row broadcasts, identities elsewhere), ONE lane performs the entry's
// atomics with the totals, the others of the group skip theirs. Everything else takes cache_fold as before.
// Result: profiles/r05x_wave_combining.txt.
template <int CTRL, int ROW_MASK>
NF_DEV uint32_t comb_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true); }
template <int CTRL, int ROW_MASK>
NF_DEV uint64_t comb_dpp64(uint64_t v) { return (uint64_t)comb_dpp<CTRL, ROW_MASK>((uint32_t)v) | ((uint64_t)comb_dpp<CTRL, ROW_MASK>((uint32_t)(v >> 32)) << 32); }
#define NF_COMB_STEPS(OP, T, F)                                                                              \
    v = OP(v, F<0x111, 0xf>(v)); v = OP(v, F<0x112, 0xf>(v)); v = OP(v, F<0x114, 0xf>(v)); v = OP(v, F<0x118, 0xf>(v)); \
    v = OP(v, F<0x142, 0xa>(v)); v = OP(v, F<0x143, 0xc>(v));
NF_DEV uint32_t comb_add32(uint32_t a, uint32_t b) { return a + b; }
NF_DEV uint32_t comb_or32(uint32_t a, uint32_t b) { return a | b; }
NF_DEV uint32_t comb_max32(uint32_t a, uint32_t b) { return a > b ? a : b; }
NF_DEV uint64_t comb_add64(uint64_t a, uint64_t b) { return a + b; }
NF_DEV uint64_t comb_max64(uint64_t a, uint64_t b) { return a > b ? a : b; }
// the wave's total (lane 63 holds it after the steps; every lane must be active)
NF_DEV uint32_t wave_add32(uint32_t v) { NF_COMB_STEPS(comb_add32, uint32_t, comb_dpp) return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
NF_DEV uint32_t wave_or32(uint32_t v) { NF_COMB_STEPS(comb_or32, uint32_t, comb_dpp) return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
NF_DEV uint32_t wave_max32(uint32_t v) { NF_COMB_STEPS(comb_max32, uint32_t, comb_dpp) return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
NF_DEV uint64_t wave_add64(uint64_t v) {
    NF_COMB_STEPS(comb_add64, uint64_t, comb_dpp64)
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32);
}
NF_DEV uint64_t wave_max64(uint64_t v) {
    NF_COMB_STEPS(comb_max64, uint64_t, comb_dpp64)
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32);
}
#undef NF_COMB_STEPS

// phase B for a whole wave (every lane calls it: `valid` lanes with ent >= 0 take part). Returns the lane's entry, -1 for a
// key mismatch (as cache_fold).
NF_DEV int cache_fold_combined(Cache& L, bool valid, int ent, const Rec& r, const uint64_t w[5], uint32_t seq32) {
    const int lane = threadIdx.x & 63;
    bool active = valid && ent >= 0;
    if (active) {
        const uint4 a = L.k0[ent], b = L.k1[ent], c = L.k2[ent];
        active = ((u64hi(a) ^ w[0]) | (u64lo(b) ^ w[1]) | (u64hi(b) ^ w[2]) | (u64lo(c) ^ w[3]) | (u64hi(c) ^ w[4])) == 0;
        if (!active) ent = -1;
    }
    const int mine = active ? ent : -1;
    // the group: of the entries of four sample lanes the one most lanes share
    int best_e = -1;
    unsigned long long best_g = 0;
#pragma unroll
    for (int s = 0; s < 64; s += 16) {
        const int e = __builtin_amdgcn_readlane(mine, s);
        if (e >= 0) {
            const unsigned long long g = __ballot(mine == e);
            if (__popcll(g) > __popcll(best_g)) { best_g = g; best_e = e; }
        }
    }
    const bool grouped = __popcll(best_g) >= 2;                       // (wave-uniform)
    const bool member = grouped && mine == best_e;
    if (grouped) {
        const uint64_t s1 = (uint64_t)seq32 + 1;
        const uint64_t bytes = wave_add64(member ? r.bytes() : 0ull);
        const uint32_t packets = wave_add32(member ? r.packets() : 0u);
        const uint32_t flags = wave_or32(member ? r.flags() : 0u);
        const uint64_t end = wave_max64(member ? r.end() : 0ull);
        const uint64_t start_inv = wave_max64(member && r.start() ? ~r.start() : 0ull);
        const uint64_t eth_tag = wave_max64(member && r.eth() ? (s1 << 16) | r.eth() : 0ull);
        const uint64_t dscp_tag = wave_max64(member && r.dscp() ? (s1 << 8) | r.dscp() : 0ull);
        const uint64_t samp_tag = wave_max64(member && r.sampling() ? (s1 << 32) | r.sampling() : 0ull);
        const uint32_t first_inv = wave_max32(member ? ~seq32 : 0u);      // (~seq32 > 0: sequence numbers stop short of 2^32 - 16)
        const uint32_t smac_inv = wave_max32(member && r.smac() ? ~seq32 : 0u);
        const uint32_t dmac_inv = wave_max32(member && r.dmac() ? ~seq32 : 0u);
        if (lane == (int)__builtin_ctzll(best_g)) {                   // the group's first lane: the entry's atomics, once
            const int e = best_e;
            const uint4 tt = L.t[e], qq = L.q[e];
            if (bytes) atomicAdd(lo64(&L.v[e]), (unsigned long long)bytes);
            if (packets) atomicAdd(&L.packets[e], packets);
            if (flags & ~qq.x) atomicOr(reinterpret_cast<uint32_t*>(&L.q[e]), flags);
            if (end > u64lo(tt)) atomicMax(lo64(&L.t[e]), (unsigned long long)end);
            if (start_inv > u64hi(tt)) atomicMax(hi64(&L.t[e]), (unsigned long long)start_inv);
            if (eth_tag) atomicMax(lo64(&L.w[e]), (unsigned long long)eth_tag);
            if (dscp_tag) atomicMax(hi64(&L.w[e]), (unsigned long long)dscp_tag);
            if (samp_tag) atomicMax(hi64(&L.v[e]), (unsigned long long)samp_tag);
            uint32_t* qw = reinterpret_cast<uint32_t*>(&L.q[e]);
            if (qq.y > ~first_inv) atomicMin(qw + 1, ~first_inv);
            if (smac_inv && qq.z > ~smac_inv) atomicMin(qw + 2, ~smac_inv);
            if (dmac_inv && qq.w > ~dmac_inv) atomicMin(qw + 3, ~dmac_inv);
        }
    }
    if (active && !member) ent = cache_fold(L, ent, r, w, seq32);      // everybody else: as before (the key was checked twice: harmless)
    return ent;
}
#endif

// Overflow list: spills that found their partition queue (or staging group) full.
// Sized by the API for the worst case; pass 3 merges its records one by one.
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

// The fold loops need dwords 0..27 of a record only (key, counters, MACs, sampling, dscp):
// identity dwords are gathered at flush time. Seven 16-byte loads instead of nine.
NF_DEV void load_record_head(const void* base, uint64_t i, Rec& r) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + i * kRecordBytes);
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const uint4 v = p[k];
        r.d[4 * k] = v.x; r.d[4 * k + 1] = v.y; r.d[4 * k + 2] = v.z; r.d[4 * k + 3] = v.w;
    }
}

// ... five or six, when the queue entry says that the others do not matter (pass 2: need6 / need7, see queue_entry)
NF_DEV void load_record_head_5to7(const void* base, uint64_t i, Rec& r, bool need6, bool need7) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + i * kRecordBytes);
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint4 v = p[k];
        r.d[4 * k] = v.x; r.d[4 * k + 1] = v.y; r.d[4 * k + 2] = v.z; r.d[4 * k + 3] = v.w;
    }
    uint4 u = make_uint4(0, 0, 0, 0), v = make_uint4(0, 0, 0, 0);     // sampling = 0, dst_mac's last four bytes = 0, dscp = 0: what the flags said
    if (need6) u = p[5];
    if (need7) v = p[6];
    r.d[20] = u.x; r.d[21] = u.y; r.d[22] = u.z; r.d[23] = u.w;
    r.d[24] = v.x; r.d[25] = v.y; r.d[26] = v.z; r.d[27] = v.w;
}

NF_DEV uint4 rec_chunk(const void* recs, uint64_t i, int k) {
    return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + i * kRecordBytes)[k];
}

// Merge cache entry e into the table: one partial per entry. The entry's earliest non-zero MACs are
// gathered from the batch; the first record's identity dwords are left to k_finalize.
// EXCL (pass 2): this workgroup owns the flow — plain read-modify-write instead of atomics (merge_partial_exclusive),
// slots it claims are collected in new_list[] and registered in the live list once per workgroup.
// The dependent round trips of an entry are kept few (a flush is latency-bound: one entry per lane, 16 waves per CU): the MAC
// words are requested from the batch together with the home slot's line, before it is known whether they will be needed; the
// claim takes its first look at the home slot from that line; EXCL: a fresh slot's tag is published by the caller (returns the
// slot then, kNoSlot otherwise).
template <bool SKETCH, bool EXCL>
NF_DEV uint32_t cache_flush_entry(const TableView& t, const SketchView& sk, Cache& L, int e, const void* recs, uint32_t seq_base32,
                                  uint32_t* new_list, uint32_t* new_cnt, bool defer, uint64_t* fresh_hash = nullptr) {
    const uint4 ka = L.k0[e], qq = L.q[e];
    if (u64lo(ka) == 0 || qq.y == 0xffffffffu) return kNoSlot;            // free, or claimed but never folded into
    const uint4 kb = L.k1[e], kc = L.k2[e];
    const uint64_t w[5] = {u64hi(ka), u64lo(kb), u64hi(kb), u64lo(kc), u64hi(kc)};
    const uint64_t h = key_hash(w);
    const uint32_t fs = qq.y, ss = qq.z, ds = qq.w;
    uint4 s4 = make_uint4(0, 0, 0, 0), d4 = s4, d5 = s4;
    if (ss != 0xffffffffu) s4 = rec_chunk(recs, (uint64_t)(ss - seq_base32), 4);
    if (ds != 0xffffffffu) { d4 = rec_chunk(recs, (uint64_t)(ds - seq_base32), 4); d5 = rec_chunk(recs, (uint64_t)(ds - seq_base32), 5); }
    // the entry as the partial of its flow: for a slot that holds nothing older every part of it counts
    Partial p;
    const uint4 tt = L.t[e], vv = L.v[e], ww = L.w[e];
    p.bytes = u64lo(vv); p.end = u64lo(tt); p.start_inv = u64hi(tt);
    p.packets = L.packets[e]; p.flags = qq.x;
    p.eth_tag = u64lo(ww); p.dscp_tag = u64hi(ww); p.samp_tag = u64hi(vv);
    // The entry's earliest record may be the flow's first: only its sequence number goes into the slot (the tag of
    // id0); k_finalize copies that record's identity dwords from the batch after the last fold kernel of the call.
    p.first_inv = ~fs; p.ident0 = 0;
    p.smac_inv = 0; p.dmac_inv = 0; p.smac = 0; p.dmac = 0;
    if (ss != 0xffffffffu) { p.smac = (uint64_t)s4.z | ((uint64_t)(s4.w & 0xffffu) << 32); p.smac_inv = p.smac ? ~ss : 0u; }
    if (ds != 0xffffffffu) { p.dmac = (uint64_t)(d4.w >> 16) | ((uint64_t)d5.x << 16); p.dmac_inv = p.dmac ? ~ds : 0u; }
    Hints x;
    bool fresh = false;
    uint32_t idx = probe_home(t, w, h, x);
    if (idx == kNoSlot) {
        idx = (EXCL && defer) ? find_or_claim<true>(t, w, h, &fresh, &x.home_tag)
                              : find_or_claim(t, w, h, &fresh, &x.home_tag, EXCL ? nullptr : &p);   // !EXCL: a claimer writes p as the slot's first value
        if (idx == kNoSlot) return kNoSlot;
        if (!EXCL && fresh) {                                    // claimed and filled in one go: nothing left to merge
            if (SKETCH) sketch_add(sk, w, p.bytes);
            return kNoSlot;
        }
        if (fresh) { x.end = 0; x.start_inv = 0; x.id0 = 0; x.smac_lo = 0; x.dmac_lo = 0; x.flags = 0; new_list[atomicAdd(new_cnt, 1u)] = idx; }
        else load_hints(&t.hot[idx], x);
    }
    // against what the slot already holds (possibly stale = smaller, see Hints): drop what cannot win
    if ((uint32_t)(x.id0 >> 32) > ~fs) p.first_inv = 0;          // tagged(0, 0) = 0 never wins
    if (p.smac_inv && (uint32_t)(x.smac_lo >> 32) > p.smac_inv) { p.smac_inv = 0; p.smac = 0; }
    if (p.dmac_inv && (uint32_t)(x.dmac_lo >> 32) > p.dmac_inv) { p.dmac_inv = 0; p.dmac = 0; }
    if (EXCL) merge_partial_exclusive(t, idx, p, fresh, w, h, false);
    else merge_partial(t, idx, p, x);
    if (SKETCH) sketch_add(sk, w, p.bytes);
    if (EXCL && fresh) { *fresh_hash = h; return idx; }
    return kNoSlot;
}

// A queue entry = the record's index in the batch, and above it (batches of up to 2^29 - 1 records: always, in practice) three
// more bits of the key hash, the flow's SUB-PARTITION: pass 2 splits what a partition's cache could not take into eight by it.
// (Tried and dropped: giving up the first round after a few probe tiles when most of them miss, and sorting the rest of the
// queue by these bits without gathering a record — no gain even on a uniform 10 M-flow stream, 30.4 against 30.6 ms.)
constexpr int kSubBits = 3, kSubs = 1 << kSubBits;
constexpr uint32_t kIdxBits = 32 - kSubBits, kIdxMask = (1u << kIdxBits) - 1u;
// ... and below them, while the batch leaves room, bits that say which of the record's last 16-byte units pass 2 NEEDS. Pass 1 has the
// record in registers when it queues it: it knows. Of everything the fold reads
//   * only dscp lives in the SEVENTH unit (bytes 96..111: metrics byte 58), and a record whose dscp is zero — most traffic is best
//     effort — contributes nothing to the flow's "last non-zero dscp": bit 28, batches of up to 2^28 - 1 records;
//   * only sampling (metrics byte 52) and the last four bytes of dst_mac live in the SIXTH (bytes 80..95); the fold wants sampling when
//     it is non-zero and of dst_mac only WHETHER it is non-zero, which its first two bytes (unit five) answer unless they are zero
//     and the last four are not: bit 27, batches of up to 2^27 - 1 records.
// Without them a gathered record is bytes 0..79 / 0..95: a 144-byte record at offset 16 k of its 128-byte line (k = i mod 8) keeps
// 80 bytes inside ONE line for k <= 3, 96 for k <= 2, 112 only for k <= 1 — 1.5 / 1.625 / 1.75 lines per gathered record — and the
// gather is what bounds pass 2 (profiles/r05x_gather_flavours.txt: six units instead of seven -14 % time;
// profiles/r06x_pass2_fewer_units.txt: the 100 M-record call 5.22 -> 5.04 ms with the first bit alone).
constexpr uint32_t kNeed7Bit = 1u << (kIdxBits - 1);                  // bit 28
constexpr uint32_t kNeed6Bit = 1u << (kIdxBits - 2);                  // bit 27
constexpr uint32_t kIdxMaskFlagged = kNeed7Bit - 1u;                  // 28-bit indices: one flag
constexpr uint32_t kIdxMaskFlagged2 = kNeed6Bit - 1u;                 // 27-bit indices: both flags
// the index's share of a queue entry in a batch of n records (and with it: which flags the entries carry)
NF_DEV uint32_t entry_idx_mask(uint64_t n) {
    return n <= (uint64_t)kIdxMaskFlagged2 ? kIdxMaskFlagged2 : (n <= (uint64_t)kIdxMaskFlagged ? kIdxMaskFlagged : (n <= (uint64_t)kIdxMask ? kIdxMask : 0xffffffffu));
}
NF_DEV uint32_t sub_shift_of(const SpillView& q) { return q.part_shift >= (uint32_t)kSubBits ? q.part_shift - kSubBits : 0; }
NF_DEV uint32_t queue_entry(uint64_t i, uint64_t h, uint32_t sub_shift, uint32_t idx_mask, const Rec& r) {
    uint32_t e = (uint32_t)i;
    if (idx_mask <= kIdxMask) e |= ((uint32_t)(h >> sub_shift) & (uint32_t)(kSubs - 1)) << kIdxBits;
    if (idx_mask <= kIdxMaskFlagged && r.dscp()) e |= kNeed7Bit;
    // r.d[19] >> 16 = dst_mac's first two bytes, r.d[20] = its last four, r.d[23] = sampling
    if (idx_mask <= kIdxMaskFlagged2 && (r.d[23] != 0u || ((r.d[19] >> 16) == 0u && r.d[20] != 0u))) e |= kNeed6Bit;
    return e;
}

// ---- pass 1 ------------------------------------------------------------------------------------------------------
// 256 workgroups stream records[0..n), tiles of 1024 consecutive records, one per lane. Hot flows fold in the workgroup's
// persistent LDS cache; a record whose flow has no entry is spilled: its index goes to the queue of its flow's partition,
// staged four at a time in LDS so that a spill costs one 16-byte store and a quarter of an atomic.
// ABL (libnfagg_diag.so only, ingest_variant 20..27: timing experiments, results are WRONG): bit 0 = spills are counted but not
// queued, bit 1 = no fold into the cache entry, bit 2 = no cache claim (every record counts as a miss). Bit 3 (variant 24, results
// RIGHT): a wave's duplicates are combined before the LDS atomics (cache_fold_combined).
template <bool SKETCH, bool TIMING, bool DOOR, int ABL = 0, bool DEEP = false>
__global__ __launch_bounds__(kBlock) void k_pass1(TableView t, SketchView sk, SpillView q, const void* __restrict__ recs,
                                                  uint64_t n, uint64_t seq_base) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    Cache& L = *reinterpret_cast<Cache*>(lds_raw);
    Stage& S = *reinterpret_cast<Stage*>(lds_raw + sizeof(Cache));
    uint32_t* door = reinterpret_cast<Door*>(lds_raw + sizeof(Cache) + sizeof(Stage))->bits;   // with DOOR only
    const int tid = threadIdx.x;
    const uint32_t seq_base32 = (uint32_t)seq_base;
    cache_init(L, tid);
    for (int p = tid; p < kSpillParts; p += kBlock) S.cnt[p] = 0;
    if (DOOR) for (int p = tid; p < kDoorBits / 32; p += kBlock) door[p] = 0;
    __syncthreads();

    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, tp = 0;          // TIMING: load+hash, A, barrier, B, barrier, C, flush
#define NF_TICK(k) do { if (TIMING) { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[k] += tn_ - tp; tp = tn_; } } while (0)
    if (TIMING) tp = __builtin_readcyclecounter();
    const uint64_t n_tiles = (n + kBlock - 1) / kBlock;
    // Tiles go round the workgroups: every workgroup's cache sees the whole stream's hot flows.
    const uint64_t tile_first = (uint64_t)blockIdx.x, tile_end = n_tiles, tile_step = (uint64_t)gridDim.x;
    unsigned long long skipped = 0, spilled = 0;
    const bool tag_on = n <= (uint64_t)kIdxMask;                       // the index leaves room for the sub-partition bits
    const uint32_t e_mask = entry_idx_mask(n);                         // ... and for the "units needed" bits
    (void)tag_on;                                                      // (the diag build's queue-store experiment reads it)
    const uint32_t sub_shift = sub_shift_of(q);
    // Drain state. The lane whose append FILLS a staging group (position kStage - 1) drains it one tile later — no lane polls the
    // 2048 group counters (round 2: two LDS reads per lane and tile) — and stores the drained group another tile later, when the
    // queue reservation (a returning atomic) has long arrived: no HBM round trip inside a tile. A lane fills at most two groups per
    // tile: one with a carried-over spill, one with its own.
    constexpr int kMine = kSpillParts / kBlock;
    constexpr uint32_t kNoPart = 0xffffffffu;
    uint4 pend_v[2];
    uint32_t pend_at[2], pend_p[2], fill_p[2];
    bool pend[2];
#pragma unroll
    for (int k = 0; k < 2; k++) { pend[k] = false; pend_at[k] = 0; pend_p[k] = 0; fill_p[k] = kNoPart; pend_v[k] = make_uint4(0, 0, 0, 0); }
    uint32_t carry = 0xffffffffu, carry_p = 0;                         // a spill that found its group full: retried next tile
#ifdef NFAGG_DIAG
    // ABL bit 4 (variant 26; results WRONG — pass 2 does not know about it — a timing experiment for PASS 1 only): a quarter of the
    // partitions, staging groups of SIXTEEN entries, every queue write one whole 64-byte sector (four 16-byte stores to
    // consecutive addresses) instead of a 16-byte piece of one. What the round-4 review's item 2(a) could give pass 1 at best.
    uint4 pw_v[2][4];
    uint32_t pw_at[2] = {0, 0}, pw_p[2] = {0, 0}, fw_p[2] = {kNoPart, kNoPart};
    bool pw[2] = {false, false};
    uint32_t* const SW = &S.buf[0][0];                                 // [512][16]
#endif
    // Software pipeline: the records of tile k+1 are requested before tile k is folded, so no HBM latency is exposed inside
    // a tile. Loads are unconditional on a clamped index; `valid` only gates the fold.
    // DEEP (experiment, round 5: libnfagg_diag.so ingest_variant 28): the records of tile k+2 are requested before tile k is folded —
    // three record buffers in fixed roles, the loop unrolled by three so that no register copy waits for the youngest load; the
    // seventh 16-byte unit shrinks to the one dword of it the fold reads (dscp) to stay inside 128 registers.
    auto fold_tile = [&](Rec& r, bool valid, const uint64_t i) __attribute__((always_inline)) {
        uint64_t w[5];
        uint64_t h = 0;
        if (valid) {
            r.canonicalize();
            r.key_words(w);
            h = key_hash(w);
            if (t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id) { valid = false; skipped++; }
        }
        const uint32_t seq32 = seq_base32 + (uint32_t)i;
        if (TIMING) { asm volatile("" :: "v"(h)); NF_TICK(0); }
        int ent = (valid && !(ABL & 4)) ? cache_claim<DOOR>(L, door, h, w) : -1;
        NF_TICK(1);
        __syncthreads();
        NF_TICK(2);
#ifdef NFAGG_DIAG
        if (ABL & 8) ent = cache_fold_combined(L, valid, ent, r, w, seq32);   // experiment: a wave's duplicates combined first (variant 24)
        else
#endif
        if (valid && ent >= 0 && !(ABL & 2)) ent = cache_fold(L, ent, r, w, seq32);
#ifdef NFAGG_DIAG
        if (ABL & 16) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (pw[k]) {
                    if (pw_at[k] + 16u <= 4u * q.qcap) {
                        uint4* dst = reinterpret_cast<uint4*>(q.queue + (uint64_t)pw_p[k] * 4u * q.qcap + pw_at[k]);
#pragma unroll
                        for (int j = 0; j < 4; j++) dst[j] = pw_v[k][j];
                    }
                    pw[k] = false;
                }
                if (fw_p[k] != kNoPart) {
                    const uint32_t p = fw_p[k];
                    const uint4* src = reinterpret_cast<const uint4*>(SW + p * 16u);
#pragma unroll
                    for (int j = 0; j < 4; j++) pw_v[k][j] = src[j];
                    S.cnt[p] = 0;
                    pw_at[k] = aadd(&q.qtail[p * 4u], 16u);
                    pw_p[k] = p; pw[k] = true; fw_p[k] = kNoPart;
                }
            }
            __syncthreads();
            if (carry != 0xffffffffu) {
                const uint32_t at = atomicAdd(&S.cnt[carry_p], 1u);
                if (at < 16u) { SW[carry_p * 16u + at] = carry; if (at == 15u) fw_p[0] = carry_p; }
                carry = 0xffffffffu;
            }
            if (valid && ent < 0) {
                spilled++;
                const uint32_t p = part_of(h, q) >> 2;
                const uint32_t at = atomicAdd(&S.cnt[p], 1u);
                const uint32_t qi = (uint32_t)i | (tag_on ? ((uint32_t)(h >> sub_shift) & (uint32_t)(kSubs - 1)) << kIdxBits : 0u);
                if (at < 16u) { SW[p * 16u + at] = qi; if (at == 15u) fw_p[1] = p; }
                else { carry = qi; carry_p = p; }
            }
            return;
        }
#endif
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (pend[k]) {
                if (pend_at[k] + kStage <= q.qcap) *reinterpret_cast<uint4*>(q.queue + (uint64_t)pend_p[k] * q.qcap + pend_at[k]) = pend_v[k];
                else overflow_push(q, pend_v[k]);                     // partition queue full (adversarial skew)
                pend[k] = false;
            }
            if (fill_p[k] != kNoPart) {                               // filled by this lane in the previous tile: every append is in LDS (barriers since)
                const uint32_t p = fill_p[k];
                pend_v[k] = *reinterpret_cast<const uint4*>(S.buf[p]);
                S.cnt[p] = 0;                                         // appends resume after the next barrier
                pend_at[k] = aadd(&q.qtail[p], (uint32_t)kStage);
                pend_p[k] = p; pend[k] = true; fill_p[k] = kNoPart;
            }
        }
        NF_TICK(3);
        __syncthreads();
        NF_TICK(4);
        if (carry != 0xffffffffu) {
            const uint32_t at = atomicAdd(&S.cnt[carry_p], 1u);
            if (at < (uint32_t)kStage) { S.buf[carry_p][at] = carry; if (at == (uint32_t)kStage - 1) fill_p[0] = carry_p; }
            else overflow_push_one(q, carry);   // full twice in a row: very rare
            carry = 0xffffffffu;
        }
        if (valid && ent < 0) {
            spilled++;
            if (!(ABL & 1)) {
                const uint32_t p = part_of(h, q);
                const uint32_t at = atomicAdd(&S.cnt[p], 1u);
                const uint32_t qi = queue_entry(i, h, sub_shift, e_mask, r);
                if (at < (uint32_t)kStage) { S.buf[p][at] = qi; if (at == (uint32_t)kStage - 1) fill_p[1] = p; }
                else { carry = qi; carry_p = p; }
            }
        }
        if (TIMING) NF_TICK(5);
        // next tile: phase A touches only h64/key of NEW entries, the barrier after it orders phase B/C as before;
        // staging appends of this tile are drained after the next tile's first barrier
    };
    if (!DEEP) {
        bool valid; uint64_t i; Rec r;
        {
            const uint64_t pos = tile_first * kBlock + tid;
            valid = pos < n; i = valid ? pos : 0;
            load_record_head(recs, i, r);
        }
        for (uint64_t tile = tile_first; tile < tile_end; tile += tile_step) {
            bool valid_n; uint64_t i_n; Rec r_n;
            {
                const uint64_t pos = (tile + tile_step) * kBlock + tid;
                valid_n = pos < n; i_n = valid_n ? pos : 0;
                load_record_head(recs, i_n, r_n);
            }
            fold_tile(r, valid, i);
            valid = valid_n; i = i_n;
#pragma unroll
            for (int k = 0; k < 28; k++) r.d[k] = r_n.d[k];
        }
    } else {
        Rec ra, rb, rc; bool va, vb, vc; uint64_t ia, ib, ic;
        auto request = [&](uint64_t tile_idx, Rec& rr, bool& vv, uint64_t& ii) __attribute__((always_inline)) {
            const uint64_t pos = tile_idx * kBlock + tid;
            vv = pos < n; ii = vv ? pos : 0;
            const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + ii * kRecordBytes);
#pragma unroll
            for (int k = 0; k < 6; k++) { const uint4 v = p[k]; rr.d[4 * k] = v.x; rr.d[4 * k + 1] = v.y; rr.d[4 * k + 2] = v.z; rr.d[4 * k + 3] = v.w; }
            rr.d[24] = reinterpret_cast<const uint32_t*>(p)[24];      // dscp; dwords 25..27 are not read by the fold
        };
        request(tile_first, ra, va, ia);
        request(tile_first + tile_step, rb, vb, ib);
        uint64_t tile = tile_first;
        for (;;) {
            if (tile >= tile_end) break;
            request(tile + 2 * tile_step, rc, vc, ic); fold_tile(ra, va, ia); tile += tile_step;
            if (tile >= tile_end) break;
            request(tile + 2 * tile_step, ra, va, ia); fold_tile(rb, vb, ib); tile += tile_step;
            if (tile >= tile_end) break;
            request(tile + 2 * tile_step, rb, vb, ib); fold_tile(rc, vc, ic); tile += tile_step;
        }
    }
    __syncthreads();
    if (carry != 0xffffffffu) {
        const uint32_t at = atomicAdd(&S.cnt[carry_p], 1u);
        if (at < (uint32_t)kStage) S.buf[carry_p][at] = carry;
        else overflow_push_one(q, carry);
    }
    __syncthreads();
    // pending groups, then whatever is staged (padded with invalid indices): groups filled in the last tile are still in LDS
    // (their fill marks are dropped here), every lane looks after its two partitions
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (pend[k]) {
            if (pend_at[k] + kStage <= q.qcap) *reinterpret_cast<uint4*>(q.queue + (uint64_t)pend_p[k] * q.qcap + pend_at[k]) = pend_v[k];
            else overflow_push(q, pend_v[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < kMine; k++) {
        const int p = tid + k * kBlock;
        uint32_t c = S.cnt[p];
        if (c > (uint32_t)kStage) c = kStage;
        if (c) {
            const uint32_t at = aadd(&q.qtail[p], (uint32_t)kStage);
            uint4 v = *reinterpret_cast<const uint4*>(S.buf[p]);
            if (c < 2) v.y = 0xffffffffu;
            if (c < 3) v.z = 0xffffffffu;
            if (c < 4) v.w = 0xffffffffu;
            if (at + kStage <= q.qcap) *reinterpret_cast<uint4*>(q.queue + (uint64_t)p * q.qcap + at) = v;
            else overflow_push(q, v);
        }
    }
    for (int e = tid; e < kEntries; e += kBlock) cache_flush_entry<SKETCH, false>(t, sk, L, e, recs, seq_base32, nullptr, nullptr, false);
    if (TIMING) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NF_TICK(6);
        if ((tid & 63) == 0) for (int k = 0; k < 7; k++) aadd(&t.ctr->phase[k], ph[k]);
    }
#undef NF_TICK
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
    if (spilled) aadd(&t.ctr->n_bypassed, spilled);
}

// ---- pass 1 WITHOUT its barriers (round 6; ingest_variant 17) -------------------------------------------------------------------
// k_pass1 above takes its 16 waves through two workgroup barriers per tile: one so that the creator of a cache entry may write the
// entry's key with plain stores before anybody compares it, one for the spill staging (a group is drained a tile after it was
// filled). Its phase timing says what that costs: 29 % of the wave-cycles at the barriers, waves that hold their loads already
// waiting for the slowest one (profiles/r03_pass1_ablation.txt; two tiles of prefetch did not help: r05x_pass1_two_tiles_ahead.txt).
// Neither barrier is needed for EXACTNESS if the two protocols say when their data is complete:
//   * cache entries are PUBLISHED. The creator claims the entry with a marker no key hash can equal (hashes are stored | 1: odd;
//     the marker is 2), writes the key, then stores the hash with release semantics. Whoever reads the hash (acquire) reads a
//     complete key; whoever meets the marker treats the record as a miss (it is spilled: pass 2 folds it). Entries are never
//     evicted and keys never change, so a fold only ever adds a record to an entry whose full key it has compared. A flow may end
//     up with two entries (two lanes racing past each other's markers): two partials, merged by the flush like any others.
//   * staging groups are drained by their LAST WRITER. An append reserves a position (low half of the group's counter), writes its
//     item, then counts itself done (high half); the lane whose "done" is the fourth reads the group — every writer's store precedes
//     its "done" — and resets the counter. Appends that meet a full group (position >= 4) are carried to the lane's next tile,
//     as before; nobody waits.
// The waves then run free: a wave that has its records folds them while another still waits for its loads.
constexpr uint64_t kEntryBusy = 2ull;              // never a stored hash (those are odd), never 0 (free)

template <bool DOOR>
NF_DEV int cache_claim_published(Cache& L, uint32_t* door, uint64_t h, const uint64_t w[5]) {
    const uint64_t hk = h | 1ull;
    uint32_t e = (uint32_t)(h >> 40) & (kEntries - 1);
#pragma unroll 1
    for (int p = 0; p < kProbe; p++) {
        uint64_t cur = __hip_atomic_load(lo64(&L.k0[e]), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == 0) {
            if (DOOR) {
                const uint32_t b = (uint32_t)(h >> 14) & (kDoorBits - 1), m = 1u << (b & 31);
                if (!(door[b >> 5] & m) && !(atomicOr(&door[b >> 5], m) & m)) return -1;
            }
            cur = atomicCAS(lo64(&L.k0[e]), 0ull, (unsigned long long)kEntryBusy);
            if (cur == 0) {
                *hi64(&L.k0[e]) = w[0];
                L.k1[e] = mk4(w[1], w[2]);
                L.k2[e] = mk4(w[3], w[4]);
                __hip_atomic_store(lo64(&L.k0[e]), (unsigned long long)hk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                return (int)e;
            }
        }
        if (cur == hk) return (int)e;
        if (cur == kEntryBusy) return -1;          // somebody is writing a key here (maybe this flow's): a miss, not a wait
        e = (e + 1) & (kEntries - 1);
    }
    return -1;
}

template <bool SKETCH, bool DOOR>
__global__ __launch_bounds__(kBlock) void k_pass1_free(TableView t, SketchView sk, SpillView q, const void* __restrict__ recs,
                                                       uint64_t n, uint64_t seq_base) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    Cache& L = *reinterpret_cast<Cache*>(lds_raw);
    Stage& S = *reinterpret_cast<Stage*>(lds_raw + sizeof(Cache));
    uint32_t* door = reinterpret_cast<Door*>(lds_raw + sizeof(Cache) + sizeof(Stage))->bits;   // with DOOR only
    const int tid = threadIdx.x;
    const uint32_t seq_base32 = (uint32_t)seq_base;
    cache_init(L, tid);
    for (int p = tid; p < kSpillParts; p += kBlock) S.cnt[p] = 0;
    if (DOOR) for (int p = tid; p < kDoorBits / 32; p += kBlock) door[p] = 0;
    __syncthreads();
    const uint64_t n_tiles = (n + kBlock - 1) / kBlock;
    const uint64_t tile_first = (uint64_t)blockIdx.x, tile_step = (uint64_t)gridDim.x;
    unsigned long long skipped = 0, spilled = 0;
    const uint32_t e_mask = entry_idx_mask(n);
    const uint32_t sub_shift = sub_shift_of(q);
    constexpr int kMine = kSpillParts / kBlock;
    // a group this lane drained: its queue position is reserved when it is drained, the store follows a tile later (the
    // reservation is a returning atomic: no HBM round trip inside a tile). A lane appends at most twice per tile.
    uint4 pend_v[2];
    uint32_t pend_at[2], pend_p[2];
    bool pend[2] = {false, false};
#pragma unroll
    for (int k = 0; k < 2; k++) { pend_at[k] = 0; pend_p[k] = 0; pend_v[k] = make_uint4(0, 0, 0, 0); }
    uint32_t carry = 0xffffffffu, carry_p = 0;
    int carried_for = 0;                                           // tiles the carried item has met a full group
    auto append = [&](uint32_t p, uint32_t qi, int slot) __attribute__((always_inline)) -> bool {
        const uint32_t at = atomicAdd(&S.cnt[p], 1u) & 0xffffu;
        if (at >= (uint32_t)kStage) return false;                  // full (its last writer has not reset it yet)
        S.buf[p][at] = qi;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the item before the "done"
        const uint32_t old = atomicAdd(&S.cnt[p], 0x10000u);
        if ((old >> 16) == (uint32_t)kStage - 1) {                 // the last of the four writers: the group is complete, and this lane's
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            pend_v[slot] = *reinterpret_cast<const uint4*>(S.buf[p]);
            __hip_atomic_store(&S.cnt[p], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // appends resume
            pend_at[slot] = aadd(&q.qtail[p], (uint32_t)kStage);
            pend_p[slot] = p; pend[slot] = true;
        }
        return true;
    };
    auto flush_pends = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (pend[k]) {
                if (pend_at[k] + kStage <= q.qcap) *reinterpret_cast<uint4*>(q.queue + (uint64_t)pend_p[k] * q.qcap + pend_at[k]) = pend_v[k];
                else overflow_push(q, pend_v[k]);
                pend[k] = false;
            }
        }
    };
    bool valid; uint64_t i; Rec r;
    {
        const uint64_t pos = tile_first * kBlock + tid;
        valid = pos < n; i = valid ? pos : 0;
        load_record_head(recs, i, r);
    }
    for (uint64_t tile = tile_first; tile < n_tiles; tile += tile_step) {
        bool valid_n; uint64_t i_n; Rec r_n;
        {
            const uint64_t pos = (tile + tile_step) * kBlock + tid;
            valid_n = pos < n; i_n = valid_n ? pos : 0;
            load_record_head(recs, i_n, r_n);
        }
        // what this lane drained in its previous tile goes out now (its reservation has long arrived)
        flush_pends();
        uint64_t w[5];
        uint64_t h = 0;
        bool v = valid;
        if (v) {
            r.canonicalize();
            r.key_words(w);
            h = key_hash(w);
            if (t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id) { v = false; skipped++; }
        }
        const uint32_t seq32 = seq_base32 + (uint32_t)i;
        int ent = v ? cache_claim_published<DOOR>(L, door, h, w) : -1;
        if (v && ent >= 0) ent = cache_fold(L, ent, r, w, seq32);
        if (carry != 0xffffffffu) {
            if (append(carry_p, carry, 0)) { carry = 0xffffffffu; carried_for = 0; }
            else if (++carried_for >= 2) { overflow_push_one(q, carry); carry = 0xffffffffu; carried_for = 0; }   // full twice in a row: very rare
        }
        if (v && ent < 0) {
            spilled++;
            const uint32_t p = part_of(h, q);
            const uint32_t qi = queue_entry(i, h, sub_shift, e_mask, r);
            if (!append(p, qi, 1)) {
                if (carry == 0xffffffffu) { carry = qi; carry_p = p; carried_for = 0; }
                else overflow_push_one(q, qi);                     // (a carried item is still waiting: at most one per lane)
            }
        }
        valid = valid_n; i = i_n;
#pragma unroll
        for (int k = 0; k < 28; k++) r.d[k] = r_n.d[k];
    }
    flush_pends();                                                 // (the carried item below may drain another group into slot 0)
    if (carry != 0xffffffffu) { if (!append(carry_p, carry, 0)) overflow_push_one(q, carry); }
    __syncthreads();                                               // every append of the workgroup is done: every complete group was drained by its last writer
    flush_pends();
    // whatever is staged in groups that never filled (padded with invalid indices): every lane looks after its two partitions
#pragma unroll
    for (int k = 0; k < kMine; k++) {
        const int p = tid + k * kBlock;
        uint32_t c = S.cnt[p] & 0xffffu;
        if (c > (uint32_t)kStage) c = kStage;
        if (c) {
            const uint32_t at = aadd(&q.qtail[p], (uint32_t)kStage);
            uint4 vq = *reinterpret_cast<const uint4*>(S.buf[p]);
            if (c < 2) vq.y = 0xffffffffu;
            if (c < 3) vq.z = 0xffffffffu;
            if (c < 4) vq.w = 0xffffffffu;
            if (at + kStage <= q.qcap) *reinterpret_cast<uint4*>(q.queue + (uint64_t)p * q.qcap + at) = vq;
            else overflow_push(q, vq);
        }
    }
    for (int e = tid; e < kEntries; e += kBlock) cache_flush_entry<SKETCH, false>(t, sk, L, e, recs, seq_base32, nullptr, nullptr, false);
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
    if (spilled) aadd(&t.ctr->n_bypassed, spilled);
}

// ---- pass 2 ------------------------------------------------------------------------------------------------------
// Workgroup b folds the records whose indices pass 1 queued for partition b, in its LDS cache, and owns the partition's
// flows while it runs (plain read-modify-write flush, nfagg_device.h merge_partial_exclusive).
//
// More flows in a partition than the cache has entries (beyond ~2 M flows per GPU and epoch): a record whose flow finds the
// probe window full is not merged into HBM on its own (4-5 atomics per record: 14 ms of a 19 ms call at 10 M flows) but RETRIED:
// its index, tagged with three more bits of the key hash (the sub-partition), goes back into the front of the workgroup's own
// queue region — in place: the write position never passes the read position. After the queue: flush, then the retry list is
// sorted by sub-partition into the free tail of the region (counting sort, counts kept during the first round) and every
// sub-partition gets a round of its own with a fresh cache: an eighth of the missed flows each. What misses even then is
// merged on its own, as before. A stream whose partitions fit (configs[1]: ~490 flows each) never enters a second round.
struct Pass2Lds {                 // after the Cache
    uint32_t new_list[kEntries];  // slots claimed by the current flush
    uint32_t new_cnt[4];          // [0] count, [1..2] base of the reserved live-list range, [3] deferred claims allowed
    uint32_t fill;                // retry rounds: cache entries in use since the last cache_init
    uint32_t retry_cnt;           // indices written back for a second round
    uint32_t sub_cnt[kSubs];      // ... per sub-partition
    uint32_t sub_off[kSubs + 1];  // counting sort: start of every sub-partition's segment
    uint32_t sub_fill[kSubs];
};

// Fold the `count` queue entries at `queue` (0xffffffff = padding). RETRY: misses go back to retry_to[] (tagged with their
// sub-partition) instead of being merged on their own; COHERENT: the entries were written by this workgroup (read past L1).
template <bool SKETCH, bool TIMING, bool RETRY, bool COHERENT>
NF_DEV void pass2_round(const TableView& t, const SketchView& sk, Cache& L, Pass2Lds& P, const uint32_t* queue, uint32_t count,
                        uint32_t* retry_to, const void* recs, uint64_t seq_base, uint32_t idx_mask, uint32_t need7_bit, unsigned long long& direct,
                        unsigned long long* ph, unsigned long long& tp) {
    // need7_bit: kNeed7Bit when pass 1 flagged its queue entries (batches of < 2^28 records), 0 when every record's seventh unit is read;
    // the sixth unit's flag rides along in batches of < 2^27 records (idx_mask says so)
    const uint32_t need6_bit = idx_mask == kIdxMaskFlagged2 ? kNeed6Bit : 0u;
    auto need7 = [&](uint32_t qi) -> bool { return need7_bit == 0u || (qi & need7_bit) != 0u; };
    auto need6 = [&](uint32_t qi) -> bool { return need6_bit == 0u || (qi & need6_bit) != 0u; };
#define NF_TICK2(k) do { if (TIMING) { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[k] += tn_ - tp; tp = tn_; } } while (0)
    const int tid = threadIdx.x;
    const uint32_t seq_base32 = (uint32_t)seq_base;
    auto qload = [&](uint32_t pos) -> uint32_t { return COHERENT ? ald(&queue[pos]) : queue[pos]; };
    const uint32_t n_tiles = (count + kBlock - 1) / kBlock;
    // Software pipeline: queue entries two tiles ahead, records one tile ahead; loads are unconditional on a clamped index.
    bool valid; uint32_t i; Rec r;
    uint32_t qi_next = 0xffffffffu, qi_cur = 0xffffffffu;
    {
        const uint32_t pos = (uint32_t)tid;
        const uint32_t qi = pos < count ? qload(pos) : 0xffffffffu;
        valid = qi != 0xffffffffu; i = valid ? (qi & idx_mask) : 0;
        qi_cur = qi;
        if (pos + kBlock < count) qi_next = qload(pos + kBlock);
        load_record_head_5to7(recs, i, r, need6(qi), need7(qi));
    }
    for (uint32_t tile = 0; tile < n_tiles; tile++) {
        bool valid_n; uint32_t i_n; Rec r_n; uint32_t qi_nn = 0xffffffffu;
        {
            valid_n = qi_next != 0xffffffffu; i_n = valid_n ? (qi_next & idx_mask) : 0;
            const uint64_t p2 = (uint64_t)(tile + 2) * kBlock + tid;
            if (p2 < count) qi_nn = qload((uint32_t)p2);
            load_record_head_5to7(recs, i_n, r_n, need6(qi_next), need7(qi_next));
        }
        uint64_t w[5];
        uint64_t h = 0;
        if (valid) { r.canonicalize(); r.key_words(w); h = key_hash(w); }
        const uint32_t seq32 = seq_base32 + i;
        if (TIMING) { asm volatile("" :: "v"(h)); NF_TICK2(0); }
        int ent = valid ? cache_claim<false>(L, nullptr, h, w, COHERENT ? &P.fill : nullptr) : -1;   // (COHERENT = the retry rounds)
        NF_TICK2(1);
        __syncthreads();
        NF_TICK2(2);
        if (valid && ent >= 0) ent = cache_fold(L, ent, r, w, seq32);
        NF_TICK2(3);
        if (valid && ent < 0) {
            if (RETRY) {
                // writes land below (tile + 1) * kBlock; the entries of the next two tiles are in registers already
                const uint32_t at = atomicAdd(&P.retry_cnt, 1u);
                atomicAdd(&P.sub_cnt[qi_cur >> kIdxBits], 1u);
                retry_to[at] = qi_cur;                            // index + sub-partition bits, as pass 1 queued it
            } else {
                // no cache entry even now: merge the record itself (all 144 bytes needed)
                direct++;
                Rec full;
                load_record(recs, i, full);
                full.canonicalize();
                Partial p;
                partial_from_record(full, seq_base + i, p);
                upsert_partial(t, w, h, p);
                if (SKETCH) sketch_add(sk, w, full.bytes());
            }
        }
        NF_TICK2(5);
        valid = valid_n; i = i_n; qi_cur = qi_next; qi_next = qi_nn;
#pragma unroll
        for (int k = 0; k < 28; k++) r.d[k] = r_n.d[k];
    }
    __syncthreads();
#undef NF_TICK2
}

// Before a round (lane 0; the round's barriers publish it): may this flush collect its claims and count them once? Only while
// the claims that workgroups may hold uncounted cannot carry the table past its claim limit — 256 resident workgroups x 1024,
// and as much again for what is registered between this look at n_live and the flush it is used in (a round earlier).
constexpr uint32_t kPackRecords = 768;
NF_DEV void pass2_defer_ok(const TableView& t, Pass2Lds& P) {
    P.new_cnt[0] = 0;
    P.new_cnt[3] = (t.defer_claims && ald(&t.ctr->n_live) + 512ull * kEntries <= t.claim_limit) ? 1u : 0u;
}

// Flush the cache into the table (exclusive) and register the slots it claimed: one range of the live list per flush.
template <bool SKETCH>
NF_DEV void pass2_flush(const TableView& t, const SketchView& sk, Cache& L, Pass2Lds& P, const void* recs, uint32_t seq_base32) {
    static_assert(kEntries == kBlock, "one cache entry per lane");
    const int tid = threadIdx.x;
    // P.new_cnt[0] = 0 and P.new_cnt[3] (deferred claims allowed, pass2_defer_ok) were written before the round: barriers since.
    const bool defer = P.new_cnt[3] != 0;
    uint64_t fresh_hash = 0;
    const uint32_t fresh_idx = cache_flush_entry<SKETCH, true>(t, sk, L, tid, recs, seq_base32, P.new_list, P.new_cnt, defer, &fresh_hash);
    // the slots this workgroup claimed: one range of the live list, reserved with one atomic. Positions at or beyond
    // claim_limit are given back (find_or_claim's rule, applied to the range): slot emptied, n_live restored, `aborted`.
    __syncthreads();
    const uint32_t cnt = P.new_cnt[0];
    unsigned long long base0 = 0;
    if (cnt && tid == 0) base0 = aadd(&t.ctr->n_live, (unsigned long long)cnt);     // in flight while the stores drain
    // key and value stores of the fresh slots are acknowledged: their tags may say `ready` (one wait per lane, all lanes at once)
    drain_stores();
    if (fresh_idx != kNoSlot) ast(&t.hot[fresh_idx].tag, tag_ready(t, fresh_hash));
    if (cnt) {
        if (tid == 0) {
            const unsigned long long base = base0;
            P.new_cnt[1] = (uint32_t)base; P.new_cnt[2] = (uint32_t)(base >> 32);
            if (base + cnt > t.claim_limit) {
                const unsigned long long keep = base < t.claim_limit ? t.claim_limit - base : 0ull;
                aadd(&t.ctr->n_live, ~(unsigned long long)(cnt - keep) + 1ull);
                atomicExch(&t.ctr->aborted, 1u);
            }
        }
        __syncthreads();
        const unsigned long long base = (unsigned long long)P.new_cnt[1] | ((unsigned long long)P.new_cnt[2] << 32);
        for (uint32_t k = tid; k < cnt; k += kBlock) {
            if (base + k < t.claim_limit) t.live_list[base + k] = P.new_list[k];
            else ast(&t.hot[P.new_list[k]].tag, (uint64_t)0);
        }
    }
    __syncthreads();
}

template <bool SKETCH, bool TIMING>
__global__ __launch_bounds__(kBlock) void k_pass2(TableView t, SketchView sk, SpillView q, const void* __restrict__ recs,
                                                  uint64_t n, uint64_t seq_base) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    Cache& L = *reinterpret_cast<Cache*>(lds_raw);
    Pass2Lds& P = *reinterpret_cast<Pass2Lds*>(lds_raw + sizeof(Cache));
    const int tid = threadIdx.x;
    const uint32_t seq_base32 = (uint32_t)seq_base;
    const uint32_t tail = q.qtail[blockIdx.x];                        // written by pass 1 (previous kernel)
    const uint32_t count = tail < q.qcap ? tail : q.qcap;
    if (count == 0) return;
    uint32_t* my_queue = q.queue + (uint64_t)blockIdx.x * q.qcap;
    cache_init(L, tid);
    if (tid == 0) { P.retry_cnt = 0; pass2_defer_ok(t, P); }
    if (tid < kSubs) { P.sub_cnt[tid] = 0; P.sub_fill[tid] = 0; }
    __syncthreads();
    if (tid == 0) q.qtail[blockIdx.x] = 0;                            // every lane has read it: ready for the next batch
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, tp = 0, direct = 0;
    if (TIMING) tp = __builtin_readcyclecounter();
    // sub-partition = the three hash bits below the partition's (slot-index bits, nfagg_create); retries need 29-bit indices
    // and room for the sorted list behind the queue
    const bool tagged = n <= (uint64_t)kIdxMask;                      // pass 1 put the sub-partition bits above the index
    const uint32_t idx_mask = entry_idx_mask(n);                      // ... and the "units needed" bits below them
    const uint32_t need7_bit = idx_mask <= kIdxMaskFlagged ? kNeed7Bit : 0u;
    const uint32_t sorted_at = (count + 3u) & ~3u;
    const bool retry_ok = tagged && (uint64_t)sorted_at + count <= q.qcap;
    if (retry_ok) pass2_round<SKETCH, TIMING, true, false>(t, sk, L, P, my_queue, count, my_queue, recs, seq_base, idx_mask, need7_bit, direct, ph, tp);
    else pass2_round<SKETCH, TIMING, false, false>(t, sk, L, P, my_queue, count, nullptr, recs, seq_base, idx_mask, need7_bit, direct, ph, tp);
#define NF_TICK3(k) do { if (TIMING) { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[k] += tn_ - tp; tp = tn_; } } while (0)
    pass2_flush<SKETCH>(t, sk, L, P, recs, seq_base32);
    NF_TICK3(4);                                                      // phase 4 = the flushes, phase 6 = sort + cache set-up
    const uint32_t m = P.retry_cnt;
    if (m) {
        // ---- counting sort of the retry list by sub-partition, into the free tail of this workgroup's region
        drain_stores();
        __syncthreads();
        if (tid == 0) { uint32_t o = 0; for (int s = 0; s < kSubs; s++) { P.sub_off[s] = o; o += P.sub_cnt[s]; } P.sub_off[kSubs] = o; }
        __syncthreads();
        uint32_t* sorted = my_queue + sorted_at;
        for (uint32_t k = tid; k < m; k += kBlock) {
            const uint32_t e = ald(&my_queue[k]);
            const uint32_t s = e >> kIdxBits;
            sorted[P.sub_off[s] + atomicAdd(&P.sub_fill[s], 1u)] = e;
        }
        drain_stores();
        __syncthreads();
        // The sub-partitions are folded one run after the other, several sharing a cache while it has room — a partition with
        // few misses gets one more round, not eight. Round 4: how many runs share a cache is decided by what the cache HOLDS
        // (P.fill, entries in use; the density seen so far sizes the next segment), not by the runs' record counts alone:
        // with several thousand flows per partition (10 M flows per GPU) a run has ~600 records over ~180 flows, and the
        // record-count rule (768 records surely fit) gave every run a round, a flush and a cache set-up of its own.
        constexpr uint32_t kFillLimit = 832;                      // entries a cache takes before its probe windows start to overflow
        int s = 0, runs = 0;
        bool open = false;                                        // a cache holds entries that have not been flushed
        uint32_t per_run = 0;
        while (s < kSubs) {
            const uint32_t fill = P.fill;
            __syncthreads();                                      // every lane has read it before the next round's claims raise it
            int take = 0;
            if (open) {
                per_run = fill / (uint32_t)runs + 1u;
                const uint32_t room = fill < kFillLimit ? kFillLimit - fill : 0u;
                take = (int)(room / (per_run + per_run / 4u));
                if (take == 0) {
                    pass2_flush<SKETCH>(t, sk, L, P, recs, seq_base32);
                    NF_TICK3(4);
                    open = false;
                }
            }
            if (!open) {
                cache_init(L, tid);
                if (tid == 0) { pass2_defer_ok(t, P); P.fill = 0; }
                __syncthreads();
                NF_TICK3(6);
                runs = 0;
                take = per_run ? (int)(kFillLimit / (per_run + per_run / 4u)) : 1;
                if (take < 1) take = 1;
            }
            int e = s + take < kSubs ? s + take : kSubs;
            if (!open) {                                           // into an empty cache: runs whose records surely fit come along
                uint32_t c0 = P.sub_off[e] - P.sub_off[s];
                while (e < kSubs && c0 + P.sub_cnt[e] <= kPackRecords) { c0 += P.sub_cnt[e]; e++; }
            }
            const uint32_t c = P.sub_off[e] - P.sub_off[s];
            if (c) {
                pass2_round<SKETCH, TIMING, false, true>(t, sk, L, P, sorted + P.sub_off[s], c, nullptr, recs, seq_base, idx_mask, need7_bit, direct, ph, tp);
                open = true;
            }
            runs += e - s;
            s = e;
        }
        if (open) {
            pass2_flush<SKETCH>(t, sk, L, P, recs, seq_base32);
            NF_TICK3(4);
        }
    }
#undef NF_TICK3
    if (TIMING) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[6] += tn_ - tp; }
        if ((tid & 63) == 0) for (int k = 0; k < 7; k++) aadd(&t.ctr->phase[k], ph[k]);
    }
    if (direct) aadd(&t.ctr->n_direct, direct);
}

// pass 3: the (normally empty) overflow list, one record per lane, merged directly.
template <bool SKETCH>
__global__ __launch_bounds__(256) void k_merge_overflow(TableView t, SketchView sk, SpillView q, const void* __restrict__ recs,
                                                        uint64_t n, uint64_t seq_base) {
    const uint32_t idx_mask = entry_idx_mask(n);
    uint32_t count = *q.ovf_tail;
    if (count > q.ovf_cap) count = q.ovf_cap;
    unsigned long long direct = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t qi = q.ovf[k];
        if (qi == 0xffffffffu) continue;
        const uint32_t i = qi & idx_mask;
        Rec r; uint64_t w[5];
        load_record(recs, i, r); r.canonicalize(); r.key_words(w);
        Partial p;
        partial_from_record(r, seq_base + i, p);
        upsert_partial(t, w, key_hash(w), p);
        if (SKETCH) sketch_add(sk, w, r.bytes());
        direct++;
    }
    if (direct) aadd(&t.ctr->n_direct, direct);
}

template <bool SKETCH, bool T1 = false, bool T2 = false, bool DOOR = true, int ABL = 0, bool DEEP = false, bool FREE = false>
static hipError_t run(const TableView& t, const SketchView& sk, const SpillView& q, const void* d_records, uint64_t n,
                      uint64_t seq_base, hipStream_t s) {
    const size_t lds1 = sizeof(Cache) + sizeof(Stage) + (DOOR ? sizeof(Door) : 0), lds2 = sizeof(Cache) + sizeof(Pass2Lds);
    static_assert(sizeof(Cache) + sizeof(Stage) + sizeof(Door) <= 160 * 1024, "pass 1 needs the whole LDS of a CU");
    static std::atomic<bool> attr_set_dev[64];   // per device: a process may drive several GPUs, from several host threads
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    std::atomic<bool>& attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = FREE ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1_free<SKETCH, DOOR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)
                            : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1<SKETCH, T1, DOOR, ABL, DEEP>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass2<SKETCH, T2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    const uint64_t tiles = (n + kBlock - 1) / kBlock;
    // a workgroup's cache pays for itself (set-up, one flush of up to 1024 entries — ~10 small coherent operations each, and the
    // chip retires ~24 G of those per second) only over several tiles: at least eight each
    uint64_t grid = (tiles + 7) / 8;
    if (grid < 64) grid = 64;               // ... but a 256 Ki batch still wants 64 CUs streaming it
    if (grid > 256) grid = 256;
    if (grid > tiles) grid = tiles;
    (void)hipGetLastError();
    if (FREE) hipLaunchKernelGGL((k_pass1_free<SKETCH, DOOR>), dim3((unsigned)grid), dim3(kBlock), lds1, s, t, sk, q, d_records, n, seq_base);
    else hipLaunchKernelGGL((k_pass1<SKETCH, T1, DOOR, ABL, DEEP>), dim3((unsigned)grid), dim3(kBlock), lds1, s, t, sk, q, d_records, n, seq_base);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_pass2<SKETCH, T2>), dim3(q.n_parts), dim3(kBlock), lds2, s, t, sk, q, d_records, n, seq_base);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_merge_overflow<SKETCH>), dim3(32), dim3(256), 0, s, t, sk, q, d_records, n, seq_base);   // normally empty
    return hipGetLastError();              // the overflow tail is reset by k_finalize, the last launch of every ingest call
}

}  // namespace part

hipError_t launch_ingest_part(const TableView& t, const SketchView& sk, const SpillView& q_in, const void* d_records, uint64_t n,
                              uint64_t seq_base, int variant, hipStream_t s) {
    if (!q_in.queue || !q_in.qtail || !q_in.ovf || !q_in.ovf_tail || q_in.qcap < 4 || (q_in.qcap & 3u)) return hipErrorInvalidValue;
    // Partitions scaled to the batch. One pass-2 workgroup per partition costs ~25 us whatever it folds (cache set-up, one
    // gather tile, a latency-bound flush): 2048 of them were 0.2 ms of a 0.28 ms call at 1 Mi records (round 2). About a
    // third of a Zipf batch spills; aim at ~700 spilled records (at most as many flows: they fit the 1024-entry cache) per
    // partition. The queue memory is the same, cut into fewer, longer queues.
    SpillView q = q_in;
    uint32_t parts = kSpillParts;
    while (parts > 256 && (uint64_t)parts * 2048 > n) parts >>= 1;     // 2048 from 4 Mi records, 1024 from 2 Mi, 512 from 1 Mi, 256 below
    int bits = 0;
    while ((1ull << bits) <= t.mask) bits++;
    int lg = 0;
    while ((1u << lg) < parts) lg++;
    q.n_parts = parts;
    q.part_shift = (uint32_t)(bits - lg);
    q.qcap = (uint32_t)((((uint64_t)q_in.qcap * kSpillParts) / parts) & ~3ull);
    TableView tq = t;
    tq.spill = q;
#ifdef NFAGG_DIAG
    if (variant == 8) return part::run<false, true, false>(tq, sk, q, d_records, n, seq_base, s);   // diagnostics: pass-1 phase timing
    if (variant == 9) return part::run<false, false, true>(tq, sk, q, d_records, n, seq_base, s);   // diagnostics: pass-2 phase timing
    switch (variant) {                                                                              // diagnostics: pass-1 ablations (wrong results)
        case 21: return part::run<false, false, false, true, 1>(tq, sk, q, d_records, n, seq_base, s);
        case 22: return part::run<false, false, false, true, 2>(tq, sk, q, d_records, n, seq_base, s);
        case 23: return part::run<false, false, false, true, 3>(tq, sk, q, d_records, n, seq_base, s);
        case 25: return part::run<false, false, false, true, 5>(tq, sk, q, d_records, n, seq_base, s);
        case 27: return part::run<false, false, false, true, 7>(tq, sk, q, d_records, n, seq_base, s);
        case 26: return part::run<false, false, false, true, 16>(tq, sk, q, d_records, n, seq_base, s);        // experiment, pass 1's time only (results WRONG): 64-byte queue stores, a quarter of the partitions
        case 24: return part::run<false, false, false, true, 8>(tq, sk, q, d_records, n, seq_base, s);         // experiment: wave-level duplicate combining before the LDS atomics (results RIGHT)
        case 28: return part::run<false, false, false, true, 0, true>(tq, sk, q, d_records, n, seq_base, s);   // experiment: records requested two tiles ahead (results RIGHT)
        default: break;
    }
#endif
    if (variant == 17)   // round 6: pass 1 without its barriers (published cache entries, staging groups drained by their last writer)
        return sk.flags ? part::run<true, false, false, true, 0, false, true>(tq, sk, q, d_records, n, seq_base, s)
                        : part::run<false, false, false, true, 0, false, true>(tq, sk, q, d_records, n, seq_base, s);
    if (variant == 11)   // A/B: pass 1 without the admission filter (first come, first served)
        return sk.flags ? part::run<true, false, false, false>(tq, sk, q, d_records, n, seq_base, s)
                        : part::run<false, false, false, false>(tq, sk, q, d_records, n, seq_base, s);
    return sk.flags ? part::run<true>(tq, sk, q, d_records, n, seq_base, s) : part::run<false>(tq, sk, q, d_records, n, seq_base, s);
}

}  // namespace nfagg
