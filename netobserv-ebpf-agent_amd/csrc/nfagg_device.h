// nfagg_device.h — device functions shared by the ingest / claim / evict kernels.
//
// Semantics restated from the reference (cited per function):
//   pkg/flow/account.go:81-96      lookup -> AccumulateBase | insert first record whole
//   pkg/model/flow_content.go:28-61 AccumulateBase
// A sequential fold over records r0,r1,... of one key yields (flow_content.go):
//   start  = min over non-zero starts, 0 if none        -> commutative
//   end    = max                                         -> commutative
//   bytes/packets = wrapping sums, flags = OR            -> commutative
//   eth_protocol/dscp/sampling = LAST non-zero in arrival order
//   src_mac/dst_mac            = FIRST non-all-zero in arrival order
//   every other field          = the FIRST record's value (account.go:95)
// The order-dependent fields are resolved with per-record sequence numbers and
// nothing but atomic max: "last non-zero" = max of (seq+1)<<k | value; "first"
// = max of (~seq)<<32 | dword over tagged words (nfagg_internal.h). There is no
// lock anywhere: a lock held by one lane while its wave-mates spin deadlocks
// under SIMT execution.
#pragma once
#include "nfagg_internal.h"

namespace nfagg {

#define NF_DEV __device__ __forceinline__

// ---- agent-scope relaxed atomics: coherent across the 8 XCDs ----
template <typename T> NF_DEV T ald(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> NF_DEV void ast(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> NF_DEV T aadd(T* p, T v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> NF_DEV T aor(T* p, T v) {
    return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> NF_DEV T amax(T* p, T v) {
    return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> NF_DEV T acas(T* p, T expected, T desired) {
    __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    return expected;  // old value
}
// Wait for this wave's outstanding global stores (write-through) to be
// acknowledged before a flag store publishes them. Inline asm so the compiler
// cannot drop it (MI355X_MICROARCH.md "Compiler hazard").
NF_DEV void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// A 16-byte agent-scope (write-through, sc1) store: what ast() is for 8 bytes, half the fabric writes for a slot's lines.
typedef unsigned int nf_u32x4 __attribute__((ext_vector_type(4)));
NF_DEV void ast16(void* p, uint64_t lo, uint64_t hi) {
    nf_u32x4 v;
    v.x = (unsigned int)lo; v.y = (unsigned int)(lo >> 32); v.z = (unsigned int)hi; v.w = (unsigned int)(hi >> 32);
    // The hazard pad is INSIDE the string: hipcc treats an asm statement as one opaque instruction and does not know that it is a
    // 128-bit store, so it neither delays the next write to v[] (gfx940+: a VALU write to the data registers of a store of more
    // than 64 bits needs 2 wait states) nor keeps the registers out of reuse. Round 4 found out the hard way: a change elsewhere
    // moved the register allocation, the claimer's key word 3 went out as the address computed for its NEXT store into the same
    // registers, and ~3 % of the fresh slots carried a key that no record has (tests/tools/dedup_anatomy.py, profiles/r04_ast16_hazard.txt).
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kSpinLimit = 1u << 22;

NF_DEV uint64_t tagged(uint32_t inv_seq, uint32_t data) { return ((uint64_t)inv_seq << 32) | data; }

// A 144-byte flow_record_t in registers, as 36 little-endian dwords.
struct Rec {
    uint32_t d[kRecordDwords];
    NF_DEV uint64_t q(int i) const { return (uint64_t)d[2 * i] | ((uint64_t)d[2 * i + 1] << 32); }
    // metrics fields (record dword = (40 + metrics offset) / 4)
    NF_DEV uint64_t start() const { return q(5); }          // @40
    NF_DEV uint64_t end() const { return q(6); }            // @48
    NF_DEV uint64_t bytes() const { return q(7); }          // @56
    NF_DEV uint32_t packets() const { return d[16]; }       // @64
    NF_DEV uint32_t eth() const { return d[17] & 0xffffu; } // @68
    NF_DEV uint32_t flags() const { return d[17] >> 16; }   // @70
    NF_DEV uint64_t smac() const { return (uint64_t)d[18] | ((uint64_t)(d[19] & 0xffffu) << 32); } // @72..77
    NF_DEV uint64_t dmac() const { return (uint64_t)(d[19] >> 16) | ((uint64_t)d[20] << 16); }     // @78..83
    NF_DEV uint32_t sampling() const { return d[23]; }      // @92
    NF_DEV uint32_t dscp() const { return (d[24] >> 16) & 0xffu; }    // @98
    // Zero the bytes Go never sees: key byte 39 (bpf_x86_bpfel.go:119 blank
    // field), metrics pad2 @66-67 and pad4 @100-103.
    NF_DEV void canonicalize() {
        d[9] &= 0x00ffffffu;
        d[26] &= 0x0000ffffu;
        d[35] = 0;
    }
    NF_DEV void key_words(uint64_t w[5]) const {
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = q(i);
    }
};

NF_DEV void load_record(const void* base, uint64_t i, Rec& r) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + i * kRecordBytes);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint4 v = p[k];
        r.d[4 * k] = v.x; r.d[4 * k + 1] = v.y; r.d[4 * k + 2] = v.z; r.d[4 * k + 3] = v.w;
    }
}

// tag = epoch (16 bits) | fingerprint = top 46 bits of the key hash | state
constexpr uint64_t kEpochMask = 0xFFFFull << 48;
NF_DEV uint64_t tag_ready(const TableView& t, uint64_t h) { return t.epoch_bits | ((h >> 18) << 2) | 3ull; }
NF_DEV uint64_t tag_locked(const TableView& t, uint64_t h) { return t.epoch_bits | ((h >> 18) << 2) | 2ull; }
NF_DEV bool tag_is_free(const TableView& t, uint64_t tag) { return (tag & kEpochMask) != t.epoch_bits; }

// Possibly stale copies of the slot's monotone words, read with plain 16-byte
// loads. Every one of these words only ever grows (max / OR) within an epoch and
// starts at zero, so a stale value is a LOWER bound: using it to skip an atomic
// that could not change the word is safe, a stale value merely skips less.
struct Hints {
    uint64_t end, start_inv, id0, smac_lo, dmac_lo;
    uint32_t flags;
    uint64_t home_tag;     // probe_home only: the tag it saw in the home slot (find_or_claim's first look, saved a round trip)
};

NF_DEV void load_hints(const SlotHot* H, Hints& x) {
    const uint4* L = reinterpret_cast<const uint4*>(H);
    const uint4 l3 = L[3], l4 = L[4], l6 = L[6], l7 = L[7];
    x.end = (uint64_t)l3.z | ((uint64_t)l3.w << 32);
    x.start_inv = (uint64_t)l4.x | ((uint64_t)l4.y << 32);
    x.flags = l4.w;
    x.id0 = (uint64_t)l6.z | ((uint64_t)l6.w << 32);
    x.smac_lo = (uint64_t)l7.x | ((uint64_t)l7.y << 32);
    x.dmac_lo = (uint64_t)l7.z | ((uint64_t)l7.w << 32);
}

// One-round-trip fast path: issue the three key loads and the four hint loads of
// the flow's home slot together (plain, cacheable 16-byte loads) and decide
// afterwards. Returns the slot when the home slot is `ready` and holds this key
// (then `x` holds its hints); kNoSlot means "not decided" — the caller runs the
// coherent find_or_claim loop. Safety of the plain loads: see find_or_claim.
// kx != 0: a sixth key word, kept in SlotHot.end (sub-flow tables of the kernel-dedup mode, nfagg_dedup.h; `end` is unused there).
NF_DEV uint32_t probe_home(const TableView& t, const uint64_t w[5], uint64_t h, Hints& x, uint64_t kx = 0) {
    const uint64_t idx = h & t.mask;
    const uint4* L = reinterpret_cast<const uint4*>(&t.hot[idx]);
    const uint4 a = L[0], b = L[1], c = L[2], l3 = L[3], l4 = L[4], l6 = L[6], l7 = L[7];
    const bool eq = (((uint64_t)a.x | ((uint64_t)a.y << 32)) == tag_ready(t, h)) &
                    (((uint64_t)a.z | ((uint64_t)a.w << 32)) == w[0]) &
                    (((uint64_t)b.x | ((uint64_t)b.y << 32)) == w[1]) & (((uint64_t)b.z | ((uint64_t)b.w << 32)) == w[2]) &
                    (((uint64_t)c.x | ((uint64_t)c.y << 32)) == w[3]) & (((uint64_t)c.z | ((uint64_t)c.w << 32)) == w[4]);
    x.end = (uint64_t)l3.z | ((uint64_t)l3.w << 32);
    x.start_inv = (uint64_t)l4.x | ((uint64_t)l4.y << 32);
    x.flags = l4.w;
    x.id0 = (uint64_t)l6.z | ((uint64_t)l6.w << 32);
    x.smac_lo = (uint64_t)l7.x | ((uint64_t)l7.y << 32);
    x.dmac_lo = (uint64_t)l7.z | ((uint64_t)l7.w << 32);
    x.home_tag = (uint64_t)a.x | ((uint64_t)a.y << 32);
    return (eq & ((kx == 0) | (x.end == kx))) ? (uint32_t)idx : kNoSlot;
}

// What one record, or a pre-folded run of records of one key, contributes.
// Sequence numbers are epoch-relative 32-bit. A single record is the trivial
// partial (partial_from_record).
struct Partial {
    uint64_t bytes, end, start_inv;
    uint32_t packets, flags;
    uint64_t eth_tag, dscp_tag, samp_tag;  // 0 = no non-zero value
    uint32_t first_inv;                    // ~seq of the first record
    uint32_t smac_inv, dmac_inv;           // ~seq of the first record with a non-zero mac; 0 = none
    uint64_t smac, dmac;                   // 48-bit
    uint32_t ident0;                       // first record's dword 21 (if_index_first_seen); the other identity dwords are
                                           // copied from the batch by k_finalize, no fold kernel carries them
};

// c.entries[record.Id] lookup, inserting the key when absent
// (pkg/flow/account.go:82,95): the coherent path. Callers try probe_home first.
// Why probe_home's plain loads are safe: (a) a `ready` tag is only ever published
// after the key's write-through stores have been acknowledged, (b) a cache
// returns one line's words no older than a word it returned before, so key words
// read together with a `ready` tag from the same 128-byte line belong to it, and
// (c) anything else it may see (empty, being claimed, another flow) may be stale
// and is re-examined here with agent-scope atomics.
// Returns the slot index, kNoSlot on failure
// (error code left in the counters). Every lane advances at most one state per
// loop trip and never waits inside a trip; the claimer publishes its key in the
// trip in which its CAS succeeded. The empty asm keeps `done` opaque so the
// compiler cannot move the publishing code out of the loop (where it would run
// only after every wave-mate had left the loop: SIMT deadlock).
// DEFER (pass 2 of the two-pass fold, which owns its partition's flows): a successful claim only takes the slot
// (tag = claimed) and reports `fresh`; the caller writes key and values itself, publishes the tag and registers the slot
// in the live list through its workgroup (one n_live atomic per workgroup instead of one per wave).
// home_tag: what probe_home's plain load saw in the home slot, used as the first trip's look at it. A stale value is
// harmless: "free" is verified by the CAS (which fails and leads to a real load), "another flow's" cannot go back to free
// within a fold that stands (only the claim undo of an aborted — rolled back — fold empties a tag), anything else is looked
// at again.
// init (non-DEFER callers that hold the flow's partial already, i.e. the cache flushes): a lane that CLAIMS the slot writes
// the partial as the slot's first value instead of zeroes — it owns the slot until the tag says `ready` — and reports it in
// *fresh: the caller skips the merge (12 atomics and 4 hint loads less per new flow; a flush is bound by the number of small
// coherent operations the chip retires, ~24 G/s).
// kx != 0: a sixth key word, written to and compared with SlotHot.end (sub-flow tables of the kernel-dedup mode, nfagg_dedup.h).
template <bool DEFER = false>
NF_DEV uint32_t find_or_claim(const TableView& t, const uint64_t w[5], uint64_t h, bool* fresh = nullptr, const uint64_t* home_tag = nullptr,
                              const Partial* init = nullptr, uint64_t kx = 0) {
    const uint64_t ready = tag_ready(t, h), locked = tag_locked(t, h);
    uint64_t idx = h & t.mask;
    uint64_t probes = 0;
    uint32_t result = kNoSlot;
    uint32_t done = 0;
    uint32_t trips = 0;
    do {
        if (++trips > kSpinLimit) { atomicExch(&t.ctr->error, 2u); break; }
        SlotHot* s = &t.hot[idx];
        uint64_t tag = (home_tag && trips == 1) ? *home_tag : ald(&s->tag);
        if (tag_is_free(t, tag)) {       // never used, or left over from an earlier epoch (eviction does not clear the table)
            // Claims are bounded: at most claim_limit slots are ever claimed in an epoch, however many new keys an
            // (optimistically folded, nfagg_api.hip) batch holds. The position handed out by the n_live increment decides
            // (a separate look at n_live before the CAS cost 2.5x in pass 2: every claimer of the chip loading the one
            // word all of them increment): a claim beyond the limit is undone — tag back to empty, the increment taken
            // back; positions below the limit stay dense and unique because n_live never drops below the limit once it
            // got there. A refused claim drops this lane's contribution and raises `aborted`: the API rolls the whole
            // batch back and folds a shorter prefix, so what the table holds meanwhile is moot.
            const uint64_t old = acas(&s->tag, tag, locked);
            if (DEFER && old == tag) { *fresh = true; result = (uint32_t)idx; done = 1; }
            else if (old == tag) {
                // The lanes of the wave that won a slot in this trip reserve their live-list positions with ONE atomic (they are
                // exactly the lanes in this branch). A returning atomic on one address retires every ~14 ns however many lanes
                // issue it: one per claimed slot was ~0.25 ms of every ingest call into an empty table (the flush of pass 1's
                // caches claims the ~20 k hottest flows), whatever the batch size.
                const unsigned long long wm = __ballot(1);
                const int lane_ = (int)(threadIdx.x & 63), leader_ = __ffsll((long long)wm) - 1;
                unsigned long long base_ = 0;
                if (lane_ == leader_) base_ = aadd(&t.ctr->n_live, (unsigned long long)__popcll(wm));
                base_ = __shfl(base_, leader_);
                const unsigned long long pos = base_ + (unsigned long long)__popcll(wm & ((1ull << lane_) - 1ull));
                if (pos >= t.claim_limit) {
                    ast(&s->tag, (uint64_t)0);
                    aadd(&t.ctr->n_live, ~0ull);
                    atomicExch(&t.ctr->aborted, 1u);
                } else {
                    // the claimer owns the slot until it publishes `ready`: key, then every word the folds combine into
                    // back to its identity — or straight to the claimer's partial (the slot may hold a flow of an earlier
                    // epoch). 16-byte write-through stores: 9 fabric writes instead of 17.
                    uint64_t* hw = reinterpret_cast<uint64_t*>(s);
                    uint64_t v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ch[2] = {0, 0};
                    if (init) {
                        v[0] = init->bytes; v[1] = init->end; v[2] = init->start_inv;
                        v[3] = (uint64_t)init->packets | ((uint64_t)init->flags << 32);
                        v[4] = init->eth_tag; v[5] = init->dscp_tag; v[6] = init->samp_tag;
                        v[7] = tagged(init->first_inv, init->ident0);
                        if (init->smac_inv) { v[8] = tagged(init->smac_inv, (uint32_t)init->smac); ch[0] = tagged(init->smac_inv, (uint32_t)(init->smac >> 32)); }
                        if (init->dmac_inv) { v[9] = tagged(init->dmac_inv, (uint32_t)init->dmac); ch[1] = tagged(init->dmac_inv, (uint32_t)(init->dmac >> 32)); }
                        *fresh = true;
                    }
                    if (kx) v[1] = kx;
                    ast(&hw[1], w[0]);
                    ast16(&hw[2], w[1], w[2]);
                    ast16(&hw[4], w[3], w[4]);
#pragma unroll
                    for (int k = 0; k < 5; k++) ast16(&hw[6 + 2 * k], v[2 * k], v[2 * k + 1]);
                    ast16(&t.cold[idx], ch[0], ch[1]);
                    if (t.aux) {
                        // kernel-dedup mode: a claimer that brings its first-record tag (init: first_inv, ident0 = the
                        // interface; everything else zero) is also the first candidate interface (nfagg_dedup.h dedup_claim)
                        uint64_t* aw = reinterpret_cast<uint64_t*>(&t.aux[idx]);
                        ast16(&aw[0], (init && init->ident0) ? tagged(init->first_inv, init->ident0) : 0ull, 0);
                        for (int k = 2; k < (int)(sizeof(SlotAux) / 8); k += 2) ast16(&aw[k], 0, 0);
                    }
                    t.live_list[pos] = (uint32_t)idx;   // read by the finalize / evict kernels only (kernel boundary)
                    drain_stores();
                    ast(&s->tag, ready);
                    result = (uint32_t)idx;
                }
                done = 1;
            }
            // else: somebody else claimed it; re-examine the same slot next trip
        } else if (tag == ready) {
            bool eq = true;
#pragma unroll
            for (int k = 0; k < 5; k++) eq &= (ald(&s->key[k]) == w[k]);
            if (kx) eq &= (ald(&s->end) == kx);
            if (eq) { result = (uint32_t)idx; done = 1; }
            else { idx = (idx + 1) & t.mask; probes++; }
        } else if (tag == locked && !DEFER) {
            // same fingerprint, key (and zeroes) not yet published: look again next trip. (DEFER — pass 2, where every flow
            // has ONE claimer on the whole chip, this lane: a slot being claimed is another flow's whatever its fingerprint,
            // and its claimer may be a lane of this workgroup that publishes only after the flush's barrier — move on.)
        } else {
            idx = (idx + 1) & t.mask; probes++;
        }
        if (probes > t.mask) { atomicExch(&t.ctr->error, 1u); done = 1; }
        asm volatile("" : "+v"(done));
    } while (!done);
    if (probes > 0) atomicMax(&t.ctr->max_probe, (unsigned int)probes);
    return result;
}

NF_DEV void partial_from_record(const Rec& r, uint64_t seq, Partial& p) {
    p.bytes = r.bytes(); p.end = r.end();
    p.start_inv = r.start() ? ~r.start() : 0ull;
    p.packets = r.packets(); p.flags = r.flags();
    const uint64_t s1 = seq + 1;
    p.eth_tag = r.eth() ? (s1 << 16) | r.eth() : 0ull;
    p.dscp_tag = r.dscp() ? (s1 << 8) | r.dscp() : 0ull;
    p.samp_tag = r.sampling() ? (s1 << 32) | r.sampling() : 0ull;
    p.first_inv = ~(uint32_t)seq;
    p.smac = r.smac(); p.dmac = r.dmac();
    p.smac_inv = p.smac ? ~(uint32_t)seq : 0u;
    p.dmac_inv = p.dmac ? ~(uint32_t)seq : 0u;
    p.ident0 = r.d[21];
}

// model.AccumulateBase(stored, &record.Metrics) (flow_content.go:28-61) for a
// partial, plus "first record stored whole" (account.go:95): commutative
// atomics and tagged-word maxima only; atomics that the hints prove to be
// no-ops are skipped.
NF_DEV void merge_partial(const TableView& t, uint32_t idx, const Partial& p, const Hints& x) {
    SlotHot* H = &t.hot[idx];
    SlotCold* C = &t.cold[idx];
    if (p.bytes) aadd(&H->bytes, p.bytes);
    if (p.packets) aadd(&H->packets, p.packets);
    if (p.flags & ~x.flags) aor(&H->flags, p.flags);
    if (p.end > x.end) amax(&H->end, p.end);
    if (p.start_inv > x.start_inv) amax(&H->start_inv, p.start_inv);
    if (p.eth_tag) amax(&H->eth_tag, p.eth_tag);
    if (p.dscp_tag) amax(&H->dscp_tag, p.dscp_tag);
    if (p.samp_tag) amax(&H->samp_tag, p.samp_tag);
    // First record: only its sequence number (the tag) has to win; "<=" because the careful path plants id0 in its
    // claim phase. The identity dwords follow in k_finalize.
    const uint64_t my0 = tagged(p.first_inv, p.ident0);
    if (x.id0 <= my0) amax(&H->id0, my0);
    if (p.smac_inv) {
        const uint64_t lo = tagged(p.smac_inv, (uint32_t)p.smac);
        if (x.smac_lo <= lo) { amax(&H->smac_lo, lo); amax(&C->smac_hi, tagged(p.smac_inv, (uint32_t)(p.smac >> 32))); }
    }
    if (p.dmac_inv) {
        const uint64_t lo = tagged(p.dmac_inv, (uint32_t)p.dmac);
        if (x.dmac_lo <= lo) { amax(&H->dmac_lo, lo); amax(&C->dmac_hi, tagged(p.dmac_inv, (uint32_t)(p.dmac >> 32))); }
    }
}

// merge_partial for a lane that OWNS the flow for the duration of the kernel: pass 2 of the two-pass fold, where a
// flow's records all went to the queue of its partition and one workgroup folds that queue, each flow in one cache
// entry — no other lane of the chip touches the value words of this slot before the kernel ends (other workgroups only
// probe its tag and key; pass 1 and pass 3 are other kernels). So the ~12 atomics of merge_partial become a plain
// read-modify-write of the line: five 16-byte loads, the operators of flow_content.go:28-61 in registers, 16-byte stores.
// fresh: the slot was claimed (tag = claimed) by this lane just now and holds stale data: nothing is loaded, the
// partial IS the value; key (write-through, other workgroups compare it), values, then the tag is published.
// publish = false: the caller publishes the tag of a fresh slot itself, after drain_stores() (pass2_flush: one wait for the whole
// workgroup, overlapped with the live-list reservation).
NF_DEV void merge_partial_exclusive(const TableView& t, uint32_t idx, const Partial& p, bool fresh, const uint64_t w[5], uint64_t h,
                                    bool publish = true) {
    uint4* HL = reinterpret_cast<uint4*>(&t.hot[idx]);
    SlotCold* C = &t.cold[idx];
    uint64_t bytes = 0, end = 0, start_inv = 0, eth_tag = 0, dscp_tag = 0, samp_tag = 0, id0 = 0, smac_lo = 0, dmac_lo = 0;
    uint32_t packets = 0, flags = 0;
    if (!fresh) {
        const uint4 l3 = HL[3], l4 = HL[4], l5 = HL[5], l6 = HL[6], l7 = HL[7];
        bytes = (uint64_t)l3.x | ((uint64_t)l3.y << 32); end = (uint64_t)l3.z | ((uint64_t)l3.w << 32);
        start_inv = (uint64_t)l4.x | ((uint64_t)l4.y << 32); packets = l4.z; flags = l4.w;
        eth_tag = (uint64_t)l5.x | ((uint64_t)l5.y << 32); dscp_tag = (uint64_t)l5.z | ((uint64_t)l5.w << 32);
        samp_tag = (uint64_t)l6.x | ((uint64_t)l6.y << 32); id0 = (uint64_t)l6.z | ((uint64_t)l6.w << 32);
        smac_lo = (uint64_t)l7.x | ((uint64_t)l7.y << 32); dmac_lo = (uint64_t)l7.z | ((uint64_t)l7.w << 32);
    }
    bytes += p.bytes; packets += p.packets; flags |= p.flags;
    if (p.end > end) end = p.end;
    if (p.start_inv > start_inv) start_inv = p.start_inv;
    if (p.eth_tag > eth_tag) eth_tag = p.eth_tag;
    if (p.dscp_tag > dscp_tag) dscp_tag = p.dscp_tag;
    if (p.samp_tag > samp_tag) samp_tag = p.samp_tag;
    const uint64_t my0 = tagged(p.first_inv, p.ident0);
    if (my0 > id0) id0 = my0;
    // the high MAC halves live in the cold half line: written when they change, and always for a fresh slot (stale data)
    bool st_s = fresh, st_d = fresh;
    uint64_t smac_hi = 0, dmac_hi = 0;
    if (p.smac_inv) {
        const uint64_t lo = tagged(p.smac_inv, (uint32_t)p.smac);
        if (lo >= smac_lo) { smac_lo = lo; smac_hi = tagged(p.smac_inv, (uint32_t)(p.smac >> 32)); st_s = true; }
    }
    if (p.dmac_inv) {
        const uint64_t lo = tagged(p.dmac_inv, (uint32_t)p.dmac);
        if (lo >= dmac_lo) { dmac_lo = lo; dmac_hi = tagged(p.dmac_inv, (uint32_t)(p.dmac >> 32)); st_d = true; }
    }
    if (st_s) C->smac_hi = smac_hi;
    if (st_d) C->dmac_hi = dmac_hi;
    HL[3] = make_uint4((uint32_t)bytes, (uint32_t)(bytes >> 32), (uint32_t)end, (uint32_t)(end >> 32));
    HL[4] = make_uint4((uint32_t)start_inv, (uint32_t)(start_inv >> 32), packets, flags);
    HL[5] = make_uint4((uint32_t)eth_tag, (uint32_t)(eth_tag >> 32), (uint32_t)dscp_tag, (uint32_t)(dscp_tag >> 32));
    HL[6] = make_uint4((uint32_t)samp_tag, (uint32_t)(samp_tag >> 32), (uint32_t)id0, (uint32_t)(id0 >> 32));
    HL[7] = make_uint4((uint32_t)smac_lo, (uint32_t)(smac_lo >> 32), (uint32_t)dmac_lo, (uint32_t)(dmac_lo >> 32));
    if (fresh) {
        SlotHot* H = &t.hot[idx];
#pragma unroll
        for (int k = 0; k < 5; k++) ast(&H->key[k], w[k]);
        if (publish) {
            drain_stores();
            ast(&H->tag, tag_ready(t, h));
        }
    }
}

// Sketch contribution of one (IP, byte count) on one side (0 = src, 1 = dst): Count-Min is linear, so a pre-folded run of
// records contributes its byte SUM once; HyperLogLog is idempotent. Spec: nfagg_sketch.hip / DESIGN.md §6.
NF_DEV void sketch_add_side(const SketchView& sk, int side, uint64_t lo, uint64_t hi, uint64_t bytes) {
    if ((sk.flags & 1u) && bytes) {
        const uint64_t ha = ip_hash(lo, hi, 0), hb = ip_hash(lo, hi, 1) | 1ull;
        for (uint32_t r = 0; r < sk.cm_depth; r++)
            aadd(&sk.cm[side][((uint64_t)r << sk.cm_log2w) + cm_index(ha, hb, r, sk.cm_log2w)], bytes);
    }
    if (sk.flags & 2u) {
        const uint64_t h = ip_hash(lo, hi, 2);
        const uint64_t idx = h >> (64 - sk.hll_p);
        const uint32_t rho = (uint32_t)__clzll((long long)((h << sk.hll_p) | (1ull << (sk.hll_p - 1)))) + 1u;
        // One byte per register (what the all-reduce moves): raised by a CAS on the 32-bit word that holds it.
        // Registers only grow and saturate quickly: the plain read skips nearly every update (a stale smaller value merely costs
        // one look at the word), and a CAS only fails when a neighbour in the same word was raised meanwhile.
        if (sk.hll[side][idx] < rho) {
            uint32_t* wp = reinterpret_cast<uint32_t*>(sk.hll[side] + (idx & ~3ull));
            const uint32_t sh = (uint32_t)(idx & 3ull) * 8u;
            uint32_t cur = ald(wp);
            while (((cur >> sh) & 0xffu) < rho) {
                const uint32_t old = acas(wp, cur, (cur & ~(0xffu << sh)) | (rho << sh));
                if (old == cur) break;
                cur = old;
            }
        }
    }
}

NF_DEV void sketch_add(const SketchView& sk, const uint64_t w[5], uint64_t bytes) {
    sketch_add_side(sk, 0, w[0], w[1], bytes);
    sketch_add_side(sk, 1, w[2], w[3], bytes);
}

// lookup-or-insert + merge of one partial: home-slot fast path, coherent loop otherwise
NF_DEV void upsert_partial(const TableView& t, const uint64_t w[5], uint64_t h, const Partial& p) {
    Hints x;
    uint32_t idx = probe_home(t, w, h, x);
    if (idx == kNoSlot) {
        idx = find_or_claim(t, w, h);
        if (idx == kNoSlot) return;
        load_hints(&t.hot[idx], x);
    }
    merge_partial(t, idx, p, x);
}

}  // namespace nfagg
