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
// The order-dependent fields are resolved with per-record sequence numbers:
// "last non-zero" = atomic max of (seq+1)<<k | value; "first" = atomic max of
// ~seq, the winner writes its bytes under the slot's cold-line lock.
#pragma once
#include "nfagg_internal.h"

namespace nfagg {

#define NF_DEV __device__ __forceinline__

// ---- agent-scope relaxed atomics (global_* ... sc1): coherent across XCDs ----
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
// Wait for this wave's outstanding global stores (sc1 write-through) to be
// acknowledged before a flag store publishes them. Inline asm so the compiler
// cannot drop it (MI355X_MICROARCH.md "Compiler hazard").
NF_DEV void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

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
    NF_DEV uint32_t if_index() const { return d[21]; }      // @84
    NF_DEV uint32_t sampling() const { return d[23]; }      // @92
    NF_DEV uint32_t direction() const { return d[24] & 0xffu; }       // @96
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

NF_DEV uint64_t tag_ready(uint64_t h) { return ((h >> 2) << 2) | 3ull; }
NF_DEV uint64_t tag_locked(uint64_t h) { return ((h >> 2) << 2) | 2ull; }

// c.entries[record.Id] lookup, inserting the key when absent
// (pkg/flow/account.go:82,95). Returns the slot index, kNoSlot on probe
// overflow. Every lane advances once per loop trip and never spins inside a
// trip, so lanes of one wave that race for the same slot cannot deadlock.
NF_DEV uint32_t find_or_claim(const TableView& t, const uint64_t w[5], uint64_t h) {
    const uint64_t ready = tag_ready(h), locked = tag_locked(h);
    uint64_t idx = h & t.mask;
    uint64_t probes = 0;
    uint32_t result = kNoSlot;
    bool done = false;
    while (!done) {
        SlotHot* s = &t.hot[idx];
        uint64_t tag = ald(&s->tag);
        if (tag == 0) {
            uint64_t old = acas(&s->tag, (uint64_t)0, locked);
            if (old == 0) {
#pragma unroll
                for (int k = 0; k < 5; k++) ast(&s->key[k], w[k]);
                unsigned long long pos = aadd(&t.ctr->n_live, 1ull);
                t.live_list[pos] = (uint32_t)idx;   // read by the evict kernel only (kernel boundary)
                drain_stores();
                ast(&s->tag, ready);
                result = (uint32_t)idx; done = true;
            }
            // else: somebody else claimed it; re-examine the same slot next trip
        } else if (tag == ready) {
            bool eq = true;
#pragma unroll
            for (int k = 0; k < 5; k++) eq &= (ald(&s->key[k]) == w[k]);
            if (eq) { result = (uint32_t)idx; done = true; }
            else { idx = (idx + 1) & t.mask; probes++; }
        } else if (tag == locked) {
            // same fingerprint, key not yet published: look again next trip
        } else {
            idx = (idx + 1) & t.mask; probes++;
        }
        if (probes > t.mask) { atomicExch(&t.ctr->error, 1u); done = true; }
    }
    if (probes > 0) atomicMax(&t.ctr->max_probe, (unsigned int)probes);
    return result;
}

// What one record, or a pre-folded run of records of one key, contributes.
// Sequence numbers are absolute within the epoch. A single record is the
// trivial partial (see partial_from_record).
struct Partial {
    uint64_t bytes, end, start_inv;
    uint32_t packets, flags;
    uint64_t eth_tag, dscp_tag, samp_tag;  // 0 = no non-zero value
    uint64_t first_inv;                    // ~seq of the first record
    uint64_t smac_inv, dmac_inv;           // ~seq of first record with non-zero mac, 0 = none
    uint64_t smac, dmac;
    uint64_t ident[8];                     // first record's dwords 21..35
};

NF_DEV void partial_from_record(const Rec& r, uint64_t seq, Partial& p) {
    p.bytes = r.bytes(); p.end = r.end();
    p.start_inv = r.start() ? ~r.start() : 0ull;
    p.packets = r.packets(); p.flags = r.flags();
    const uint64_t s1 = seq + 1;
    p.eth_tag = r.eth() ? (s1 << 16) | r.eth() : 0ull;
    p.dscp_tag = r.dscp() ? (s1 << 8) | r.dscp() : 0ull;
    p.samp_tag = r.sampling() ? (s1 << 32) | r.sampling() : 0ull;
    p.first_inv = ~seq;
    p.smac = r.smac(); p.dmac = r.dmac();
    p.smac_inv = p.smac ? ~seq : 0ull;
    p.dmac_inv = p.dmac ? ~seq : 0ull;
#pragma unroll
    for (int k = 0; k < 7; k++) p.ident[k] = (uint64_t)r.d[21 + 2 * k] | ((uint64_t)r.d[22 + 2 * k] << 32);
    p.ident[7] = (uint64_t)r.d[35];
}

// model.AccumulateBase(stored, &record.Metrics) (flow_content.go:28-61) for a
// partial, plus "first record stored whole" (account.go:95), as commutative
// atomics + sequence-resolved writes.
NF_DEV void merge_partial(const TableView& t, uint32_t idx, const Partial& p) {
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
    unsigned cand = 0;
    // "<=": the careful path has already planted first_inv in its claim phase
    if (amax(&H->first_inv, p.first_inv) <= p.first_inv) cand |= 1u;
    if (p.smac_inv && amax(&C->smac_inv, p.smac_inv) <= p.smac_inv) cand |= 2u;
    if (p.dmac_inv && amax(&C->dmac_inv, p.dmac_inv) <= p.dmac_inv) cand |= 4u;
    // Rare: this partial may hold the earliest record seen so far. Under the
    // slot lock, re-check against the current winner and write the bytes.
    bool done = (cand == 0);
    while (!done) {
        if (acas(&H->lock, 0u, 1u) == 0u) {
            if ((cand & 1u) && ald(&H->first_inv) == p.first_inv) {
#pragma unroll
                for (int k = 0; k < 8; k++) ast(&C->ident[k], p.ident[k]);
            }
            if ((cand & 2u) && ald(&C->smac_inv) == p.smac_inv) ast(&C->smac, p.smac);
            if ((cand & 4u) && ald(&C->dmac_inv) == p.dmac_inv) ast(&C->dmac, p.dmac);
            drain_stores();
            ast(&H->lock, 0u);
            done = true;
        }
    }
}

}  // namespace nfagg
