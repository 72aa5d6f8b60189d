// nfagg_dedup.h — device functions shared by the kernel-dedup kernels (nfagg_dedup.hip:
// direct per-record passes + evict; nfagg_dedup_cached.hip: the LDS-cached streaming + partition passes).
// Semantics and the two-pass scheme are described at the top of nfagg_dedup.hip.
#pragma once
#include "nfagg_device.h"

namespace nfagg {

constexpr uint32_t kObservedMax = 6;      // MAX_OBSERVED_INTERFACES (bpf/types.h)
constexpr uint32_t kDirBoth = 3;          // OBSERVED_DIRECTION_BOTH
constexpr uint32_t kTlsServerHello = 2;   // TLSTRACKER_BF_SERVER_HELLO (bpf/tls_tracker.h)
constexpr uint32_t kMiscSslMismatch = 1;  // MISC_FLAGS_SSL_MISMATCH
constexpr int kCand = 7;

// Keep the K largest words among the best word per distinct match-key (the bits under
// MATCH) in w[0..K). Larger word = earlier record ((~seq) in the high bits). Lock-free:
//   * a word with my key and a value >= mine: nothing to do;
//   * a word with my key and a smaller value: CAS it up;
//   * otherwise replace the smallest word (ties: lowest index; empty = 0) when mine is larger.
// Every position only ever grows, and a word is installed by a CAS on the smallest
// position of a snapshot, so one key can never sit in two positions (if it did, both
// installers would have seen the other's position as small as their own, which strict
// growth only allows for two empty positions — excluded by the lowest-index rule).
// A failed CAS means another lane made progress; nobody waits on anybody.
template <int K, uint64_t MATCH>
NF_DEV void topk_insert(const TableView& t, uint64_t* w, uint64_t v) {
    for (uint32_t trip = 0; trip < kSpinLimit; trip++) {
        // snapshot; pos/cur = the word holding my key if any, else the smallest word (lowest index on ties)
        uint64_t cur = ald(&w[0]);
        int pos = 0;
        bool hit = cur != 0 && ((cur ^ v) & MATCH) == 0;
#pragma unroll
        for (int k = 1; k < K; k++) {
            const uint64_t c = ald(&w[k]);
            const bool mine = c != 0 && ((c ^ v) & MATCH) == 0;
            if (mine || (!hit && c < cur)) { cur = c; pos = k; }
            hit |= mine;
        }
        if (cur >= v) return;
        if (acas(&w[pos], cur, v) == cur) return;
    }
    atomicExch(&t.ctr->error, 4u);
}

// the part of record_prologue after the load (for loops that request the next tile's record ahead of time)
NF_DEV bool record_keys(const TableView& t, Rec& r, uint64_t w[5], uint64_t& h) {
    r.canonicalize();
    r.key_words(w);
    h = key_hash(w);
    return !(t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id);
}

NF_DEV bool record_prologue(const TableView& t, const void* recs, uint64_t i, Rec& r, uint64_t w[5], uint64_t& h) {
    load_record(recs, i, r);
    r.canonicalize();
    r.key_words(w);
    h = key_hash(w);
    return !(t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id);
}


// ---- Local fold across GPUs (TableView.subflow; DESIGN.md §7 a'). What a table counts for a flow depends on the flow's first
// interface F (flows.c:100-126), and with several GPUs folding parts of ONE stream F is the interface of the earliest record
// ANYWHERE. A table that may not hold a flow's earliest record therefore keeps nothing that depends on F: it is keyed by the
// SUB-FLOW (flow key, if_index_first_seen) — the sixth key word lives in SlotHot.end, which this mode does not use — and every
// slot holds BOTH roles of its interface: the sums, "last value" tags and TLS words of a counted interface, and the two earliest
// directions of a side interface (dir[0]), plus its own first record stored whole. Such slots merge across tables word by word
// (nfagg_combine.hip); the epoch ends with a join (nfagg_dedup_join.hip): every sub-flow goes through the two phases below
// into a flow-keyed table, F being the interface of the sub-flow with the smallest first sequence number.
NF_DEV uint64_t sub_kx(const TableView& t, uint32_t ifx) { return t.subflow ? (1ull << 32) | (uint64_t)ifx : 0ull; }
// slot hash of a sub-flow: the partition bits (the top 11 bits of the home slot, SpillView.part_shift) stay the FLOW's — the
// partition pass owns a flow's sub-flows together and its claims stay inside one window of the table — everything else is mixed
// with the interface.
NF_DEV uint64_t sub_hash(const TableView& t, uint64_t h, uint32_t ifx) {
    if (!t.subflow) return h;
    uint64_t z = (h ^ ((uint64_t)ifx * 0xD6E8FEB86659FD93ull + 0x9E3779B97F4A7C15ull)) * kMul;
    z ^= z >> 29;
    const uint64_t pm = (uint64_t)(kSpillParts - 1) << t.spill.part_shift;
    return (z & ~pm) | (h & pm);
}

// ---- pass 1 for one (flow, interface): first record + earliest interfaces. `seq32` is the
// smallest sequence number of the records represented (a single record, or a cached run).
NF_DEV void dedup_claim(const TableView& t, uint32_t idx, uint64_t id0_hint, uint32_t ifx, uint32_t seq32) {
    const uint32_t inv = ~seq32;
    const uint64_t my0 = tagged(inv, ifx);                      // record dword 21 IS if_index_first_seen
    if (id0_hint < my0) amax(&t.hot[idx].id0, my0);
    if (ifx != 0 && !t.subflow) {                               // (a sub-flow slot has one interface: its own)
        // cheap exit on possibly stale plain loads: a word that once held this interface with an
        // earlier-or-equal record makes this record irrelevant for good (see topk_insert)
        const uint64_t* cw = t.aux[idx].cand;
        bool known = false;
#pragma unroll
        for (int k = 0; k < kCand; k++) { const uint64_t c = cw[k]; known |= ((uint32_t)c == ifx) & (c >= my0); }
        if (!known) topk_insert<kCand, 0xffffffffull>(t, t.aux[idx].cand, my0);
    }
}

// What one record, or a pre-folded run of records of ONE (flow, interface), contributes in pass 2.
// Everything is kept for both roles; which role applies is only known from the slot (F).
struct DedupPartial {
    uint64_t bytes;
    uint32_t packets, flags;            // flags | tls_types << 16
    uint64_t endl_lo, endl_hi;          // (seq+1)<<32 | half of end
    uint64_t dscp_tag, samp_tag;        // (seq+1)<<8 | dscp, (seq+1)<<32 | sampling — zero values included
    uint64_t ssl_first;                 // tagged first non-zero ssl_version; 0 = none
    uint32_t ssl_max, ssl_minv;
    uint64_t cs_tag, ks_tag;            // server-hello cipher suite / key share, last; 0 = none
    uint64_t dir0, dir1;                // two earliest distinct directions, (~seq)<<8 | direction; 0 = none
    uint32_t ifx;
};

NF_DEV void dedup_partial_from_record(const Rec& r, uint32_t seq32, DedupPartial& p) {
    const uint32_t inv = ~seq32;
    const uint64_t s1 = (uint64_t)seq32 + 1;
    const uint32_t types = (r.d[34] >> 16) & 0xffu;
    p.bytes = r.bytes(); p.packets = r.packets(); p.flags = r.flags() | (types << 16);
    const uint64_t e = r.end();
    p.endl_lo = (s1 << 32) | (uint32_t)e; p.endl_hi = (s1 << 32) | (uint32_t)(e >> 32);
    p.dscp_tag = (s1 << 8) | r.dscp(); p.samp_tag = (s1 << 32) | r.sampling();
    const uint32_t ssl = r.d[33] & 0xffffu;
    p.ssl_first = ssl ? tagged(inv, ssl) : 0ull; p.ssl_max = ssl; p.ssl_minv = ssl ? 0x10000u - ssl : 0u;
    const uint32_t cs = r.d[33] >> 16, ks = r.d[34] & 0xffffu;
    p.cs_tag = (cs && types == kTlsServerHello) ? (s1 << 16) | cs : 0ull;
    p.ks_tag = (ks && types == kTlsServerHello) ? (s1 << 16) | ks : 0ull;
    p.dir0 = ((uint64_t)inv << 8) | (r.d[24] & 0xffu); p.dir1 = 0;
    p.ifx = r.d[21];
}

// update_existing_flow (flows.c:98-143) for a partial; x.id0 (exact: every claim of the flow is done) gives F.
// COHERENT: the claims were made earlier in THIS kernel (by the caller's workgroup, a barrier since): the candidate list is read
// past the caches. Otherwise the claim pass is a previous kernel and plain loads see it.
template <bool COHERENT = false>
NF_DEV void dedup_merge(const TableView& t, uint32_t idx, const Hints& x, const DedupPartial& p) {
    SlotHot* H = &t.hot[idx];
    SlotAux* A = &t.aux[idx];
    const bool sub = t.subflow != 0;                             // sub-flow slot: its interface plays both roles until the join
    const uint32_t F = sub ? p.ifx : (uint32_t)x.id0;
    const bool counted = p.ifx == F;
    if (!counted && p.ifx == 0) return;                          // flows.c:126: `else if (if_index != 0)`
    // end = r.end, by the LAST record that reaches either branch (flows.c:108,128)
    amax(&A->endl_lo, p.endl_lo);
    amax(&A->endl_hi, p.endl_hi);
    uint32_t fl = p.flags & 0xffffu;
    if (counted) {
        if (p.bytes) aadd(&H->bytes, p.bytes);
        if (p.packets) aadd(&H->packets, p.packets);
        fl = p.flags;                                            // tls_types |= (flows.c:125)
        amax(&H->dscp_tag, p.dscp_tag);                          // dscp = pkt->dscp (zero included)
        amax(&H->samp_tag, p.samp_tag);                          // sampling = sampling
        if (p.ssl_first) {
            amax(&A->ssl_first, p.ssl_first);
            atomicMax(&A->ssl_max, p.ssl_max);
            atomicMax(&A->ssl_minv, p.ssl_minv);
        }
        if (p.cs_tag) amax(&A->cs_tag, p.cs_tag);
        if (p.ks_tag) amax(&A->ks_tag, p.ks_tag);
    }
    if (!counted || sub) {
        // side records: remember the two earliest distinct directions of their interface
        int pos = sub ? 0 : -1;
        if (!sub) {
#pragma unroll
            for (int k = 0; k < kCand; k++) { const uint64_t c = COHERENT ? ald(&A->cand[k]) : A->cand[k]; if (c != 0 && (uint32_t)c == p.ifx) pos = k; }
        }
        if (pos >= 0) {
            uint64_t* dw = A->dir[pos];
            const uint64_t d0 = dw[0], d1 = dw[1];               // stale copies are lower bounds per direction
            for (int k = 0; k < 2; k++) {
                const uint64_t v = k ? p.dir1 : p.dir0;
                if (v == 0) continue;
                const bool known = (((d0 ^ v) & 0xffull) == 0 && d0 >= v) || (((d1 ^ v) & 0xffull) == 0 && d1 >= v);
                if (!known) topk_insert<2, 0xffull>(t, dw, v);
            }
        }
    }
    if (fl & ~x.flags) aor(&H->flags, fl);
}

// The first record of the flow in this epoch is stored whole (account.go:95): start, eth_protocol and the MACs here
// (plain values: only this record ever writes them in dedup mode), the identity dwords by k_finalize.
NF_DEV void dedup_publish_first(const TableView& t, uint32_t idx, const Rec& r, uint32_t seq32) {
    SlotHot* H = &t.hot[idx];
    SlotCold* C = &t.cold[idx];
    const uint32_t inv = ~seq32;
    ast(&H->start_inv, r.start());                               // raw start
    ast(&H->eth_tag, (uint64_t)r.eth());                         // raw eth_protocol
    ast(&H->smac_lo, tagged(inv, (uint32_t)r.smac()));
    ast(&C->smac_hi, tagged(inv, (uint32_t)(r.smac() >> 32)));
    ast(&H->dmac_lo, tagged(inv, (uint32_t)r.dmac()));
    ast(&C->dmac_hi, tagged(inv, (uint32_t)(r.dmac() >> 32)));
}

// ---- the two passes for ONE record, straight on the table (direct kernels; cache misses)
NF_DEV void dedup_claim_record(const TableView& t, const Rec& r, const uint64_t w[5], uint64_t h, uint32_t seq32) {
    Hints x;
    const uint64_t kx = sub_kx(t, r.d[21]), hs = sub_hash(t, h, r.d[21]);
    uint32_t idx = probe_home(t, w, hs, x, kx);
    if (idx == kNoSlot) {
        idx = find_or_claim(t, w, hs, nullptr, nullptr, nullptr, kx);
        if (idx == kNoSlot) return;
        x.id0 = 0;
    }
    dedup_claim(t, idx, x.id0, r.d[21], seq32);
}

NF_DEV void dedup_fold_record(const TableView& t, const Rec& r, const uint64_t w[5], uint64_t h, uint32_t seq32) {
    Hints x;
    const uint64_t kx = sub_kx(t, r.d[21]), hs = sub_hash(t, h, r.d[21]);
    uint32_t idx = probe_home(t, w, hs, x, kx);
    if (idx == kNoSlot) {
        idx = find_or_claim(t, w, hs, nullptr, nullptr, nullptr, kx);          // pass 1 claimed it: this only walks the probe sequence
        if (idx == kNoSlot) return;
        load_hints(&t.hot[idx], x);
    }
    if ((uint32_t)(x.id0 >> 32) == ~seq32) dedup_publish_first(t, idx, r, seq32);
    DedupPartial p;
    dedup_partial_from_record(r, seq32, p);
    dedup_merge(t, idx, x, p);
}

}  // namespace nfagg
