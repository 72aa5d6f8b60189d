// nfagg_ingest_cached.hip — the default ingest kernel: a persistent per-workgroup
// flow cache in LDS (variant 0; 3..5 are other geometries of the same kernel).
//
// A Zipf(1.1) stream over 1 M flows sends ~60-70 % of its records to the ~1000
// hottest flows. Each workgroup keeps those flows in an LDS-resident
// open-addressed cache for its whole lifetime: a record that hits the cache is
// folded with LDS atomics only, and the cache is merged into the HBM table once,
// when the workgroup has consumed its share of the batch. Records whose flow
// does not get a cache entry (probe window full) are merged into the HBM table
// directly, one record at a time (they are the cold tail: mostly one record per
// flow per workgroup anyway).
//
// Exactness (pkg/model/flow_content.go:28-61, pkg/flow/account.go:95): the
// commutative fields and the sequence-tagged "last non-zero" fields are folded
// in LDS with the same operators as in HBM, so a cache entry is just a partial
// (nfagg_device.h) — merging partials is associative and commutative. "First
// record" data never enters the cache: for every entry the earliest record the
// workgroup sees for that flow (LDS atomic min over sequence numbers; tiles are
// consumed in increasing sequence order) publishes its identity dwords to HBM
// itself, and so does the earliest record with a non-zero src/dst MAC.
#include <atomic>
#include "nfagg_device.h"

namespace nfagg {

template <int K>
struct FlowCache {
    uint64_t h64[K];          // 0 = free, else the key hash (| 1 so it is never 0)
    uint64_t key[5][K];
    uint64_t bytes[K];
    uint64_t end[K];
    uint64_t start_inv[K];
    uint64_t eth_tag[K];
    uint64_t dscp_tag[K];
    uint64_t samp_tag[K];
    uint32_t packets[K];
    uint32_t flags[K];
    uint32_t first_seq[K];    // min seq32 over the records that hit the entry
    uint32_t smac_seq[K];     // min seq32 over those with a non-zero src_mac
    uint32_t dmac_seq[K];
    uint32_t gslot[K];        // slot of the flow in the HBM table (set by the first record)
};

constexpr int kCacheProbe = 8;   // probe window: a flow that finds no entry within it bypasses the cache

// Publish "first record" / MAC words (used by the records that are the workgroup's earliest for their flow).
// Only the first record's sequence number goes into the slot; its identity dwords follow in k_finalize.
NF_DEV void merge_ident(const TableView& t, uint32_t idx, uint32_t inv, const Rec& r, const Hints& x) {
    const uint64_t my0 = tagged(inv, r.d[21]);
    if (x.id0 <= my0) amax(&t.hot[idx].id0, my0);
}
NF_DEV void merge_smac(const TableView& t, uint32_t idx, uint32_t inv, uint64_t mac, const Hints& x) {
    const uint64_t lo = tagged(inv, (uint32_t)mac);
    if (x.smac_lo <= lo) { amax(&t.hot[idx].smac_lo, lo); amax(&t.cold[idx].smac_hi, tagged(inv, (uint32_t)(mac >> 32))); }
}
NF_DEV void merge_dmac(const TableView& t, uint32_t idx, uint32_t inv, uint64_t mac, const Hints& x) {
    const uint64_t lo = tagged(inv, (uint32_t)mac);
    if (x.dmac_lo <= lo) { amax(&t.hot[idx].dmac_lo, lo); amax(&t.cold[idx].dmac_hi, tagged(inv, (uint32_t)(mac >> 32))); }
}

template <int BLOCK, int K, bool SKETCH, bool TIMING = false>
__global__ __launch_bounds__(BLOCK) void k_ingest_cached(TableView t, SketchView sk, const void* __restrict__ recs, uint64_t n,
                                                         uint64_t seq_base) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FlowCache<K>& L = *reinterpret_cast<FlowCache<K>*>(lds_raw);
    const int tid = threadIdx.x;
    for (int e = tid; e < K; e += BLOCK) {
        L.h64[e] = 0; L.bytes[e] = 0; L.end[e] = 0; L.start_inv[e] = 0; L.eth_tag[e] = 0; L.dscp_tag[e] = 0;
        L.samp_tag[e] = 0; L.packets[e] = 0; L.flags[e] = 0;
        L.first_seq[e] = 0xffffffffu; L.smac_seq[e] = 0xffffffffu; L.dmac_seq[e] = 0xffffffffu; L.gslot[e] = kNoSlot;
    }
    __syncthreads();

    const uint64_t n_tiles = (n + BLOCK - 1) / BLOCK;
    unsigned long long skipped = 0, bypassed = 0;
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, tp = 0;   // TIMING: load+hash, A, barrier, B, barrier, C, flush
#define NF_TICK(k) do { if (TIMING) { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[k] += tn_ - tp; tp = tn_; } } while (0)
    if (TIMING) tp = __builtin_readcyclecounter();
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t i = tile * BLOCK + tid;
        bool valid = i < n;
        Rec r;
        uint64_t w[5];
        uint64_t h = 0;
        if (valid) {
            load_record(recs, i, r);
            r.canonicalize();
            r.key_words(w);
            h = key_hash(w);
            if (t.n_shards > 1 && shard_of_hash(h, t.n_shards) != t.shard_id) { valid = false; skipped++; }
        }
        const uint32_t seq32 = (uint32_t)(seq_base + i);
        if (TIMING) { asm volatile("" :: "v"(h)); NF_TICK(0); }
        // ---- phase A: find or claim a cache entry by the 64-bit hash
        int ent = -1;
        bool creator = false;
        if (valid) {
            const uint64_t hk = h | 1ull;
            uint32_t e = (uint32_t)(h >> 40) & (K - 1);
#pragma unroll 1
            for (int p = 0; p < kCacheProbe; p++) {
                uint64_t cur = L.h64[e];
                if (cur == 0) {
                    cur = atomicCAS((unsigned long long*)&L.h64[e], 0ull, (unsigned long long)hk);
                    if (cur == 0) { ent = (int)e; creator = true; break; }
                }
                if (cur == hk) { ent = (int)e; break; }
                e = (e + 1) & (K - 1);
            }
            if (creator) {
#pragma unroll
                for (int k = 0; k < 5; k++) L.key[k][ent] = w[k];
            }
        }
        NF_TICK(1);
        __syncthreads();
        NF_TICK(2);
        // ---- phase B: verify the full key (two flows may share a 64-bit hash), fold into the entry
        if (valid && ent >= 0) {
            bool same = true;
#pragma unroll
            for (int k = 0; k < 5; k++) same &= (L.key[k][ent] == w[k]);
            if (!same) ent = -1;   // hash collision inside the cache: treat as a bypass record
        }
        if (valid && ent >= 0) {
            if (r.bytes()) atomicAdd((unsigned long long*)&L.bytes[ent], (unsigned long long)r.bytes());
            if (r.packets()) atomicAdd(&L.packets[ent], r.packets());
            if (r.flags() & ~L.flags[ent]) atomicOr(&L.flags[ent], r.flags());
            if (r.end() > L.end[ent]) atomicMax((unsigned long long*)&L.end[ent], (unsigned long long)r.end());
            if (r.start() && ~r.start() > L.start_inv[ent])
                atomicMax((unsigned long long*)&L.start_inv[ent], (unsigned long long)~r.start());
            const uint64_t s1 = (uint64_t)seq32 + 1;
            if (r.eth()) atomicMax((unsigned long long*)&L.eth_tag[ent], (unsigned long long)((s1 << 16) | r.eth()));
            if (r.dscp()) atomicMax((unsigned long long*)&L.dscp_tag[ent], (unsigned long long)((s1 << 8) | r.dscp()));
            if (r.sampling()) atomicMax((unsigned long long*)&L.samp_tag[ent], (unsigned long long)((s1 << 32) | r.sampling()));
            // earliest-record trackers: only records of the tile that first touches the entry can lower them
            if (L.first_seq[ent] > seq32) atomicMin(&L.first_seq[ent], seq32);
            if (r.smac() && L.smac_seq[ent] > seq32) atomicMin(&L.smac_seq[ent], seq32);
            if (r.dmac() && L.dmac_seq[ent] > seq32) atomicMin(&L.dmac_seq[ent], seq32);
        }
        NF_TICK(3);
        __syncthreads();
        NF_TICK(4);
        // ---- phase C: HBM work of this tile
        if (valid) {
            if (ent < 0) {
                // bypass: this flow has no cache entry, merge the record itself
                bypassed++;
                Partial p;
                partial_from_record(r, seq_base + i, p);
                upsert_partial(t, w, h, p);
                if (SKETCH) sketch_add(sk, w, r.bytes());
            } else {
                const bool is_first = L.first_seq[ent] == seq32;
                const bool is_smac = r.smac() && L.smac_seq[ent] == seq32;
                const bool is_dmac = r.dmac() && L.dmac_seq[ent] == seq32;
                if (is_first || is_smac || is_dmac) {
                    // the workgroup's earliest record of this flow (or earliest with a MAC): publish its own words
                    Hints x;
                    uint32_t idx = probe_home(t, w, h, x);
                    if (idx == kNoSlot) {
                        idx = find_or_claim(t, w, h);
                        if (idx != kNoSlot) load_hints(&t.hot[idx], x);
                    }
                    if (idx != kNoSlot) {
                        if (is_first) { L.gslot[ent] = idx; merge_ident(t, idx, ~seq32, r, x); }
                        if (is_smac) merge_smac(t, idx, ~seq32, r.smac(), x);
                        if (is_dmac) merge_dmac(t, idx, ~seq32, r.dmac(), x);
                    }
                }
            }
        }
        if (TIMING) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); NF_TICK(5); }
        // the next tile's phase A only reads/claims h64 and writes keys of NEW entries: no barrier needed here,
        // phase B of the next tile is separated from this phase C's LDS reads by the barrier after phase A
    }
    __syncthreads();
    // ---- final flush: every cache entry is one partial for the HBM table
    for (int e = tid; e < K; e += BLOCK) {
        if (L.h64[e] == 0) continue;
        const uint32_t idx = L.gslot[e];
        if (idx == kNoSlot) continue;   // cannot happen: the entry's first record set it
        SlotHot* H = &t.hot[idx];
        Hints x;
        load_hints(H, x);
        if (L.bytes[e]) aadd(&H->bytes, L.bytes[e]);
        if (L.packets[e]) aadd(&H->packets, L.packets[e]);
        if (L.flags[e] & ~x.flags) aor(&H->flags, L.flags[e]);
        if (L.end[e] > x.end) amax(&H->end, L.end[e]);
        if (L.start_inv[e] > x.start_inv) amax(&H->start_inv, L.start_inv[e]);
        if (L.eth_tag[e]) amax(&H->eth_tag, L.eth_tag[e]);
        if (L.dscp_tag[e]) amax(&H->dscp_tag, L.dscp_tag[e]);
        if (L.samp_tag[e]) amax(&H->samp_tag, L.samp_tag[e]);
        if (SKETCH) {   // the entry's records reach the sketches as one (IPs, byte sum) contribution
            uint64_t w[5];
#pragma unroll
            for (int k = 0; k < 5; k++) w[k] = L.key[k][e];
            sketch_add(sk, w, L.bytes[e]);
        }
    }
    if (TIMING) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NF_TICK(6);
        if ((tid & 63) == 0) for (int k = 0; k < 7; k++) aadd(&t.ctr->phase[k], ph[k]);
    }
#undef NF_TICK
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
    if (bypassed) aadd(&t.ctr->n_bypassed, bypassed);
}

template <int BLOCK, int K, bool SKETCH, bool TIMING = false>
static hipError_t run_cached(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, uint64_t seq_base,
                             int blocks_per_cu, hipStream_t s) {
    const size_t lds = sizeof(FlowCache<K>);
    static std::atomic<bool> attr_set_dev[64];   // per device: a process may drive several GPUs, from several host threads
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    std::atomic<bool>& attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ingest_cached<BLOCK, K, SKETCH, TIMING>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    const uint64_t tiles = (n + BLOCK - 1) / BLOCK;
    uint64_t grid = 256ull * blocks_per_cu;
    if (grid > tiles) grid = tiles;
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_ingest_cached<BLOCK, K, SKETCH, TIMING>), dim3((unsigned)grid), dim3(BLOCK), lds, s, t, sk, d_records, n, seq_base);
    return hipGetLastError();
}

// sk.flags != 0: the sketch updates are fused into the kernel (no separate k_sketch_update launch).
hipError_t launch_ingest_cached(const TableView& t, const SketchView& sk, const void* d_records, uint64_t n, uint64_t seq_base,
                                int variant, hipStream_t s) {
    const bool f = sk.flags != 0;
    switch (variant) {
        case 3: return f ? run_cached<512, 512, true>(t, sk, d_records, n, seq_base, 2, s)       // 2 WG/CU x 64 KB
                         : run_cached<512, 512, false>(t, sk, d_records, n, seq_base, 2, s);
        case 4: return f ? run_cached<256, 256, true>(t, sk, d_records, n, seq_base, 4, s)       // 4 WG/CU x 32 KB
                         : run_cached<256, 256, false>(t, sk, d_records, n, seq_base, 4, s);
        case 5: return f ? run_cached<1024, 512, true>(t, sk, d_records, n, seq_base, 2, s)
                         : run_cached<1024, 512, false>(t, sk, d_records, n, seq_base, 2, s);
#ifdef NFAGG_DIAG
        case 6: return run_cached<1024, 1024, false, true>(t, sk, d_records, n, seq_base, 1, s);    // diagnostics: phase timing
#endif
        default: return f ? run_cached<1024, 1024, true>(t, sk, d_records, n, seq_base, 1, s)    // 7 (and 0 for small batches): 1 WG/CU x 120 KB
                          : run_cached<1024, 1024, false>(t, sk, d_records, n, seq_base, 1, s);
    }
}

}  // namespace nfagg
