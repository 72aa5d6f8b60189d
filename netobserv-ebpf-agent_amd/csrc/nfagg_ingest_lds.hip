// nfagg_ingest_lds.hip — ingest variant 2: LDS fold per tile, then one global
// merge per distinct key of the tile (kept for A/B; the default is the persistent
// LDS flow cache of nfagg_ingest_cached.hip).
//
// Why: a Zipf(1.1) stream sends ~12 % of all records to one key and the
// adversarial config 90 %. Merging every record straight into HBM serialises
// on that slot's atomics. Instead each workgroup folds a tile of consecutive
// records in LDS with the exact sequential semantics of
// model.AccumulateBase (pkg/model/flow_content.go:28-61) — consecutive records
// of a tile are consecutive in arrival order, so "first"/"last" are decided by
// the local index — and only the tile's first record of each key goes to the
// table, carrying the folded partial (nfagg_device.h merge_partial).
#include "nfagg_device.h"

namespace nfagg {

template <int T>
struct TileLds {
    uint64_t key[5][T];        // SoA: lane-consecutive, conflict-free
    uint64_t htab[2 * T];      // (hash bits << 32) | (rep index + 1); 0 = empty
    uint64_t bytes[T];
    uint64_t end[T];
    uint64_t start_inv[T];
    uint64_t smac_tag[T];      // min of (local idx << 48) | mac48 over non-zero macs; ~0 = none
    uint64_t dmac_tag[T];
    uint64_t samp_tag[T];      // max of ((local idx+1) << 32) | sampling; 0 = none
    uint32_t packets[T];
    uint32_t flags[T];
    uint32_t eth_tag[T];       // max of ((local idx+1) << 16) | eth
    uint32_t dscp_tag[T];      // max of ((local idx+1) << 8) | dscp
    uint32_t first_idx[T];     // min local idx of the group
};

// One record per lane per tile (T == blockDim.x).
template <int T>
__global__ __launch_bounds__(T) void k_ingest_lds(TableView t, const void* __restrict__ recs, uint64_t n,
                                                  uint64_t seq_base) {
    __shared__ TileLds<T> L;
    const int li = threadIdx.x;
    const uint64_t n_tiles = (n + T - 1) / T;
    unsigned long long skipped = 0;

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t i = tile * T + li;
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
        // ---- phase 0: publish my key, reset my aggregate row and my share of the tile hash table
        if (valid) {
#pragma unroll
            for (int k = 0; k < 5; k++) L.key[k][li] = w[k];
        }
        L.htab[li] = 0; L.htab[li + T] = 0;
        L.bytes[li] = 0; L.end[li] = 0; L.start_inv[li] = 0;
        L.smac_tag[li] = ~0ull; L.dmac_tag[li] = ~0ull; L.samp_tag[li] = 0;
        L.packets[li] = 0; L.flags[li] = 0; L.eth_tag[li] = 0; L.dscp_tag[li] = 0;
        L.first_idx[li] = 0xffffffffu;
        __syncthreads();

        // ---- phase 1: find my group's representative in the tile, fold into its row
        int rep = li;
        if (valid) {
            const uint32_t hb = (uint32_t)(h >> 8);
            uint32_t e = (uint32_t)(h >> 44) & (2 * T - 1);
            const uint64_t mine = ((uint64_t)hb << 32) | (uint32_t)(li + 1);
            bool done = false;
            while (!done) {
                uint64_t cur = L.htab[e];
                if (cur == 0) {
                    cur = atomicCAS((unsigned long long*)&L.htab[e], 0ull, (unsigned long long)mine);
                    if (cur == 0) { rep = li; done = true; }
                }
                if (!done) {
                    bool same = (uint32_t)(cur >> 32) == hb;
                    if (same) {
                        const int c = (int)(uint32_t)cur - 1;
#pragma unroll
                        for (int k = 0; k < 5; k++) same &= (L.key[k][c] == w[k]);
                        if (same) { rep = c; done = true; }
                    }
                    if (!done) e = (e + 1) & (2 * T - 1);
                }
            }
            // model.AccumulateBase over the tile, order-resolved by local index
            if (r.bytes()) atomicAdd((unsigned long long*)&L.bytes[rep], (unsigned long long)r.bytes());
            if (r.packets()) atomicAdd(&L.packets[rep], r.packets());
            if (r.flags()) atomicOr(&L.flags[rep], r.flags());
            if (r.end()) atomicMax((unsigned long long*)&L.end[rep], (unsigned long long)r.end());
            if (r.start()) atomicMax((unsigned long long*)&L.start_inv[rep], (unsigned long long)~r.start());
            if (r.eth()) atomicMax(&L.eth_tag[rep], ((uint32_t)(li + 1) << 16) | r.eth());
            if (r.dscp()) atomicMax(&L.dscp_tag[rep], ((uint32_t)(li + 1) << 8) | r.dscp());
            if (r.sampling())
                atomicMax((unsigned long long*)&L.samp_tag[rep], ((unsigned long long)(li + 1) << 32) | r.sampling());
            if (r.smac()) atomicMin((unsigned long long*)&L.smac_tag[rep], ((unsigned long long)li << 48) | r.smac());
            if (r.dmac()) atomicMin((unsigned long long*)&L.dmac_tag[rep], ((unsigned long long)li << 48) | r.dmac());
            atomicMin(&L.first_idx[rep], (uint32_t)li);
        }
        __syncthreads();

        // ---- phase 2: the group's first record carries the folded partial to the table
        if (valid && L.first_idx[rep] == (uint32_t)li) {
            const uint64_t tile_seq = seq_base + tile * T;
            Partial p;
            p.bytes = L.bytes[rep]; p.end = L.end[rep]; p.start_inv = L.start_inv[rep];
            p.packets = L.packets[rep]; p.flags = L.flags[rep];
            const uint32_t et = L.eth_tag[rep], dt = L.dscp_tag[rep];
            const uint64_t st = L.samp_tag[rep];
            p.eth_tag = et ? ((tile_seq + (et >> 16)) << 16) | (et & 0xffffu) : 0ull;   // (seq+1)<<16 | eth
            p.dscp_tag = dt ? ((tile_seq + (dt >> 8)) << 8) | (dt & 0xffu) : 0ull;
            p.samp_tag = st ? ((tile_seq + (st >> 32)) << 32) | (st & 0xffffffffull) : 0ull;
            p.first_inv = ~(uint32_t)(tile_seq + li);
            const uint64_t sm = L.smac_tag[rep], dm = L.dmac_tag[rep];
            p.smac = (sm != ~0ull) ? (sm & 0xffffffffffffull) : 0ull;
            p.dmac = (dm != ~0ull) ? (dm & 0xffffffffffffull) : 0ull;
            p.smac_inv = (sm != ~0ull) ? ~(uint32_t)(tile_seq + (sm >> 48)) : 0u;
            p.dmac_inv = (dm != ~0ull) ? ~(uint32_t)(tile_seq + (dm >> 48)) : 0u;
#pragma unroll
            for (int k = 0; k < 15; k++) p.ident[k] = r.d[21 + k];
            upsert_partial(t, w, h, p);
        }
        __syncthreads();
    }
    if (skipped) aadd(&t.ctr->n_skipped, skipped);
}

hipError_t launch_ingest_lds(const TableView& t, const void* d_records, uint64_t n, uint64_t seq_base,
                             int variant, hipStream_t s) {
    (void)variant;
    constexpr int T = 256;
    uint64_t tiles = (n + T - 1) / T;
    uint64_t grid = tiles < 256ull * 8 ? tiles : 256ull * 8;
    (void)hipGetLastError(); hipLaunchKernelGGL(k_ingest_lds<T>, dim3((unsigned)grid), dim3(T), 0, s, t, d_records, n, seq_base);
    return hipGetLastError();
}

}  // namespace nfagg
