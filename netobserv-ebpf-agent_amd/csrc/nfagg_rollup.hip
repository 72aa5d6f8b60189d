// nfagg_rollup.hip — per-CPU partial rollup for the kernel-map eviction branch.
// Replaces the accumulator closures of FlowFetcher.LookupAndDeleteMap
// (pkg/tracer/tracer.go:1057-1110) as driven by lookupAndDeletePerCPUMap
// (:1118-1146): for each flow, element 0 of the per-CPU slice is adopted whole,
// elements 1..n_cpu-1 are folded into it in CPU order with the matching
// model.Accumulate* (pkg/model/flow_content.go), and every element feeds
// buildBaseFromAdditional (flow_content.go:63-74). One lane per flow; the
// per-flow fold is sequential by definition, the flows are independent.
#include <hip/hip_runtime.h>
#include "../../include/nfagg.h"
#include "nfagg_internal.h"
#include "nfagg_hash.h"

namespace nfagg {

#define RD __device__ __forceinline__

// flow_content.go:63-74
RD void base_from(nfagg_flow_metrics& b, uint64_t start, uint64_t end, uint16_t eth) {
    if (b.start_mono_time_ts == 0 || (b.start_mono_time_ts > start && start != 0)) b.start_mono_time_ts = start;
    if (b.end_mono_time_ts == 0 || b.end_mono_time_ts < end) b.end_mono_time_ts = end;
    if (b.eth_protocol == 0) b.eth_protocol = eth;
}

RD uint16_t sat_add16(uint16_t a, uint16_t b) {   // flow_content.go:209-215
    uint16_t s = (uint16_t)(a + b);
    return s < a ? (uint16_t)0xFFFF : s;
}

// flow_content.go:154-177
RD void fold(nfagg_additional_metrics& p, const nfagg_additional_metrics& o) {
    if (p.flow_rtt < o.flow_rtt) p.flow_rtt = o.flow_rtt;
    if (p.ipsec_encrypted_ret < o.ipsec_encrypted_ret) { p.ipsec_encrypted = o.ipsec_encrypted; p.ipsec_encrypted_ret = o.ipsec_encrypted_ret; }
    if (p.ipsec_encrypted_ret == o.ipsec_encrypted_ret) { if (o.ipsec_encrypted) p.ipsec_encrypted = o.ipsec_encrypted; }
}
// flow_content.go:76-96 — name[] and the struct's own start/end stay those of CPU 0
RD void fold(nfagg_dns_metrics& p, const nfagg_dns_metrics& o) {
    p.flags |= o.flags;
    if (o.id != 0) p.id = o.id;
    p.errno_ = o.errno_;
    if (p.latency < o.latency) p.latency = o.latency;
}
// flow_content.go:98-118
RD void fold(nfagg_pkt_drop_metrics& p, const nfagg_pkt_drop_metrics& o) {
    p.bytes = sat_add16(p.bytes, o.bytes);
    p.packets = sat_add16(p.packets, o.packets);
    p.latest_flags |= o.latest_flags;
    if (o.latest_drop_cause != 0) p.latest_drop_cause = o.latest_drop_cause;
    if (o.latest_state != 0) p.latest_state = o.latest_state;
}
RD uint64_t md8(const uint8_t* m) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) v |= (uint64_t)m[i] << (8 * i);
    return v;
}
// flow_content.go:120-137 (+ record.go:189-196 networkEventsMDExist)
RD void fold(nfagg_network_events_metrics& p, const nfagg_network_events_metrics& o) {
    for (int i = 0; i < 4; i++) {
        if (o.packets[i] == 0) continue;
        const uint64_t md = md8(o.network_events[i]);
        bool exists = false;
        for (int k = 0; k < 4; k++) exists |= (md8(p.network_events[k]) == md);
        if (exists) continue;
        const uint8_t idx = p.network_events_idx;
        if (idx >= 4) return;   // Go would panic on the index; never produced by the kernel side
        p.bytes[idx] = sat_add16(p.bytes[idx], o.bytes[i]);
        p.packets[idx] = sat_add16(p.packets[idx], o.packets[i]);
        for (int b = 0; b < 8; b++) p.network_events[idx][b] = o.network_events[i][b];
        p.network_events_idx = (uint8_t)((idx + 1) % 4);
    }
}
RD bool ip_all_zero(const uint8_t* ip) {   // record.go:233-238: 0.0.0.0 (v4-mapped) or ::
    bool lead = true;
    for (int i = 0; i < 10; i++) lead &= (ip[i] == 0);
    bool tail = true;
    for (int i = 12; i < 16; i++) tail &= (ip[i] == 0);
    const bool v4 = ip[10] == 0xff && ip[11] == 0xff, v6 = ip[10] == 0 && ip[11] == 0;
    return lead && tail && (v4 || v6);
}
// flow_content.go:139-152
RD void fold(nfagg_xlat_metrics& p, const nfagg_xlat_metrics& o) {
    if (!ip_all_zero(o.saddr) && !ip_all_zero(o.daddr)) p = o;
}
// flow_content.go:179-198
RD void fold(nfagg_quic_metrics& p, const nfagg_quic_metrics& o) {
    if (p.version < o.version) p.version = o.version;
    if (p.seen_long_hdr < o.seen_long_hdr) p.seen_long_hdr = o.seen_long_hdr;
    if (p.seen_short_hdr < o.seen_short_hdr) p.seen_short_hdr = o.seen_short_hdr;
}

// CPU 0 adopted whole, CPUs 1.. folded in order (tracer.go:1057-1110 closures), buildBaseFromAdditional after every
// partial. The partials of one flow are read four at a time: the loads of a group are independent and in flight together,
// only the folds are sequential (a lane-strided stream is latency-bound, not bandwidth-bound).
template <typename M>
RD void fold_partials(const M* __restrict__ p, uint32_t n_cpu, nfagg_flow_metrics& b, M& acc) {
    acc = p[0];
    base_from(b, acc.start_mono_time_ts, acc.end_mono_time_ts, acc.eth_protocol);
    uint32_t c = 1;
    for (; c + 4 <= n_cpu; c += 4) {
        const M o0 = p[c], o1 = p[c + 1], o2 = p[c + 2], o3 = p[c + 3];
        base_from(b, o0.start_mono_time_ts, o0.end_mono_time_ts, o0.eth_protocol); fold(acc, o0);
        base_from(b, o1.start_mono_time_ts, o1.end_mono_time_ts, o1.eth_protocol); fold(acc, o1);
        base_from(b, o2.start_mono_time_ts, o2.end_mono_time_ts, o2.eth_protocol); fold(acc, o2);
        base_from(b, o3.start_mono_time_ts, o3.end_mono_time_ts, o3.eth_protocol); fold(acc, o3);
    }
    for (; c < n_cpu; c++) {
        const M o = p[c];
        base_from(b, o.start_mono_time_ts, o.end_mono_time_ts, o.eth_protocol);
        fold(acc, o);
    }
}

template <typename M>
__global__ __launch_bounds__(256) void k_rollup(const M* __restrict__ partials, uint64_t n_flows, uint64_t n_cpu,
                                                nfagg_flow_metrics* __restrict__ base, M* __restrict__ folded) {
    const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_flows) return;
    nfagg_flow_metrics b = base[f];
    M acc;
    fold_partials(partials + f * n_cpu, (uint32_t)n_cpu, b, acc);
    base[f] = b;
    folded[f] = acc;
}

// ---------------------------------------------------------------------------------------------
// Merge of the drained eBPF maps — FlowFetcher.LookupAndDeleteMap (pkg/tracer/tracer.go:1022-1116)
// with lookupAndDeletePerCPUMap (:1118-1146) as ONE device-side join: the main map (aggregated_flows)
// and up to six per-CPU feature maps, each a list of (40-byte id, value | n_cpu partials), become one
// dense array of merged flows {id, base metrics, present bits, folded parts}.
//
// The join is a hash build over every input row. A slot never stores a key: it stores `rep`, the
// global position of a row that carries the key (claimed with one CAS); a later row compares its key
// with the key of row `rep` in the INPUT arrays, which nobody writes — so there is no publish
// protocol and no spinning. Per slot: `first` = smallest global position of the key (atomic min; it
// orders the output by first appearance, main map first, then the maps in the order Go walks them)
// and `row[m]` = the key's row in map m (atomic min: an id listed twice in one map keeps its first
// row, as the second LookupAndDelete of the Go loop fails and is skipped, :1048-1052,1130-1134).
// Then: flag rows with first == position, scan, and one lane per merged flow replays the Go order
// (dns, drops, network events, xlat, additional, quic; CPUs ascending) over its rows.
struct MergeSlot { uint32_t rep, first, row[7]; uint32_t pad_; };   // 40 bytes, initialised to 0xFF

RD void load_key(const uint8_t* ids, uint32_t row, uint64_t w[5]) {
    const uint64_t* p = reinterpret_cast<const uint64_t*>(ids + (size_t)row * 40);
#pragma unroll
    for (int k = 0; k < 5; k++) w[k] = p[k];
    w[4] &= 0x00FFFFFFFFFFFFFFull;          // byte 39 is a blank field in Go (bpf_x86_bpfel.go:119): not part of the key
}

RD void locate(const MergeIn& in, uint32_t g, int& q, uint32_t& row) {
    q = 0;
#pragma unroll
    for (int k = 1; k < 7; k++) q += (g >= in.off[k]) ? 1 : 0;
    row = g - in.off[q];
}

__global__ __launch_bounds__(256) void k_merge_build(MergeIn in, MergeSlot* __restrict__ slots, uint32_t mask,
                                                     uint32_t* __restrict__ slot_of, unsigned int* __restrict__ n_dup) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= in.off[7]) return;
    int q; uint32_t row;
    locate(in, g, q, row);
    uint64_t w[5];
    load_key(in.ids[q], row, w);
    uint32_t s = (uint32_t)key_hash(w) & mask;
    for (;;) {
        uint32_t rep = __hip_atomic_load(&slots[s].rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rep == 0xFFFFFFFFu) {
            uint32_t expected = 0xFFFFFFFFu;
            if (__hip_atomic_compare_exchange_strong(&slots[s].rep, &expected, g, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) rep = g;
            else rep = expected;
        }
        bool same = rep == g;
        if (!same) {
            int rq; uint32_t rrow;
            locate(in, rep, rq, rrow);
            uint64_t v[5];
            load_key(in.ids[rq], rrow, v);
            same = v[0] == w[0] && v[1] == w[1] && v[2] == w[2] && v[3] == w[3] && v[4] == w[4];
        }
        if (same) break;
        s = (s + 1) & mask;
    }
    __hip_atomic_fetch_min(&slots[s].first, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t old = __hip_atomic_fetch_min(&slots[s].row[q], row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old != 0xFFFFFFFFu) atomicAdd(n_dup, 1u);
    slot_of[g] = s;
}

// flags + block-local exclusive scan (block = 1024 rows); block sums go through launch_scan_block_sums
__global__ __launch_bounds__(1024) void k_merge_flags(uint32_t n_rows, const MergeSlot* __restrict__ slots, const uint32_t* __restrict__ slot_of,
                                                      uint32_t* __restrict__ local_off, uint32_t* __restrict__ block_sum) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t g = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t flag = (g < n_rows && slots[slot_of[g]].first == g) ? 1u : 0u;
    uint32_t v = flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < wave; k++) base += wave_tot[k];
    if (g < n_rows) local_off[g] = flag ? base + v - 1 : 0xFFFFFFFFu;
    if (threadIdx.x == 1023) block_sum[blockIdx.x] = base + v;
}

template <typename M>
RD bool merge_part(const MergeIn& in, int q, uint32_t row, nfagg_flow_metrics& b, M* __restrict__ out, uint64_t j) {
    M acc;
    const bool have = row != 0xFFFFFFFFu;
    if (have) {
        fold_partials(reinterpret_cast<const M*>(in.vals[q]) + (size_t)row * in.n_cpu, in.n_cpu, b, acc);
    } else {
        memset(&acc, 0, sizeof acc);
    }
    if (out) out[j] = acc;
    return have;
}

__global__ __launch_bounds__(256) void k_merge_fold(MergeIn in, MergeOut out, const MergeSlot* __restrict__ slots,
                                                    const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ local_off,
                                                    const uint64_t* __restrict__ block_base) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= in.off[7]) return;
    const uint32_t lo = local_off[g];
    if (lo == 0xFFFFFFFFu) return;
    const uint64_t j = block_base[g / 1024] + lo;
    const MergeSlot sl = slots[slot_of[g]];
    int q; uint32_t row;
    locate(in, g, q, row);
    nfagg_flow_record rec;
    memset(&rec, 0, sizeof rec);
    {
        uint64_t w[5];
        load_key(in.ids[q], row, w);
        memcpy(&rec.id, w, 40);
    }
    // position 0 of the walk order is the main map: flows[id] = NewBpfFlowContent(baseMetrics), else zero metrics (:1136-1139)
    if (sl.row[0] != 0xFFFFFFFFu) rec.metrics = reinterpret_cast<const nfagg_flow_metrics*>(in.vals[0])[sl.row[0]];
    uint32_t present = 0;
    // walk order positions 1..6 = dns, drops, network events, xlat, additional, quic (tracer.go:1057-1110)
    if (merge_part<nfagg_dns_metrics>(in, 1, sl.row[1], rec.metrics, out.dns, j)) present |= NFAGG_FEAT_DNS;
    if (merge_part<nfagg_pkt_drop_metrics>(in, 2, sl.row[2], rec.metrics, out.drops, j)) present |= NFAGG_FEAT_DROPS;
    if (merge_part<nfagg_network_events_metrics>(in, 3, sl.row[3], rec.metrics, out.network_events, j)) present |= NFAGG_FEAT_NETWORK_EVENTS;
    if (merge_part<nfagg_xlat_metrics>(in, 4, sl.row[4], rec.metrics, out.xlat, j)) present |= NFAGG_FEAT_XLAT;
    if (merge_part<nfagg_additional_metrics>(in, 5, sl.row[5], rec.metrics, out.additional, j)) present |= NFAGG_FEAT_ADDITIONAL;
    if (merge_part<nfagg_quic_metrics>(in, 6, sl.row[6], rec.metrics, out.quic, j)) present |= NFAGG_FEAT_QUIC;
    out.records[j] = rec;
    out.present[j] = (uint8_t)present;
}

hipError_t launch_merge_build(const MergeIn& in, void* d_slots, uint32_t n_slots, uint32_t* d_slot_of, unsigned int* d_n_dup,
                              uint32_t* d_local_off, uint32_t* d_block_sum, hipStream_t s) {
    const uint32_t n = in.off[7];
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_merge_build, dim3((n + 255) / 256), dim3(256), 0, s, in, (MergeSlot*)d_slots, n_slots - 1, d_slot_of, d_n_dup);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_merge_flags, dim3((n + 1023) / 1024), dim3(1024), 0, s, n, (const MergeSlot*)d_slots, d_slot_of, d_local_off, d_block_sum);
    return hipGetLastError();
}

hipError_t launch_merge_fold(const MergeIn& in, const MergeOut& out, const void* d_slots, const uint32_t* d_slot_of,
                             const uint32_t* d_local_off, const uint64_t* d_block_base, hipStream_t s) {
    const uint32_t n = in.off[7];
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_merge_fold, dim3((n + 255) / 256), dim3(256), 0, s, in, out, (const MergeSlot*)d_slots, d_slot_of, d_local_off, d_block_base);
    return hipGetLastError();
}

size_t merge_slot_bytes() { return sizeof(MergeSlot); }

size_t rollup_struct_size(int kind) {
    switch (kind) {
        case 0: return sizeof(nfagg_additional_metrics);
        case 1: return sizeof(nfagg_dns_metrics);
        case 2: return sizeof(nfagg_pkt_drop_metrics);
        case 3: return sizeof(nfagg_network_events_metrics);
        case 4: return sizeof(nfagg_xlat_metrics);
        default: return sizeof(nfagg_quic_metrics);
    }
}

template <typename M>
static hipError_t run(const void* p, uint64_t nf, uint64_t nc, void* base, void* folded, hipStream_t s) {
    (void)hipGetLastError(); hipLaunchKernelGGL(k_rollup<M>, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s,
                       (const M*)p, nf, nc, (nfagg_flow_metrics*)base, (M*)folded);
    return hipGetLastError();
}

hipError_t launch_rollup(int kind, const void* d_partials, uint64_t n_flows, uint64_t n_cpu,
                         void* d_base, void* d_folded, hipStream_t s) {
    switch (kind) {
        case 0: return run<nfagg_additional_metrics>(d_partials, n_flows, n_cpu, d_base, d_folded, s);
        case 1: return run<nfagg_dns_metrics>(d_partials, n_flows, n_cpu, d_base, d_folded, s);
        case 2: return run<nfagg_pkt_drop_metrics>(d_partials, n_flows, n_cpu, d_base, d_folded, s);
        case 3: return run<nfagg_network_events_metrics>(d_partials, n_flows, n_cpu, d_base, d_folded, s);
        case 4: return run<nfagg_xlat_metrics>(d_partials, n_flows, n_cpu, d_base, d_folded, s);
        case 5: return run<nfagg_quic_metrics>(d_partials, n_flows, n_cpu, d_base, d_folded, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace nfagg
