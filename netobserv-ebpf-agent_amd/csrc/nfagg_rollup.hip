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

template <typename M>
__global__ __launch_bounds__(256) void k_rollup(const M* __restrict__ partials, uint64_t n_flows, uint64_t n_cpu,
                                                nfagg_flow_metrics* __restrict__ base, M* __restrict__ folded) {
    const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_flows) return;
    nfagg_flow_metrics b = base[f];
    const M* p = partials + f * n_cpu;
    M acc = p[0];                                         // adopted whole (tracer.go / flow_content.go "== nil" arm)
    base_from(b, acc.start_mono_time_ts, acc.end_mono_time_ts, acc.eth_protocol);
    for (uint64_t c = 1; c < n_cpu; c++) {
        const M o = p[c];
        base_from(b, o.start_mono_time_ts, o.end_mono_time_ts, o.eth_protocol);
        fold(acc, o);
    }
    base[f] = b;
    folded[f] = acc;
}

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
