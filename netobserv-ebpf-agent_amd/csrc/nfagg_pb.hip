// nfagg_pb.hip — evicted flow_record_t -> serialized pbflow.Record, in bulk on the GPU
// (SURVEY.md §8(f) rank 1). Replaces, for the records Accounter.evict produces,
//   pkg/model/record.go:82-125      NewRecord (flow start/end wall-clock times, interface list)
//   pkg/pbflow/proto.go:18-149      FlowsToPB / FlowToPB
//   proto.Marshal of pbflow.Record  (pkg/exporter/kafka_proto.go:53, grpc marshalling of pbflow.Records)
//   pkg/exporter/kafka_proto.go:37-47 getFlowKey
// Output: for record i the frame `0x0A varint(len) body` at frame_offsets[i]; any run of frames
// [a, b) is a serialized pbflow.Records{entries[a..b)} (proto/flow.proto:16-18, what GRPCProto sends,
// split at will for GRPC_MESSAGE_MAX_FLOWS); the body alone is the Kafka message value.
// Wire rules as google.golang.org/protobuf applies them: fields in field-number order, proto3
// scalars omitted at zero, message fields emitted whenever FlowToPB sets the pointer (DataLink,
// Network, Transport, both Timestamps, agent_ip, time_flow_rtt = durationpb.New(0) -> empty message),
// oneof members emitted even at their zero value (IP.ipv4 = 0).
//
// With a PbFeat the same kernels encode the MapTracer branch (pkg/flow/tracer_map.go:103-146): the
// full model.BpfFlowContent — DNS, packet drops, xlat, RTT/IPsec, QUIC (proto.go:79-118,129-138;
// record.go:116-125) — with a nil SampleDecoder (record.go:126): network events are not decoded.
//
// Byte-granular, HBM-bound streaming work: 144 B read + ~110 B written per record (more with
// features). Three kernels: sizes + block-local scan, scan of the block sums, encode. A wave
// encodes its 64 records into LDS at their final relative byte positions (consecutive records are
// contiguous in the output), one 16 KiB window of the output at a time, and copies each window out
// with aligned 16-byte stores.
#include "nfagg_device.h"
#include "nfagg_pb.h"

namespace nfagg {

// ---- byte sinks: one counts, one writes (LDS or global bytes)
struct CountSink {
    uint32_t n = 0;
    NF_DEV void put(uint8_t) { n++; }
};
// Writes only the bytes whose position (relative to the wave's LDS image) falls into [lo, lo + len):
// a frame that straddles two windows is encoded once per window.
struct WindowSink {
    uint8_t* lds;      // window base
    uint32_t pos;      // position of the next byte in the wave image
    uint32_t lo, len;
    NF_DEV void put(uint8_t b) { const uint32_t k = pos - lo; if (k < len) lds[k] = b; pos++; }
};

NF_DEV uint32_t varint_len(uint64_t v) {
    uint32_t n = 1;
    while (v >= 0x80) { v >>= 7; n++; }
    return n;
}
template <typename S> NF_DEV void put_varint(S& s, uint64_t v) {
    while (v >= 0x80) { s.put((uint8_t)(v | 0x80)); v >>= 7; }
    s.put((uint8_t)v);
}
template <typename S> NF_DEV void put_tag(S& s, uint32_t field, uint32_t wt) { put_varint(s, ((uint64_t)field << 3) | wt); }
template <typename S> NF_DEV void put_uint(S& s, uint32_t field, uint64_t v) { if (v) { put_tag(s, field, 0); put_varint(s, v); } }
NF_DEV uint32_t uint_len(uint32_t field, uint64_t v) { return v ? varint_len((uint64_t)field << 3) + varint_len(v) : 0; }

// message IP { oneof { fixed32 ipv4 = 1; bytes ipv6 = 2; } } as a sub-message of `field`
// The address travels as four little-endian dwords in registers, never as a byte pointer: a byte loop over global
// memory is one exposed load latency per byte.
struct Ip4w { uint32_t w[4]; };
NF_DEV uint8_t ip_byte(const Ip4w& a, int k) { return (uint8_t)(a.w[k >> 2] >> (8 * (k & 3))); }
template <typename S> NF_DEV void put_ip(S& s, uint32_t field, const Ip4w& ip, bool v6) {
    put_tag(s, field, 2);
    if (v6) {
        s.put(18); s.put(0x12); s.put(16);
#pragma unroll
        for (int k = 0; k < 16; k++) s.put(ip_byte(ip, k));
    } else {   // model.IntEncodeV4 (record.go:202-204): big-endian value of the last four bytes; fixed32 is little-endian on the wire
        s.put(5); s.put(0x0D);
        s.put(ip_byte(ip, 15)); s.put(ip_byte(ip, 14)); s.put(ip_byte(ip, 13)); s.put(ip_byte(ip, 12));
    }
}

// google.protobuf.Timestamp of currentTime.Add(-Duration(mono_now - ts)) (record.go:90-97, proto.go:61-68)
struct TimeParts { int64_t sec, nsec; };
NF_DEV TimeParts flow_time(const PbParams& P, uint64_t ts) {
    const int64_t delta = (int64_t)(P.mono_now - ts);
    const int64_t d = (int64_t)(0ull - (uint64_t)delta);
    int64_t dsec = d / 1000000000ll, nsec = P.now_nsec + d % 1000000000ll;   // time.Time.Add
    if (nsec >= 1000000000ll) { dsec++; nsec -= 1000000000ll; } else if (nsec < 0) { dsec--; nsec += 1000000000ll; }
    return TimeParts{P.now_sec + dsec, nsec};
}
template <typename S> NF_DEV void put_time(S& s, uint32_t field, const TimeParts& t) {
    put_tag(s, field, 2);
    s.put((uint8_t)(uint_len(1, (uint64_t)t.sec) + uint_len(2, (uint64_t)t.nsec)));
    put_uint(s, 1, (uint64_t)t.sec);      // int64: a negative value takes ten bytes
    put_uint(s, 2, (uint64_t)t.nsec);
}

// interfaceNamer(ifIndex, mac) + udnsCache lookup, as a table: exact (index, MAC) row first,
// then the first row of that index that matches any MAC; no row -> the "unknown" name, no UDN. The host hands the
// table over STABLY SORTED by if_index (rows of one index keep their order, so the answer is that of a scan in table
// order): binary search for the first row of the index, then only that index's rows. `tab` is a flat pointer: the
// kernels stage the table in LDS when it fits (kNamesLdsRows rows), so a lookup costs LDS latencies, not HBM ones.
// Row layout (nfagg_intf_name, 92 bytes): if_index@0 mac@4 has_mac@10 name_len@11 name@12 udn_len@28 udn@29.
constexpr uint32_t kNameRowBytes = sizeof(nfagg_intf_name);
constexpr uint32_t kNamesLdsRows = 96;
static_assert(kNameRowBytes == 92 && kNameRowBytes % 4 == 0, "nfagg_intf_name layout");
NF_DEV const uint8_t* lookup_name(const uint8_t* tab, uint32_t n_names, uint32_t if_index, uint64_t mac48) {
    uint32_t lo = 0, hi = n_names;
    while (lo < hi) {                                   // first row with if_index >= the one looked for
        const uint32_t mid = (lo + hi) >> 1;
        if (*reinterpret_cast<const uint32_t*>(tab + (size_t)mid * kNameRowBytes) < if_index) lo = mid + 1; else hi = mid;
    }
    const uint8_t* any = nullptr;
    for (uint32_t k = lo; k < n_names; k++) {
        const uint8_t* e = tab + (size_t)k * kNameRowBytes;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(e);
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        if (w0 != if_index) break;
        if ((w2 >> 16) & 0xffu) {                       // has_mac: mac bytes 4..9, byte 4 most significant
            const uint64_t m = ((uint64_t)__builtin_bswap32(w1) << 16) | (uint64_t)((w2 & 0xffu) << 8) | (uint64_t)((w2 >> 8) & 0xffu);
            if (m == mac48) return e;
        } else if (!any) any = e;
    }
    return any;
}

// Copy the namer table into LDS if it fits; returns the pointer the lookups use.
template <int THREADS>
NF_DEV const uint8_t* stage_names(const PbParams& P, uint32_t* lds_words) {
    if (P.n_names > kNamesLdsRows) return reinterpret_cast<const uint8_t*>(P.names);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(P.names);
    const uint32_t words = P.n_names * (kNameRowBytes / 4);
    for (uint32_t k = threadIdx.x; k < words; k += THREADS) lds_words[k] = src[k];
    __syncthreads();
    return reinterpret_cast<const uint8_t*>(lds_words);
}

// message DupMapEntry { string interface = 1; Direction direction = 2; string udn = 3; } as Record.dup_list (26)
template <typename S> NF_DEV void put_dup(S& s, const PbParams& P, const uint8_t* tab, uint32_t if_index, uint64_t mac48, uint32_t dir) {
    const uint8_t* e = lookup_name(tab, P.n_names, if_index, mac48);
    const uint32_t lens = e ? *reinterpret_cast<const uint32_t*>(e + 8) : 0u;      // bytes 8..11: mac[4..5], has_mac, name_len
    const uint32_t nlen = e ? lens >> 24 : P.unknown_len;
    const uint32_t ulen = e ? e[28] : 0u;
    const uint32_t body = (nlen ? 2 + nlen : 0) + uint_len(2, dir) + (ulen ? 2 + ulen : 0);
    put_tag(s, 26, 2);
    s.put((uint8_t)body);                  // < 128 by the table's field widths
    if (nlen) {
        s.put(0x0A); s.put((uint8_t)nlen);
        if (e) { for (uint32_t k = 0; k < nlen; k++) s.put(e[12 + k]); }
        else { for (uint32_t k = 0; k < nlen; k++) s.put((uint8_t)P.unknown[k]); }
    }
    put_uint(s, 2, dir);
    if (ulen) { s.put(0x1A); s.put((uint8_t)ulen); for (uint32_t k = 0; k < ulen; k++) s.put(e[29 + k]); }
}

NF_DEV uint64_t mac_be(uint64_t mac_le48) {   // Rec::smac() holds byte 0 in the low bits; macToUint64 (proto.go:246-253) wants it on top
    uint64_t v = 0;
    for (int b = 0; b < 6; b++) v = (v << 8) | ((mac_le48 >> (8 * b)) & 0xff);
    return v;
}

// google.protobuf.Duration = durationpb.New(time.Duration(d)): secs = d / 1e9 truncating, nanos = d - secs*1e9,
// both carry the sign; Nanos is an int32 (a negative one is sign-extended to a ten-byte varint).
template <typename S> NF_DEV void put_duration(S& s, uint32_t field, uint64_t d_u) {
    const int64_t d = (int64_t)d_u, secs = d / 1000000000ll;
    const uint64_t nanos = (uint64_t)(int64_t)(int32_t)(d - secs * 1000000000ll);
    put_tag(s, field, 2);
    s.put((uint8_t)(uint_len(1, (uint64_t)secs) + uint_len(2, nanos)));
    put_uint(s, 1, (uint64_t)secs);
    put_uint(s, 2, nanos);
}

// utils.DNSRawNameToDotted (pkg/utils/utils.go:18-58) over the 32-byte kernel copy at `raw` (the lane's LDS slot):
// bytes up to the first NUL, label by label; stops at a zero length, a compression pointer, or a label that
// runs past the end. EMIT = false only measures.
template <bool EMIT, typename S> NF_DEV uint32_t dns_dotted(S& s, const uint8_t* __restrict__ raw) {
    uint32_t nb = 0;
    while (nb < 32 && raw[nb] != 0) nb++;
    uint32_t i = 0, out = 0;
    while (i < nb) {
        const uint32_t l = raw[i];
        if (l == 0 || (l & 0xC0u) == 0xC0u) break;
        i++;
        if (i + l > nb) break;
        if (out) { if (EMIT) s.put('.'); out++; }
        if (EMIT) for (uint32_t k = 0; k < l; k++) s.put(raw[i + k]);
        out += l; i += l;
    }
    return out;
}

template <int N> NF_DEV void load_dwords16(const uint8_t* p, uint32_t (&w)[N]) {   // N/4 16-byte loads
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int k = 0; k < N / 4; k++) { const uint4 v = q[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
}
template <int N> NF_DEV void load_dwords8(const uint8_t* p, uint32_t (&w)[N]) {    // N/2 8-byte loads
    const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int k = 0; k < N / 2; k++) { const uint2 v = q[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
}

// The body of pbflow.Record for one flow: evicted record `r`, plus the feature parts of flow `i` when F carries
// them — each part is read whole with vector loads before anything is emitted. `name_lds`: 32 bytes of LDS owned by
// this lane (the DNS name is walked byte by byte). Same code sizes (CountSink) and writes (WindowSink).
template <typename S> NF_DEV void encode_record(S& s, const Rec& r, const PbParams& P, const PbFeat& F, uint64_t i, uint8_t* name_lds, const uint8_t* tab) {
    const uint32_t have = F.present ? F.present[i] : 0u;
    const bool add = F.additional && (have & 1u), dns = F.dns && (have & 2u), drp = F.drops && (have & 4u);
    const bool xlt = F.xlat && (have & 16u), quc = F.quic && (have & 32u);
    uint32_t addw[8] = {}, dnsw[16] = {}, drpw[8] = {}, xltw[14] = {}, qucw[6] = {};
    if (add) load_dwords16(F.additional + i * 32, addw);     // start@0 end@8 flow_rtt@16 ipsec_ret@24 eth@28 ipsec_encrypted@30
    if (dns) load_dwords16(F.dns + i * 64, dnsw);            // latency@16 id@24 flags@26 eth@28 errno@30 name@31..62
    if (drp) load_dwords16(F.drops + i * 32, drpw);          // bytes@16 packets@18 cause@20 flags@24 eth@26 state@28
    if (xlt) load_dwords8(F.xlat + i * 56, xltw);            // saddr@16 daddr@32 sport@48 dport@50 zone@52
    if (quc) load_dwords8(F.quic + i * 24, qucw);            // version@16 eth@20 long@22 short@23
    if (dns) {
#pragma unroll
        for (int k = 0; k < 32; k++) name_lds[k] = (uint8_t)(dnsw[(31 + k) >> 2] >> (8 * ((31 + k) & 3)));
    }
    const uint32_t eth = r.eth();
    const uint32_t dirn = r.d[24] & 0xffu;
    put_uint(s, 1, eth);
    put_uint(s, 2, dirn);
    put_time(s, 3, flow_time(P, r.start()));
    put_time(s, 4, flow_time(P, r.end()));
    const uint64_t smac = mac_be(r.smac()), dmac = mac_be(r.dmac());
    put_tag(s, 5, 2); s.put((uint8_t)(uint_len(1, smac) + uint_len(2, dmac)));
    put_uint(s, 1, smac); put_uint(s, 2, dmac);
    {   // Network: addresses by eth_protocol (proto.go:125-139), dscp
        const bool v6 = eth == 0x86DDu;    // model.IPv6Type
        const Ip4w sip{{r.d[0], r.d[1], r.d[2], r.d[3]}}, dip{{r.d[4], r.d[5], r.d[6], r.d[7]}};
        const uint32_t ipl = v6 ? 20 : 7;   // tag + len + body of one IP sub-message
        put_tag(s, 6, 2); s.put((uint8_t)(2 * ipl + uint_len(3, r.dscp())));
        put_ip(s, 1, sip, v6); put_ip(s, 2, dip, v6);
        put_uint(s, 3, r.dscp());
    }
    {   // Transport
        const uint32_t sp = r.d[8] & 0xffffu, dp = r.d[8] >> 16, pr = r.d[9] & 0xffu;
        put_tag(s, 7, 2); s.put((uint8_t)(uint_len(1, sp) + uint_len(2, dp) + uint_len(3, pr)));
        put_uint(s, 1, sp); put_uint(s, 2, dp); put_uint(s, 3, pr);
    }
    put_uint(s, 8, r.bytes());
    put_uint(s, 9, r.packets());
    put_ip(s, 12, Ip4w{{P.agent_ip_w[0], P.agent_ip_w[1], P.agent_ip_w[2], P.agent_ip_w[3]}}, !P.agent_is_v4);
    put_uint(s, 13, r.flags());
    put_uint(s, 14, (r.d[9] >> 8) & 0xffu);      // icmp_type
    put_uint(s, 15, (r.d[9] >> 16) & 0xffu);     // icmp_code
    if (drp) {                                   // proto.go:92-98
        const uint32_t bp = drpw[4], fe = drpw[6];
        put_uint(s, 16, bp & 0xffffu); put_uint(s, 17, bp >> 16);
        put_uint(s, 18, fe & 0xffffu); put_uint(s, 19, drpw[7] & 0xffu);
        put_uint(s, 20, drpw[5]);
    }
    if (dns) {                                   // proto.go:79-91, record.go:116-120
        const uint32_t idf = dnsw[6];
        put_uint(s, 21, idf & 0xffffu); put_uint(s, 22, idf >> 16);
        const uint64_t lat = (uint64_t)dnsw[4] | ((uint64_t)dnsw[5] << 32);
        if (lat) put_duration(s, 23, lat);
    }
    put_duration(s, 24, add ? ((uint64_t)addw[4] | ((uint64_t)addw[5] << 32)) : 0ull);   // time_flow_rtt = durationpb.New(fr.TimeFlowRtt): always present
    if (dns) put_uint(s, 25, (dnsw[7] >> 16) & 0xffu);
    {   // dup_list = record.Interfaces (record.go:100-114)
        const uint64_t lmac = mac_be(dirn == 0 ? r.dmac() : r.smac());
        put_dup(s, P, tab, r.d[21], lmac, dirn);
        uint32_t nb = r.d[24] >> 24; if (nb > 6) nb = 6;
        for (uint32_t k = 0; k < nb; k++) {
            const uint32_t od = k < 4 ? (r.d[25] >> (8 * k)) & 0xffu : (r.d[26] >> (8 * (k - 4))) & 0xffu;
            uint32_t oi = r.d[27];
#pragma unroll
            for (int q = 1; q < 6; q++) oi = ((uint32_t)q == k) ? r.d[27 + q] : oi;
            put_dup(s, P, tab, oi, lmac, od);
        }
    }
    if (xlt) {                                   // proto.go:99-105,129-138: address family by the FLOW's eth_protocol
        const bool v6 = eth == 0x86DDu;
        const uint32_t sd = xltw[12], ze = xltw[13];
        const uint32_t ipl = v6 ? 20 : 7;
        put_tag(s, 28, 2);
        s.put((uint8_t)(2 * ipl + uint_len(3, sd & 0xffffu) + uint_len(4, sd >> 16) + uint_len(5, ze & 0xffffu)));
        put_ip(s, 1, Ip4w{{xltw[4], xltw[5], xltw[6], xltw[7]}}, v6); put_ip(s, 2, Ip4w{{xltw[8], xltw[9], xltw[10], xltw[11]}}, v6);
        put_uint(s, 3, sd & 0xffffu); put_uint(s, 4, sd >> 16); put_uint(s, 5, ze & 0xffffu);
    }
    put_uint(s, 29, r.sampling());
    if (add) {                                   // proto.go:106-111
        put_uint(s, 30, ((addw[7] >> 16) & 0xffu) ? 1u : 0u);
        put_uint(s, 31, (uint64_t)(int64_t)(int32_t)addw[6]);
    }
    if (dns) {                                   // dns_name = 32
        CountSink c;
        const uint32_t nl = dns_dotted<false>(c, name_lds);
        if (nl) { put_tag(s, 32, 2); s.put((uint8_t)nl); dns_dotted<true>(s, name_lds); }
    }
    put_uint(s, 33, r.d[33] & 0xffffu);          // ssl_version
    put_uint(s, 34, (r.d[34] >> 24) & 1u);       // HasSSLMismatch (record.go:255-257)
    put_uint(s, 35, (r.d[34] >> 16) & 0xffu);    // tls_types
    put_uint(s, 36, r.d[33] >> 16);              // tls_cipher_suite
    put_uint(s, 37, r.d[34] & 0xffffu);          // tls_key_share
    if (quc) {                                   // proto.go:112-118
        const uint32_t ver = qucw[4], lh = (qucw[5] >> 16) & 0xffu, sh = qucw[5] >> 24;
        put_tag(s, 38, 2);
        s.put((uint8_t)(uint_len(1, ver) + uint_len(2, lh) + uint_len(3, sh)));
        put_uint(s, 1, ver); put_uint(s, 2, lh); put_uint(s, 3, sh);
    }
}

constexpr int kScanBlock = 1024;

// ---- kernel 1: body length per record, frame length, block-local exclusive scan of the frame lengths
__global__ __launch_bounds__(kScanBlock) void k_pb_size(const void* __restrict__ recs, uint64_t n, PbParams P, PbFeat F,
                                                        uint32_t* __restrict__ body_len, uint32_t* __restrict__ local_off,
                                                        uint32_t* __restrict__ block_sum) {
    __shared__ uint32_t wave_tot[kScanBlock / 64];
    __shared__ __align__(4) uint8_t name_lds[kScanBlock][32];
    __shared__ uint32_t tab_lds[kNamesLdsRows * (kNameRowBytes / 4)];
    const uint8_t* tab = stage_names<kScanBlock>(P, tab_lds);
    const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
    uint32_t frame = 0;
    if (i < n) {
        Rec r;
        load_record(recs, i, r);
        r.canonicalize();
        CountSink c;
        encode_record(c, r, P, F, i, name_lds[threadIdx.x], tab);
        body_len[i] = c.n;
        frame = 1 + varint_len(c.n) + c.n;
    }
    // inclusive scan inside the wave, then across the 16 waves
    uint32_t v = frame;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    if (i < n) local_off[i] = base + v - frame;
    if (threadIdx.x == kScanBlock - 1) block_sum[blockIdx.x] = base + v;
}

// ---- kernel 2: exclusive scan of the block sums (one workgroup), total in block_base[n_blocks]
__global__ __launch_bounds__(1024) void k_pb_scan_blocks(const uint32_t* __restrict__ block_sum, uint32_t n_blocks,
                                                         uint64_t* __restrict__ block_base) {
    __shared__ uint64_t part[1024];
    const uint32_t per = (n_blocks + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = (lo + per < n_blocks) ? lo + per : n_blocks;
    uint64_t s = 0;
    for (uint32_t k = lo; k < hi; k++) s += block_sum[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t acc = 0; for (int k = 0; k < 1024; k++) { const uint64_t x = part[k]; part[k] = acc; acc += x; } block_base[n_blocks] = acc; }
    __syncthreads();
    uint64_t acc = part[threadIdx.x];
    for (uint32_t k = lo; k < hi; k++) { block_base[k] = acc; acc += block_sum[k]; }
}

// ---- kernel 3: encode. One wave per 64 consecutive records, one kPbWindow-byte window of its output at a time.
// The window (LDS per wave, beside 2 KiB of DNS-name slots and the 8.6 KiB namer table) is chosen by the host from the
// average frame length the size pass found: the whole encoder runs once per window a wave's 64 frames span, so the
// window should hold them all (64 x ~110 B for Accounter records, 64 x ~270 B with every feature part).
// (a frame is at most 1033 bytes, DESIGN.md §4.7: any window of at least that size makes progress)

template <uint32_t kPbWindow>
__global__ __launch_bounds__(64) void k_pb_write(const void* __restrict__ recs, uint64_t n, PbParams P, PbFeat F,
                                                 const uint32_t* __restrict__ body_len, const uint32_t* __restrict__ local_off,
                                                 const uint64_t* __restrict__ block_base, uint8_t* __restrict__ out,
                                                 uint64_t* __restrict__ frame_offsets, uint8_t* __restrict__ kafka_keys) {
    __shared__ __align__(16) unsigned char lds[kPbWindow];
    __shared__ __align__(4) uint8_t name_lds[64][32];
    __shared__ uint32_t tab_lds[kNamesLdsRows * (kNameRowBytes / 4)];
    const uint8_t* tab = stage_names<64>(P, tab_lds);
    const uint64_t i0 = (uint64_t)blockIdx.x * 64, i = i0 + threadIdx.x;
    const uint64_t wave_base = block_base[i0 / kScanBlock] + local_off[i0];
    const uint32_t shift = (uint32_t)(wave_base & 15);       // the LDS image has the alignment of the destination
    uint64_t my_off = 0; uint32_t my_len = 0, bl = 0;
    Rec r;
    if (i < n) {
        load_record(recs, i, r);
        r.canonicalize();
        my_off = block_base[i / kScanBlock] + local_off[i];
        bl = body_len[i];
        my_len = 1 + varint_len(bl) + bl;
        frame_offsets[i] = my_off;
        if (kafka_keys) {
            // getFlowKey (kafka_proto.go:37-47): the two 16-byte addresses, the smaller one first
            int c = 0;
            for (int k = 0; k < 16 && c == 0; k++) {
                const int a = (r.d[k / 4] >> (8 * (k & 3))) & 0xff, b = (r.d[4 + k / 4] >> (8 * (k & 3))) & 0xff;
                c = a - b;
            }
            uint4* kk = reinterpret_cast<uint4*>(kafka_keys + i * 32);
            const uint4 sip = make_uint4(r.d[0], r.d[1], r.d[2], r.d[3]), dip = make_uint4(r.d[4], r.d[5], r.d[6], r.d[7]);
            kk[0] = c <= 0 ? sip : dip; kk[1] = c <= 0 ? dip : sip;
        }
        if (i == n - 1) frame_offsets[n] = my_off + my_len;
    }
    // total bytes of this wave = end of its last valid record
    uint64_t end = my_off + my_len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(end, d, 64); end = o > end ? o : end; }
    const uint32_t span = shift + (uint32_t)(end - wave_base);   // image bytes [shift, span)
    const uint32_t p0 = shift + (uint32_t)(my_off - wave_base);  // my frame = image bytes [p0, p0 + my_len)
    uint8_t* dst = out + (wave_base - shift);                     // 16-byte aligned
    for (uint32_t lo = 0; lo < span; lo += kPbWindow) {
        const uint32_t hi = lo + kPbWindow < span ? lo + kPbWindow : span;
        if (my_len && p0 < hi && p0 + my_len > lo) {
            WindowSink s{lds, p0, lo, hi - lo};
            s.put(0x0A);                                          // Records.entries = 1, length-delimited
            put_varint(s, bl);
            encode_record(s, r, P, F, i, name_lds[threadIdx.x], tab);
        }
        __syncthreads();
        for (uint32_t c = lo + threadIdx.x * 16; c < hi; c += 64 * 16) {
            if (c >= shift && c + 16 <= hi) {
                *reinterpret_cast<uint4*>(dst + c) = *reinterpret_cast<const uint4*>(lds + (c - lo));
            } else {
                for (uint32_t b = c < shift ? shift : c; b < c + 16 && b < hi; b++) dst[b] = lds[b - lo];
            }
        }
        __syncthreads();
    }
}

hipError_t launch_pb_size(const void* d_recs, uint64_t n, const PbParams& P, const PbFeat& F, uint32_t* d_body_len, uint32_t* d_local_off,
                          uint32_t* d_block_sum, uint64_t* d_block_base, hipStream_t s) {
    const uint32_t blocks = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_pb_size, dim3(blocks), dim3(kScanBlock), 0, s, d_recs, n, P, F, d_body_len, d_local_off, d_block_sum);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pb_scan_blocks, dim3(1), dim3(1024), 0, s, d_block_sum, blocks, d_block_base);
    return hipGetLastError();
}

hipError_t launch_scan_block_sums(const uint32_t* d_block_sum, uint32_t n_blocks, uint64_t* d_block_base, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_pb_scan_blocks, dim3(1), dim3(1024), 0, s, d_block_sum, n_blocks, d_block_base);
    return hipGetLastError();
}

hipError_t launch_pb_write(const void* d_recs, uint64_t n, const PbParams& P, const PbFeat& F, const uint32_t* d_body_len, const uint32_t* d_local_off,
                           const uint64_t* d_block_base, void* d_out, uint64_t* d_frame_offsets, void* d_kafka_keys, uint64_t total_bytes, hipStream_t s) {
    const uint64_t avg = n ? total_bytes / n : 0;
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    (void)hipGetLastError();
    if (avg <= 120)
        hipLaunchKernelGGL(k_pb_write<8192>, grid, block, 0, s, d_recs, n, P, F, d_body_len, d_local_off, d_block_base, (uint8_t*)d_out, d_frame_offsets, (uint8_t*)d_kafka_keys);
    else if (avg <= 248)
        hipLaunchKernelGGL(k_pb_write<16384>, grid, block, 0, s, d_recs, n, P, F, d_body_len, d_local_off, d_block_base, (uint8_t*)d_out, d_frame_offsets, (uint8_t*)d_kafka_keys);
    else
        hipLaunchKernelGGL(k_pb_write<24576>, grid, block, 0, s, d_recs, n, P, F, d_body_len, d_local_off, d_block_base, (uint8_t*)d_out, d_frame_offsets, (uint8_t*)d_kafka_keys);
    return hipGetLastError();
}

}  // namespace nfagg
