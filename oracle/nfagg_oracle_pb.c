/* nfagg_oracle_pb.c — CPU ORACLE (test infrastructure only): evicted flow_record_t ->
 * serialized pbflow.Record, restating
 *   pkg/model/record.go:82-125   NewRecord (times, interface list)
 *   pkg/pbflow/proto.go:40-149   FlowToPB
 *   proto/flow.proto:31-138      field numbers and types
 * and the protobuf wire format as google.golang.org/protobuf emits it (fields in field-number
 * order; proto3 scalars omitted at their zero value; a message field is emitted whenever the Go
 * pointer is non-nil, even if empty; a oneof member is emitted even at its zero value).
 * Pinned against golden vectors produced by the Python protobuf runtime from a descriptor that
 * mirrors flow.proto (tests/golden/gen_pb_golden.py). */
#include <string.h>
#include "nfagg_oracle.h"

typedef struct { uint8_t* p; } wr;

static void put_varint(wr* w, uint64_t v) {
    while (v >= 0x80) { *w->p++ = (uint8_t)(v | 0x80); v >>= 7; }
    *w->p++ = (uint8_t)v;
}
static void put_tag(wr* w, uint32_t field, uint32_t wt) { put_varint(w, ((uint64_t)field << 3) | wt); }
static void put_uint(wr* w, uint32_t field, uint64_t v) { if (v) { put_tag(w, field, 0); put_varint(w, v); } }
static void put_bytes(wr* w, uint32_t field, const void* b, size_t n) {
    put_tag(w, field, 2); put_varint(w, n); memcpy(w->p, b, n); w->p += n;
}
/* nested message built in a scratch buffer, then length-prefixed */
static void put_msg(wr* w, uint32_t field, const uint8_t* body, size_t n) { put_bytes(w, field, body, n); }

/* message IP { oneof { fixed32 ipv4 = 1; bytes ipv6 = 2; } } */
static size_t enc_ip(uint8_t* buf, const uint8_t ip[16], int v6) {
    wr w = { buf };
    if (v6) put_bytes(&w, 2, ip, 16);
    else {   /* model.IntEncodeV4 (record.go:202-204): big-endian value of the last four bytes, as fixed32 (little-endian on the wire) */
        uint32_t v = ((uint32_t)ip[12] << 24) | ((uint32_t)ip[13] << 16) | ((uint32_t)ip[14] << 8) | ip[15];
        put_tag(&w, 1, 5);
        *w.p++ = (uint8_t)v; *w.p++ = (uint8_t)(v >> 8); *w.p++ = (uint8_t)(v >> 16); *w.p++ = (uint8_t)(v >> 24);
    }
    return (size_t)(w.p - buf);
}

/* time.Time.Add(-delta) on (sec, nsec), then Timestamp{Seconds: Unix(), Nanos: Nanosecond()} */
static size_t enc_time(uint8_t* buf, int64_t now_unix_ns, uint64_t mono_now, uint64_t mono_ts) {
    int64_t sec = now_unix_ns / 1000000000; int64_t nsec = now_unix_ns % 1000000000;
    if (nsec < 0) { nsec += 1000000000; sec -= 1; }                 /* time.Unix normalises */
    int64_t delta = (int64_t)(mono_now - mono_ts);                   /* time.Duration(monotonicCurrentTime - ts), uint64 wrap */
    int64_t d = (int64_t)(0 - (uint64_t)delta);                      /* -delta (wraps for MinInt64 as Go does) */
    int64_t dsec = d / 1000000000; nsec += d % 1000000000;           /* Time.Add */
    if (nsec >= 1000000000) { dsec++; nsec -= 1000000000; } else if (nsec < 0) { dsec--; nsec += 1000000000; }
    sec += dsec;
    wr w = { buf };
    put_uint(&w, 1, (uint64_t)sec);                                  /* int64: negative -> 10-byte varint */
    put_uint(&w, 2, (uint64_t)nsec);
    return (size_t)(w.p - buf);
}

static const orc_intf_name* lookup_name(const orc_pb_options* o, uint32_t if_index, const uint8_t mac[6]) {
    const orc_intf_name* any = 0;
    for (uint32_t k = 0; k < o->n_names; k++) {
        const orc_intf_name* e = &o->names[k];
        if (e->if_index != if_index) continue;
        if (e->has_mac) { if (memcmp(e->mac, mac, 6) == 0) return e; }
        else if (!any) any = e;
    }
    return any;
}

/* message DupMapEntry { string interface = 1; Direction direction = 2; string udn = 3; } */
static size_t enc_dup(uint8_t* buf, const orc_pb_options* o, uint32_t if_index, const uint8_t mac[6], uint32_t dir) {
    wr w = { buf };
    const orc_intf_name* e = lookup_name(o, if_index, mac);
    if (e) { if (e->name_len) put_bytes(&w, 1, e->name, e->name_len); }
    else if (o->unknown_len) put_bytes(&w, 1, o->unknown_name, o->unknown_len);
    put_uint(&w, 2, dir);
    if (e && e->udn_len) put_bytes(&w, 3, e->udn, e->udn_len);
    return (size_t)(w.p - buf);
}

/* utils.DNSRawNameToDotted (pkg/utils/utils.go:18-58): bytes up to the first NUL, then label
 * by label; stops at a zero length, a compression pointer, or a label that runs past the end. */
size_t orc_dns_name_dotted(const char raw[32], char* out) {
    uint8_t b[32]; size_t nb = 0, no = 0, i = 0; int first = 1;
    while (nb < 32 && raw[nb] != 0) { b[nb] = (uint8_t)raw[nb]; nb++; }
    while (i < nb) {
        size_t l = b[i];
        if (l == 0) break;
        if ((l & 0xC0) == 0xC0) break;
        i++;
        if (i + l > nb) break;
        if (!first) out[no++] = '.';
        first = 0;
        memcpy(out + no, b + i, l); no += l; i += l;
    }
    return no;
}

/* durationpb.New(d) (types/known/durationpb): secs = d / 1e9 (truncating), nanos = d - secs*1e9;
 * Seconds int64 = 1, Nanos int32 = 2 (a negative int32 is sign-extended to a 10-byte varint). */
static size_t enc_duration(uint8_t* buf, int64_t d) {
    int64_t secs = d / 1000000000, nanos = d - secs * 1000000000;
    wr w = { buf };
    put_uint(&w, 1, (uint64_t)secs);
    put_uint(&w, 2, (uint64_t)(int64_t)(int32_t)nanos);
    return (size_t)(w.p - buf);
}

static size_t pb_encode(const orc_flow_id* id, const orc_flow_metrics* m, const orc_content* c, const orc_pb_options* o, uint8_t* out);

size_t orc_pb_encode_record(const orc_flow_record* r, const orc_pb_options* o, uint8_t* out) {
    return pb_encode(&r->id, &r->metrics, 0, o, out);
}

/* The MapTracer branch (pkg/flow/tracer_map.go:103-146): NewRecord + FlowToPB over a full
 * BpfFlowContent. The SampleDecoder is nil here (record.go:126 `s != nil &&`): network events are
 * not decoded, field 27 stays empty and no drop is injected from them. */
size_t orc_pb_encode_content(const orc_flow_id* id, const orc_content* c, const orc_pb_options* o, uint8_t* out) {
    return pb_encode(id, &c->base, c, o, out);
}

static size_t pb_encode(const orc_flow_id* id, const orc_flow_metrics* m, const orc_content* c, const orc_pb_options* o, uint8_t* out) {
    wr w = { out };
    uint8_t t[128], u[64];
    size_t n;
    put_uint(&w, 1, m->eth_protocol);
    put_uint(&w, 2, m->direction_first_seen);
    n = enc_time(t, o->now_unix_ns, o->mono_now_ns, m->start); put_msg(&w, 3, t, n);
    n = enc_time(t, o->now_unix_ns, o->mono_now_ns, m->end);   put_msg(&w, 4, t, n);
    {   /* DataLink: macToUint64 (proto.go:246-253) */
        uint64_t s = 0, d = 0;
        for (int k = 0; k < 6; k++) { s = (s << 8) | m->src_mac[k]; d = (d << 8) | m->dst_mac[k]; }
        wr x = { t }; put_uint(&x, 1, s); put_uint(&x, 2, d);
        put_msg(&w, 5, t, (size_t)(x.p - t));
    }
    {   /* Network: addresses by eth_protocol (proto.go:125-139), dscp */
        const int v6 = m->eth_protocol == 0x86DD;   /* model.IPv6Type */
        wr x = { t };
        n = enc_ip(u, id->src_ip, v6); put_msg(&x, 1, u, n);
        n = enc_ip(u, id->dst_ip, v6); put_msg(&x, 2, u, n);
        put_uint(&x, 3, m->dscp);
        put_msg(&w, 6, t, (size_t)(x.p - t));
    }
    {   /* Transport */
        wr x = { t }; put_uint(&x, 1, id->src_port); put_uint(&x, 2, id->dst_port); put_uint(&x, 3, id->proto);
        put_msg(&w, 7, t, (size_t)(x.p - t));
    }
    put_uint(&w, 8, m->bytes);
    put_uint(&w, 9, m->packets);
    {   /* agent_ip (proto.go:255-261): To4() != nil -> ipv4 */
        static const uint8_t v4pre[12] = {0,0,0,0,0,0,0,0,0,0,0xff,0xff};
        const int is4 = memcmp(o->agent_ip, v4pre, 12) == 0;
        n = enc_ip(t, o->agent_ip, !is4); put_msg(&w, 12, t, n);
    }
    put_uint(&w, 13, m->flags);
    put_uint(&w, 14, id->icmp_type);
    put_uint(&w, 15, id->icmp_code);
    if (c && c->has_drops) {                        /* proto.go:92-98 */
        put_uint(&w, 16, c->drops.bytes); put_uint(&w, 17, c->drops.packets);
        put_uint(&w, 18, c->drops.latest_flags); put_uint(&w, 19, c->drops.latest_state);
        put_uint(&w, 20, c->drops.latest_drop_cause);
    }
    if (c && c->has_dns) {                          /* proto.go:79-91 */
        put_uint(&w, 21, c->dns.id); put_uint(&w, 22, c->dns.flags);
        if (c->dns.latency != 0) { n = enc_duration(t, (int64_t)c->dns.latency); put_msg(&w, 23, t, n); }   /* record.go:116-120 */
    }
    {   /* time_flow_rtt = durationpb.New(fr.TimeFlowRtt): always present; record.go:121-125 */
        int64_t rtt = (c && c->has_additional) ? (int64_t)c->additional.flow_rtt : 0;
        n = enc_duration(t, rtt); put_msg(&w, 24, t, n);
    }
    if (c && c->has_dns) put_uint(&w, 25, c->dns.err_no);
    {   /* dup_list from record.Interfaces (record.go:100-114) */
        const uint8_t* lmac = m->direction_first_seen == 0 ? m->dst_mac : m->src_mac;
        n = enc_dup(t, o, m->if_index_first_seen, lmac, m->direction_first_seen); put_msg(&w, 26, t, n);
        uint32_t nb = m->nb_observed_intf; if (nb > 6) nb = 6;   /* the Go loop would index past [6]: clamp */
        for (uint32_t k = 0; k < nb; k++) {
            n = enc_dup(t, o, m->observed_intf[k], lmac, m->observed_direction[k]); put_msg(&w, 26, t, n);
        }
    }
    if (c && c->has_xlat) {                         /* proto.go:99-105,129-138: addresses by the FLOW's eth_protocol */
        const int v6 = m->eth_protocol == 0x86DD;
        wr x = { t };
        n = enc_ip(u, c->xlat.saddr, v6); put_msg(&x, 1, u, n);
        n = enc_ip(u, c->xlat.daddr, v6); put_msg(&x, 2, u, n);
        put_uint(&x, 3, c->xlat.sport); put_uint(&x, 4, c->xlat.dport); put_uint(&x, 5, c->xlat.zone_id);
        put_msg(&w, 28, t, (size_t)(x.p - t));
    }
    put_uint(&w, 29, m->sampling);
    if (c && c->has_additional) {                   /* proto.go:106-111 */
        put_uint(&w, 30, c->additional.ipsec_encrypted ? 1 : 0);
        put_uint(&w, 31, (uint64_t)(int64_t)c->additional.ipsec_ret);   /* int32: negative -> 10 bytes */
    }
    if (c && c->has_dns) {
        char name[64];
        size_t nl = orc_dns_name_dotted(c->dns.name, name);
        if (nl) put_bytes(&w, 32, name, nl);
    }
    put_uint(&w, 33, m->ssl_version);
    put_uint(&w, 34, (m->misc_flags & 1) ? 1 : 0);  /* HasSSLMismatch (record.go:255-257) */
    put_uint(&w, 35, m->tls_types);
    put_uint(&w, 36, m->tls_cipher_suite);
    put_uint(&w, 37, m->tls_key_share);
    if (c && c->has_quic) {                         /* proto.go:112-118 */
        wr x = { t };
        put_uint(&x, 1, c->quic.version); put_uint(&x, 2, c->quic.seen_long_hdr); put_uint(&x, 3, c->quic.seen_short_hdr);
        put_msg(&w, 38, t, (size_t)(x.p - t));
    }
    return (size_t)(w.p - out);
}

void orc_kafka_key(const orc_flow_record* r, uint8_t out[32]) {
    int c = memcmp(r->id.src_ip, r->id.dst_ip, 16);
    const uint8_t* a = c <= 0 ? r->id.src_ip : r->id.dst_ip;
    const uint8_t* b = c <= 0 ? r->id.dst_ip : r->id.src_ip;
    memcpy(out, a, 16); memcpy(out + 16, b, 16);
}
