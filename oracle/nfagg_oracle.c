/*
 * nfagg_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY (see nfagg_oracle.h).
 *
 * Each function restates one piece of the reference and cites it. Paths are
 * relative to the reference repository root. Written from the reference's
 * behaviour, not copied: the reference is Go, this is C.
 */
#include "nfagg_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ================================================================== */
/* pkg/model/flow_content.go                                           */
/* ================================================================== */

static int mac_is_zero(const uint8_t m[6]) {                 /* flow_content.go:200-207 AllZerosMac */
    return (m[0] | m[1] | m[2] | m[3] | m[4] | m[5]) == 0;
}

/* flow_content.go:28-61 AccumulateBase(p, other), both non-nil */
void orc_accumulate_base(orc_flow_metrics* p, const orc_flow_metrics* other) {
    /* :36-38  time == 0 means "unset" */
    if (p->start == 0 || (p->start > other->start && other->start != 0))
        p->start = other->start;
    /* :39-41 */
    if (p->end == 0 || p->end < other->end)
        p->end = other->end;
    p->bytes += other->bytes;                                 /* :42 uint64 wraps */
    p->packets += other->packets;                             /* :43 uint32 wraps */
    p->flags |= other->flags;                                 /* :44 */
    if (other->eth_protocol != 0) p->eth_protocol = other->eth_protocol; /* :45-47 */
    if (mac_is_zero(p->src_mac)) memcpy(p->src_mac, other->src_mac, 6); /* :48-50 */
    if (mac_is_zero(p->dst_mac)) memcpy(p->dst_mac, other->dst_mac, 6); /* :51-53 */
    if (other->dscp != 0) p->dscp = other->dscp;              /* :54-56 */
    if (other->sampling != 0) p->sampling = other->sampling;  /* :57-59 */
}

/* flow_content.go:63-74 buildBaseFromAdditional */
static void base_from_additional(orc_flow_metrics* b, uint64_t start, uint64_t end, uint16_t eth) {
    if (b->start == 0 || (b->start > start && start != 0)) b->start = start;
    if (b->end == 0 || b->end < end) b->end = end;
    if (b->eth_protocol == 0) b->eth_protocol = eth;
}

uint16_t orc_add_uint16(uint16_t a, uint16_t b) {             /* :209-215 saturating */
    uint16_t s = (uint16_t)(a + b);
    return s < a ? 0xFFFF : s;
}

/* flow_content.go:76-96 */
void orc_accumulate_dns(orc_content* p, const orc_dns* o) {
    if (!o) return;
    base_from_additional(&p->base, o->start, o->end, o->eth_protocol);
    if (!p->has_dns) { p->dns = *o; p->has_dns = 1; return; } /* :81-84 adopted whole */
    p->dns.flags |= o->flags;
    if (o->id != 0) p->dns.id = o->id;
    if (p->dns.err_no != o->err_no) p->dns.err_no = o->err_no; /* :90-92 plain overwrite */
    if (p->dns.latency < o->latency) p->dns.latency = o->latency;
}

/* flow_content.go:98-118 */
void orc_accumulate_drops(orc_content* p, const orc_drops* o) {
    if (!o) return;
    base_from_additional(&p->base, o->start, o->end, o->eth_protocol);
    if (!p->has_drops) { p->drops = *o; p->has_drops = 1; return; }
    p->drops.bytes = orc_add_uint16(p->drops.bytes, o->bytes);
    p->drops.packets = orc_add_uint16(p->drops.packets, o->packets);
    p->drops.latest_flags |= o->latest_flags;
    if (o->latest_drop_cause != 0) p->drops.latest_drop_cause = o->latest_drop_cause;
    if (o->latest_state != 0) p->drops.latest_state = o->latest_state;
}

/* record.go:189-196 networkEventsMDExist */
static int netev_md_exists(const uint8_t ev[4][8], const uint8_t md[8]) {
    for (int i = 0; i < 4; i++) if (memcmp(ev[i], md, 8) == 0) return 1;
    return 0;
}

/* flow_content.go:120-137 */
void orc_accumulate_netev(orc_content* p, const orc_netev* o) {
    if (!o) return;
    base_from_additional(&p->base, o->start, o->end, o->eth_protocol);
    if (!p->has_netev) { p->netev = *o; p->has_netev = 1; return; }
    for (int i = 0; i < 4; i++) {
        if (o->packets[i] != 0 && !netev_md_exists(p->netev.network_events, o->network_events[i])) {
            uint8_t idx = p->netev.network_events_idx;
            if (idx >= 4) return; /* Go would panic on the out-of-range index */
            p->netev.bytes[idx] = orc_add_uint16(p->netev.bytes[idx], o->bytes[i]);
            p->netev.packets[idx] = orc_add_uint16(p->netev.packets[idx], o->packets[i]);
            memcpy(p->netev.network_events[idx], o->network_events[i], 8);
            p->netev.network_events_idx = (uint8_t)((idx + 1) % 4);
        }
    }
}

/* record.go:233-238 AllZeroIP: equal to 0.0.0.0 (v4-mapped) or :: */
static int ip_all_zero(const uint8_t ip[16]) {
    static const uint8_t v4zero[16] = {0,0,0,0,0,0,0,0,0,0,0xff,0xff,0,0,0,0};
    static const uint8_t v6zero[16] = {0};
    return memcmp(ip, v4zero, 16) == 0 || memcmp(ip, v6zero, 16) == 0;
}

/* flow_content.go:139-152 */
void orc_accumulate_xlat(orc_content* p, const orc_xlat* o) {
    if (!o) return;
    base_from_additional(&p->base, o->start, o->end, o->eth_protocol);
    if (!p->has_xlat) { p->xlat = *o; p->has_xlat = 1; return; }
    if (!ip_all_zero(o->saddr) && !ip_all_zero(o->daddr)) p->xlat = *o; /* replaced whole */
}

/* flow_content.go:154-177 */
void orc_accumulate_additional(orc_content* p, const orc_additional* o) {
    if (!o) return;
    base_from_additional(&p->base, o->start, o->end, o->eth_protocol);
    if (!p->has_additional) { p->additional = *o; p->has_additional = 1; return; }
    if (p->additional.flow_rtt < o->flow_rtt) p->additional.flow_rtt = o->flow_rtt;
    if (p->additional.ipsec_ret < o->ipsec_ret) {
        p->additional.ipsec_encrypted = o->ipsec_encrypted;
        p->additional.ipsec_ret = o->ipsec_ret;
    }
    if (p->additional.ipsec_ret == o->ipsec_ret) {
        if (o->ipsec_encrypted) p->additional.ipsec_encrypted = o->ipsec_encrypted;
    }
}

/* flow_content.go:179-198 */
void orc_accumulate_quic(orc_content* p, const orc_quic* o) {
    if (!o) return;
    base_from_additional(&p->base, o->start, o->end, o->eth_protocol);
    if (!p->has_quic) { p->quic = *o; p->has_quic = 1; return; }
    if (p->quic.version < o->version) p->quic.version = o->version;
    if (p->quic.seen_long_hdr < o->seen_long_hdr) p->quic.seen_long_hdr = o->seen_long_hdr;
    if (p->quic.seen_short_hdr < o->seen_short_hdr) p->quic.seen_short_hdr = o->seen_short_hdr;
}

/* ================================================================== */
/* pkg/tracer/tracer.go:1118-1146 + the closures at :1057-1110         */
/* ================================================================== */
void orc_rollup(int kind, const void* partials, size_t n_flows, size_t n_cpu,
                orc_flow_metrics* base, void* folded) {
    for (size_t f = 0; f < n_flows; f++) {
        orc_content c;
        memset(&c, 0, sizeof c);
        c.base = base[f];      /* flows[id], or zero metrics when not found (:1136-1139) */
        for (size_t k = 0; k < n_cpu; k++) {   /* CPU index ascending */
            size_t at = f * n_cpu + k;
            switch (kind) {
            case 0: orc_accumulate_additional(&c, (const orc_additional*)partials + at); break;
            case 1: orc_accumulate_dns(&c, (const orc_dns*)partials + at); break;
            case 2: orc_accumulate_drops(&c, (const orc_drops*)partials + at); break;
            case 3: orc_accumulate_netev(&c, (const orc_netev*)partials + at); break;
            case 4: orc_accumulate_xlat(&c, (const orc_xlat*)partials + at); break;
            case 5: orc_accumulate_quic(&c, (const orc_quic*)partials + at); break;
            }
        }
        base[f] = c.base;
        switch (kind) {
        case 0: ((orc_additional*)folded)[f] = c.additional; break;
        case 1: ((orc_dns*)folded)[f] = c.dns; break;
        case 2: ((orc_drops*)folded)[f] = c.drops; break;
        case 3: ((orc_netev*)folded)[f] = c.netev; break;
        case 4: ((orc_xlat*)folded)[f] = c.xlat; break;
        case 5: ((orc_quic*)folded)[f] = c.quic; break;
        }
    }
}

/* ================================================================== */
/* bpf/flows.c:76-143 — kernel "dedup" merge (mode 1). Pinned to oracle/_ref */
/* The incoming record plays the role of one observation (pkt + ifindex */
/* + direction + tls): what flow_monitor would have put in new_flow.    */
/* ================================================================== */
#define ORC_MAX_OBSERVED 6
#define ORC_DIR_BOTH 3
#define ORC_TLS_SERVER_HELLO 0x02   /* bpf/tls_tracker.h:19 */
#define ORC_MISC_SSL_MISMATCH 0x01  /* bpf/types.h MISC_FLAGS_SSL_MISMATCH */

static void add_observed_intf(orc_flow_metrics* v, uint32_t if_index, uint8_t direction) { /* :76-96 */
    if (v->nb_observed_intf >= ORC_MAX_OBSERVED) return;
    for (uint8_t i = 0; i < v->nb_observed_intf; i++) {
        if (v->observed_intf[i] == if_index) {
            if (v->observed_direction[i] != direction && v->observed_direction[i] != ORC_DIR_BOTH)
                v->observed_direction[i] = ORC_DIR_BOTH;
            return;
        }
    }
    v->observed_intf[v->nb_observed_intf] = if_index;
    v->observed_direction[v->nb_observed_intf] = direction;
    v->nb_observed_intf++;
}

static void update_existing_flow(orc_flow_metrics* agg, const orc_flow_metrics* o) {       /* :98-143 */
    uint32_t if_index = o->if_index_first_seen;
    if (agg->if_index_first_seen == if_index) {
        agg->packets += o->packets;       /* kernel: += 1 for its single packet */
        agg->bytes += o->bytes;
        agg->end = o->end;                /* assigned, not max (:108) */
        agg->flags |= o->flags;
        agg->dscp = o->dscp;
        agg->sampling = o->sampling;
        if (o->ssl_version > 0 && agg->ssl_version != o->ssl_version) {
            if (agg->ssl_version == 0) agg->ssl_version = o->ssl_version;
            else agg->misc_flags |= ORC_MISC_SSL_MISMATCH;
        }
        if (o->tls_cipher_suite > 0 && o->tls_types == ORC_TLS_SERVER_HELLO)
            agg->tls_cipher_suite = o->tls_cipher_suite;
        if (o->tls_key_share > 0 && o->tls_types == ORC_TLS_SERVER_HELLO)
            agg->tls_key_share = o->tls_key_share;
        agg->tls_types |= o->tls_types;
    } else if (if_index != 0) {
        agg->end = o->end;
        agg->flags |= o->flags;
        add_observed_intf(agg, if_index, o->direction_first_seen);
    }
}

/* ================================================================== */
/* pkg/flow/account.go — the Accounter                                  */
/* ================================================================== */
/* entries map[BpfFlowId]*BpfFlowMetrics (:22): open addressing over key
 * bytes with one heap node per flow (Go stores a pointer per entry too). The
 * table hash is FNV-1a — deliberately unrelated to the product's key hash. */
typedef struct { orc_flow_id key; orc_flow_metrics* val; } acc_slot;
struct orc_accounter {
    uint64_t max_entries;
    int mode;
    size_t cap, len;         /* cap is a power of two */
    acc_slot* slots;
};

static uint64_t fnv1a40(const uint8_t* k) {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 40; i++) { h ^= k[i]; h *= 1099511628211ull; }
    return h ^ (h >> 29);
}

static void acc_alloc(orc_accounter* a, size_t cap) {
    a->cap = cap; a->len = 0;
    a->slots = (acc_slot*)calloc(cap, sizeof(acc_slot));
}

orc_accounter* orc_acc_new(uint64_t max_entries, int mode) {
    orc_accounter* a = (orc_accounter*)calloc(1, sizeof *a);
    a->max_entries = max_entries; a->mode = mode;
    acc_alloc(a, 1024);
    return a;
}

static void acc_clear(orc_accounter* a) {
    for (size_t i = 0; i < a->cap; i++) free(a->slots[i].val);
    free(a->slots);
}

void orc_acc_free(orc_accounter* a) { if (a) { acc_clear(a); free(a); } }
size_t orc_acc_len(const orc_accounter* a) { return a->len; }

static acc_slot* acc_find(orc_accounter* a, const orc_flow_id* k) {
    size_t m = a->cap - 1, i = (size_t)fnv1a40((const uint8_t*)k) & m;
    while (a->slots[i].val && memcmp(&a->slots[i].key, k, 40) != 0) i = (i + 1) & m;
    return &a->slots[i];
}

static void acc_grow(orc_accounter* a) {
    acc_slot* old = a->slots; size_t oc = a->cap;
    size_t len = a->len;
    acc_alloc(a, oc * 2);
    for (size_t i = 0; i < oc; i++) if (old[i].val) *acc_find(a, &old[i].key) = old[i];
    a->len = len;
    free(old);
}

size_t orc_acc_ingest(orc_accounter* a, const void* records, size_t n) {
    const orc_flow_record* r = (const orc_flow_record*)records;
    for (size_t i = 0; i < n; i++) {
        orc_flow_id key = r[i].id;
        key.pad = 0;   /* Go's blank field: not part of map identity */
        acc_slot* s = acc_find(a, &key);
        if (s->val) {                                            /* account.go:82-83 */
            if (a->mode == 0) orc_accumulate_base(s->val, &r[i].metrics);
            else update_existing_flow(s->val, &r[i].metrics);
        } else {
            if (a->len >= a->max_entries) return i;              /* :85 -> caller evicts "full" */
            if ((a->len + 1) * 2 > a->cap) { acc_grow(a); s = acc_find(a, &key); }
            s->key = key;
            s->val = (orc_flow_metrics*)malloc(sizeof(orc_flow_metrics));
            *s->val = r[i].metrics;                              /* :95 first record stored whole */
            /* binary.Read skips blank fields: padding never reaches Go */
            memset(s->val->pad2, 0, 2); memset(s->val->pad4, 0, 4);
            a->len++;
        }
    }
    return n;
}

static uint32_t shard_mix(const orc_flow_id* id) {      /* cheap word mix, only used to split work between cores */
    uint64_t w[5]; memcpy(w, id, 40); w[4] &= 0x00FFFFFFFFFFFFFFull;
    uint64_t h = (w[0] ^ (w[1] * 0x9E3779B97F4A7C15ull)) + (w[2] ^ (w[3] * 0xC2B2AE3D27D4EB4Full)) + w[4] * 0x165667B19E3779F9ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return (uint32_t)h;
}

/* Best-effort multi-core CPU baseline (bench.py cpu_baseline.multicore): the records whose
 * FNV hash of the key falls into `shard` of `n_shards`, folded as orc_acc_ingest does; every worker scans the whole
 * batch and skips the records of the other shards. Not a reference path: the reference's Accounter is one goroutine. */
size_t orc_acc_ingest_shard(orc_accounter* a, const void* records, size_t n, uint32_t n_shards, uint32_t shard) {
    const orc_flow_record* r = (const orc_flow_record*)records;
    size_t i = 0, mine = 0;
    while (i < n) {
        size_t j = i;
        while (j < n && shard_mix(&r[j].id) % n_shards != shard) j++;
        if (j == n) break;
        size_t k = j + 1;   /* fold one record of this shard */
        if (orc_acc_ingest(a, &r[j], 1) != 1) return mine;
        mine++;
        i = k;
    }
    return mine;
}

static int rec_key_cmp(const void* x, const void* y) { return memcmp(x, y, 40); }

size_t orc_acc_evict(orc_accounter* a, void* out, size_t cap) {       /* account.go:102-124 */
    orc_flow_record* o = (orc_flow_record*)out;
    size_t n = 0;
    for (size_t i = 0; i < a->cap; i++) {
        if (!a->slots[i].val) continue;
        if (n < cap) { o[n].id = a->slots[i].key; o[n].metrics = *a->slots[i].val; }
        n++;
    }
    if (n <= cap) qsort(o, n, sizeof(orc_flow_record), rec_key_cmp);
    acc_clear(a);                       /* c.entries = map[...]{} (:68,87) */
    acc_alloc(a, 1024);
    return n;
}

/* pkg/model/record.go:90-97 */
void orc_record_times(int64_t now_unix_ns, uint64_t mono_now, const orc_flow_metrics* m,
                      int64_t* start_unix_ns, int64_t* end_unix_ns) {
    int64_t start_delta = (int64_t)(mono_now - m->start);   /* uint64 wrap, then time.Duration */
    int64_t end_delta = (int64_t)(mono_now - m->end);
    *start_unix_ns = (int64_t)((uint64_t)now_unix_ns - (uint64_t)start_delta);
    *end_unix_ns = (int64_t)((uint64_t)now_unix_ns - (uint64_t)end_delta);
}

/* ================================================================== */
/* Hashes and sketches — our own spec. PARITY UNPINNED.     */
/* ================================================================== */
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t fmix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33; return x;
}
static uint64_t le64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}
#define ORC_KMUL 0x9E3779B97F4A7C15ull

uint64_t orc_key_hash(const void* key40) {
    uint8_t k[40];
    memcpy(k, key40, 40); k[39] = 0;
    uint64_t h = 0x6E66616767206B31ull;
    for (int i = 0; i < 5; i++) h = (rotl64(h, 27) ^ le64(k + 8 * i)) * ORC_KMUL;
    return fmix64(h);
}

static const uint64_t ORC_IP_SEEDS[4] = {
    0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull };

uint64_t orc_ip_hash(const uint8_t ip[16], uint32_t seed_index) {
    uint64_t h = ORC_IP_SEEDS[seed_index & 3];
    h = (rotl64(h, 27) ^ le64(ip)) * ORC_KMUL;
    h = (rotl64(h, 27) ^ le64(ip + 8)) * ORC_KMUL;
    return fmix64(h);
}

uint32_t orc_shard_of(const void* key40, uint32_t n_shards) {
    if (n_shards <= 1) return 0;
    uint64_t hi = orc_key_hash(key40) >> 32;
    return (uint32_t)((hi * n_shards) >> 32);
}

static uint64_t cm_index(const uint8_t ip[16], uint32_t row, uint32_t log2w) {
    uint64_t ha = orc_ip_hash(ip, 0), hb = orc_ip_hash(ip, 1) | 1;
    return (ha + (uint64_t)row * hb) >> (64 - log2w);
}

void orc_cm_update(uint64_t* cm, uint32_t depth, uint32_t log2w, const uint8_t ip[16], uint64_t add) {
    for (uint32_t r = 0; r < depth; r++) cm[((uint64_t)r << log2w) + cm_index(ip, r, log2w)] += add;
}

uint64_t orc_cm_query(const uint64_t* cm, uint32_t depth, uint32_t log2w, const uint8_t ip[16]) {
    uint64_t best = ~0ull;
    for (uint32_t r = 0; r < depth; r++) {
        uint64_t v = cm[((uint64_t)r << log2w) + cm_index(ip, r, log2w)];
        if (v < best) best = v;
    }
    return best;
}

/* Heavy hitters (our own spec): among the distinct src (side 0) / dst (side 1) addresses of `records`, the k with
 * the largest Count-Min estimate; order: estimate descending, then address bytes ascending. out: k x {ip[16], u64 estimate}. */
typedef struct { uint8_t ip[16]; uint64_t est; } hh_row;
static int hh_ip_cmp(const void* a, const void* b) { return memcmp(a, b, 16); }
static int hh_rank_cmp(const void* a, const void* b) {
    const hh_row* x = (const hh_row*)a; const hh_row* y = (const hh_row*)b;
    if (x->est != y->est) return x->est > y->est ? -1 : 1;
    return memcmp(x->ip, y->ip, 16);
}
size_t orc_cm_topk(const uint64_t* cm, uint32_t depth, uint32_t log2w, const void* records, size_t n, int side, size_t k, void* out) {
    const orc_flow_record* r = (const orc_flow_record*)records;
    hh_row* rows = (hh_row*)calloc(n ? n : 1, sizeof *rows);
    for (size_t i = 0; i < n; i++) memcpy(rows[i].ip, side ? r[i].id.dst_ip : r[i].id.src_ip, 16);
    qsort(rows, n, sizeof *rows, hh_ip_cmp);
    size_t d = 0;
    for (size_t i = 0; i < n; i++) if (i == 0 || memcmp(rows[i].ip, rows[d - 1].ip, 16)) rows[d++] = rows[i];
    for (size_t i = 0; i < d; i++) rows[i].est = orc_cm_query(cm, depth, log2w, rows[i].ip);
    qsort(rows, d, sizeof *rows, hh_rank_cmp);
    if (d > k) d = k;
    memcpy(out, rows, d * sizeof *rows);
    free(rows);
    return d;
}

void orc_hll_update(uint8_t* regs, uint32_t p, const uint8_t ip[16]) {
    uint64_t h = orc_ip_hash(ip, 2);
    uint64_t idx = h >> (64 - p);
    uint64_t w = (h << p) | (1ull << (p - 1));
    uint8_t rho = (uint8_t)(__builtin_clzll(w) + 1);
    if (regs[idx] < rho) regs[idx] = rho;
}

/* Classic HyperLogLog (Flajolet et al. 2007) with the small-range linear
 * counting correction; 64-bit hash so no large-range correction. Straight
 * loop over the registers in index order. */
double orc_hll_estimate(const uint8_t* regs, uint32_t p) {
    uint64_t m = 1ull << p;
    double alpha = (m == 16) ? 0.673 : (m == 32) ? 0.697 : (m == 64) ? 0.709
                 : 0.7213 / (1.0 + 1.079 / (double)m);
    double sum = 0.0; uint64_t zeros = 0;
    for (uint64_t i = 0; i < m; i++) {
        sum += ldexp(1.0, -(int)regs[i]);
        if (regs[i] == 0) zeros++;
    }
    double e = alpha * (double)m * (double)m / sum;
    if (e <= 2.5 * (double)m && zeros != 0) e = (double)m * log((double)m / (double)zeros);
    return e;
}

void orc_sketch_ingest(const void* records, size_t n, uint64_t* cm_src, uint64_t* cm_dst,
                       uint32_t depth, uint32_t log2w, uint8_t* hll_src, uint8_t* hll_dst, uint32_t p) {
    const orc_flow_record* r = (const orc_flow_record*)records;
    for (size_t i = 0; i < n; i++) {
        if (cm_src) orc_cm_update(cm_src, depth, log2w, r[i].id.src_ip, r[i].metrics.bytes);
        if (cm_dst) orc_cm_update(cm_dst, depth, log2w, r[i].id.dst_ip, r[i].metrics.bytes);
        if (hll_src) orc_hll_update(hll_src, p, r[i].id.src_ip);
        if (hll_dst) orc_hll_update(hll_dst, p, r[i].id.dst_ip);
    }
}

/* ================================================================== */
/* Synthetic streams                                  */
/* ================================================================== */
uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* pkg/model/bench_fixtures_test.go:19-32 benchFlowID. byte(i>>16) etc. are
 * Go byte truncations. */
void orc_bench_flow_id(uint64_t i, orc_flow_id* id) {
    memset(id, 0, sizeof *id);
    id->src_ip[10] = id->src_ip[11] = 0xff;
    id->dst_ip[10] = id->dst_ip[11] = 0xff;
    id->src_ip[12] = 10; id->src_ip[13] = (uint8_t)(i >> 16); id->src_ip[14] = (uint8_t)(i >> 8); id->src_ip[15] = (uint8_t)i;
    id->dst_ip[12] = 10; id->dst_ip[13] = (uint8_t)(i >> 16); id->dst_ip[14] = (uint8_t)(i >> 8); id->dst_ip[15] = (uint8_t)(i + 1);
    id->src_port = (uint16_t)(1024 + (i % 60000));
    id->dst_port = 443;
    id->proto = 6;
    /* the reference's formula repeats keys beyond 2^24 members; fold the high
     * bits of i into the (otherwise zero) ICMP bytes so large populations stay unique */
    id->icmp_type = (uint8_t)(i >> 24);
    id->icmp_code = (uint8_t)(i >> 32);
}

/* pkg/model/bench_fixtures_test.go:36-50 benchFlowMetrics, with the per-record
 * terms driven by the stream position j. */
void orc_bench_record(uint64_t i, uint64_t j, orc_flow_record* r) {
    memset(r, 0, sizeof *r);
    orc_bench_flow_id(i, &r->id);
    orc_flow_metrics* m = &r->metrics;
    m->start = 1000000ull + j;
    m->end = 2000000ull + j;
    m->bytes = 1500ull * (1 + j % 10);
    m->packets = (uint32_t)(1 + j % 10);
    m->eth_protocol = 0x0800;
    m->flags = 0x10;
    m->src_mac[0] = 0x02; m->src_mac[5] = 0x01;
    m->dst_mac[0] = 0x02; m->dst_mac[5] = 0x02;
    m->if_index_first_seen = (uint32_t)(2 + i % 4);
    m->direction_first_seen = (uint8_t)(i % 2);
}

/* variant 1: every order-dependent field of AccumulateBase varies per record
 * (zeros included), identity fields differ between records of one key. */
static void variant1_scramble(uint64_t seed, uint64_t j, orc_flow_record* r) {
    orc_flow_metrics* m = &r->metrics;
    uint64_t a = orc_splitmix64(seed ^ (j * 0xD1B54A32D192ED03ull) ^ 0x5bd1e995);
    uint64_t b = orc_splitmix64(a);
    uint64_t c = orc_splitmix64(b);
    if ((a & 7) == 0) m->start = 0;                       /* unset start */
    if (((a >> 3) & 15) == 0) m->end = 0;
    m->flags = (uint16_t)(1u << ((a >> 8) & 15)) | (uint16_t)(((a >> 12) & 1) ? 0x10 : 0);
    switch ((a >> 16) & 3) { case 0: m->eth_protocol = 0; break; case 1: m->eth_protocol = 0x86DD; break; default: break; }
    switch ((a >> 18) & 3) { case 0: m->dscp = (uint8_t)((a >> 20) & 0x3f); break; default: m->dscp = 0; }
    switch ((a >> 26) & 3) { case 0: m->sampling = (uint32_t)(b & 0xffffffffu); break; case 1: m->sampling = 50; break; default: m->sampling = 0; }
    if (((a >> 28) & 3) == 0) memset(m->src_mac, 0, 6); else m->src_mac[4] = (uint8_t)(a >> 32);
    if (((a >> 30) & 3) == 0) memset(m->dst_mac, 0, 6); else m->dst_mac[3] = (uint8_t)(a >> 40);
    m->if_index_first_seen = (uint32_t)(1 + ((b >> 32) & 7));
    m->direction_first_seen = (uint8_t)((b >> 35) & 1);
    m->err_no = (uint8_t)((b >> 36) & 1 ? 16 : 7);
    m->lock = (uint32_t)((b >> 37) & 1);
    m->nb_observed_intf = (uint8_t)((b >> 38) % 7);
    for (int k = 0; k < 6; k++) {
        m->observed_direction[k] = (uint8_t)((c >> (2 * k)) & 3);
        m->observed_intf[k] = (uint32_t)((c >> (12 + 4 * k)) & 15);
    }
    m->ssl_version = (uint16_t)(((c >> 40) & 1) ? 0x0303 : 0x0304);
    m->tls_cipher_suite = (uint16_t)(c >> 44);
    m->tls_key_share = (uint16_t)(c >> 28);
    m->tls_types = (uint8_t)(1u << ((c >> 60) & 3));
    m->misc_flags = (uint8_t)((c >> 63) & 1);
    if (((b >> 41) & 31) == 0) { m->bytes = ~0ull - (b & 0xffff); m->packets = 0xffffff00u + (uint32_t)(c & 0xff); } /* wrap */
    /* dirty padding: must never influence identity or output */
    r->id.pad = (uint8_t)(b >> 48);
    m->pad2[0] = (uint8_t)(b >> 50); m->pad4[1] = (uint8_t)(b >> 52);
}

void orc_zipf_thresholds(uint64_t n_keys, double s, uint64_t* th) {
    long double total = 0.0L;
    for (uint64_t k = 1; k <= n_keys; k++) total += powl((long double)k, -(long double)s);
    long double acc = 0.0L;
    const long double two64 = 18446744073709551616.0L;
    for (uint64_t k = 1; k <= n_keys; k++) {
        acc += powl((long double)k, -(long double)s);
        long double v = acc / total * two64;
        th[k - 1] = (v >= two64) ? ~0ull : (uint64_t)v;
    }
    th[n_keys - 1] = ~0ull;
}

uint64_t orc_stream_key_index(uint64_t seed, uint64_t j, uint64_t n_keys,
                              const uint64_t* th, uint32_t hot_permille) {
    uint64_t u = orc_splitmix64(seed + j * 0x9E3779B97F4A7C15ull);
    if (hot_permille) {
        uint64_t v = orc_splitmix64(u ^ 0xA5A5A5A5A5A5A5A5ull);
        if (v % 1000 < hot_permille) return 0;
    }
    if (!th) return (uint64_t)(((unsigned __int128)u * n_keys) >> 64);
    /* first k with u <= th[k] */
    uint64_t lo = 0, hi = n_keys - 1;
    while (lo < hi) { uint64_t mid = lo + (hi - lo) / 2; if (u <= th[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

void orc_gen_stream(uint64_t seed, uint64_t j0, size_t n, uint64_t n_keys,
                    const uint64_t* th, uint32_t hot_permille, uint32_t variant,
                    const uint64_t* pop_index, void* out) {
    orc_flow_record* r = (orc_flow_record*)out;
    for (size_t t = 0; t < n; t++) {
        uint64_t j = j0 + t;
        uint64_t i = orc_stream_key_index(seed, j, n_keys, th, hot_permille);
        if (pop_index) i = pop_index[i];   /* rank -> population member (sharded populations) */
        orc_bench_record(i, j, &r[t]);
        if (variant == 1) variant1_scramble(seed, j, &r[t]);
        if (variant == 2) {   /* configs[4]: every flow is seen on two interfaces, in both directions */
            r[t].metrics.if_index_first_seen = (uint32_t)(2 + (j & 1));
            r[t].metrics.direction_first_seen = (uint8_t)((j >> 1) & 1);
        }
    }
}
