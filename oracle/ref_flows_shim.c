/* oracle/_ref — the reference's OWN dedup merge, compiled from /root/reference/bpf where it lies.
 * TEST INFRASTRUCTURE ONLY (checker for the restatement in nfagg_oracle.c; never linked by the product).
 *
 * What is compiled from the reference, verbatim, at build time (oracle/Makefile, target `ref`):
 *   - bpf/types.h            (included by path: -I$(REF)/bpf) — flow_metrics, flow_id, pkt_info, tls_info, the counter enum
 *   - bpf/flows.c            the text from `add_observed_intf` up to (not including) `update_dns`, i.e. :75-143,
 *                            cut out by line pattern into oracle/_ref/flows_dedup.inc (a build output, git-ignored)
 *   - bpf/tls_tracker.h:19   the TLSTRACKER_BF_* #defines, cut into oracle/_ref/tls_defs.inc the same way
 * What this shim supplies in place of the kernel environment (nothing of it is arithmetic of the path):
 *   __u8.. typedefs (linux/types.h), struct bpf_spin_lock + lock/unlock (single-threaded: no-ops), BPF_PRINTK (no-op),
 *   increase_counter (counts per key so the test can also check OBSERVED_INTF_MISSED).
 * Then a small driver of our own: an open-addressed map keyed by the 40-byte id whose hit arm calls the reference's
 * update_existing_flow with the record as the observation — the same mapping nfagg_oracle.c's mode 1 documents:
 *   pkt.current_ts = end, pkt.flags = flags, pkt.dscp = dscp, len = bytes, sampling, if_index = if_index_first_seen,
 *   direction = direction_first_seen, tls = {ssl_version, tls_cipher_suite, tls_key_share, tls_types}.
 * The kernel counts ONE packet per call (flows.c:106); a ring-buffer record stands for `packets` packets, so the driver
 * adds the remaining `packets - 1` when (and only when) the reference counted the call (it watches aggregate->packets).
 */
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <linux/types.h>
typedef __s32 s32;          /* vmlinux.h names bpf/types.h relies on beyond its own u8..u64 typedefs */
typedef __s64 s64;

struct bpf_spin_lock { __u32 val; };
static inline void bpf_spin_lock(struct bpf_spin_lock* l) { (void)l; }
static inline void bpf_spin_unlock(struct bpf_spin_lock* l) { (void)l; }
#ifndef __always_inline
#define __always_inline inline __attribute__((always_inline))
#endif
#define BPF_PRINTK(fmt, args...) do { } while (0)

#include "types.h"              /* the reference's bpf/types.h */
#include "tls_defs.inc"         /* the reference's TLSTRACKER_BF_* defines */

static uint64_t ref_counters[MAX_COUNTERS];
static inline void increase_counter(u32 key) { if (key < MAX_COUNTERS) ref_counters[key]++; }

#include "flows_dedup.inc"      /* the reference's add_observed_intf + update_existing_flow, verbatim */

_Static_assert(sizeof(flow_metrics) == 104 && sizeof(flow_id) == 40, "ABI");
_Static_assert(sizeof(flow_record) == 144, "ABI");

/* One observation = one 144-byte record applied to an existing aggregate through the reference's function. */
void ref_update_existing_flow(void* aggregate104, const void* record144) {
    flow_record r;
    memcpy(&r, record144, sizeof r);
    flow_metrics* agg = (flow_metrics*)aggregate104;
    const flow_metrics* o = &r.metrics;
    pkt_info pkt;
    memset(&pkt, 0, sizeof pkt);
    pkt.id = &r.id;
    pkt.current_ts = o->end_mono_time_ts;
    pkt.flags = o->flags;
    pkt.dscp = o->dscp;
    tls_info tls;
    memset(&tls, 0, sizeof tls);
    tls.hello_version = o->ssl_version;
    tls.cipher_suite = o->tls_cipher_suite;
    tls.key_share = o->tls_key_share;
    tls.type = o->tls_types;
    u32 before = agg->packets;
    update_existing_flow(agg, &pkt, o->bytes, o->sampling, o->if_index_first_seen, o->direction_first_seen, &tls);
    if (agg->packets != before) agg->packets += o->packets - 1;   /* the record's other packets (see header) */
}

uint64_t ref_counter_observed_intf_missed(void) { return ref_counters[OBSERVED_INTF_MISSED]; }
void ref_counters_reset(void) { memset(ref_counters, 0, sizeof ref_counters); }

/* ---- driver: map[flow_id]*flow_metrics as account.go:82-95 keeps it, hit arm = the reference's merge ---- */
typedef struct { uint8_t key[40]; flow_metrics* val; } ref_slot;
typedef struct { ref_slot* slots; size_t cap, len; } ref_map;

static uint64_t ref_hash(const uint8_t* k) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 40; i++) h = (h ^ k[i]) * 0x100000001B3ull;
    return h ^ (h >> 31);
}
static ref_slot* ref_find(ref_map* m, const uint8_t* k) {
    size_t i = ref_hash(k) & (m->cap - 1);
    while (m->slots[i].val && memcmp(m->slots[i].key, k, 40) != 0) i = (i + 1) & (m->cap - 1);
    return &m->slots[i];
}
static void ref_grow(ref_map* m) {
    ref_slot* old = m->slots; size_t oc = m->cap;
    m->cap = oc * 2; m->slots = (ref_slot*)calloc(m->cap, sizeof(ref_slot));
    for (size_t i = 0; i < oc; i++) if (old[i].val) *ref_find(m, old[i].key) = old[i];
    free(old);
}
static int ref_cmp(const void* a, const void* b) { return memcmp(a, b, 40); }

/* Fold `n` records; when a new key arrives with len >= max_entries (account.go:85) everything accumulated is flushed
 * to `out` (sorted by key, batches back to back; batch sizes in batch_len[]). Returns the number of records written,
 * or (size_t)-1 if out/batch_len are too small. */
size_t ref_dedup_run(const void* records, size_t n, uint64_t max_entries, void* out, size_t cap,
                     size_t* batch_len, size_t batch_cap, size_t* n_batches) {
    const uint8_t* r = (const uint8_t*)records;
    flow_record* o = (flow_record*)out;
    ref_map m; m.cap = 1024; m.len = 0; m.slots = (ref_slot*)calloc(m.cap, sizeof(ref_slot));
    size_t written = 0, nb = 0; int fail = 0;
    for (size_t i = 0; i <= n && !fail; i++) {
        uint8_t key[40];
        ref_slot* s = NULL;
        if (i < n) {
            memcpy(key, r + i * 144, 40); key[39] = 0;          /* Go's blank field is not part of map identity */
            s = ref_find(&m, key);
            if (s->val) { ref_update_existing_flow(s->val, r + i * 144); continue; }
        }
        if (i == n || m.len >= max_entries) {                   /* closing, or evict-on-full before the insert */
            if (written + m.len > cap || nb >= batch_cap) { fail = 1; break; }
            size_t at = written;
            for (size_t k = 0; k < m.cap; k++) {
                if (!m.slots[k].val) continue;
                memcpy(&o[written].id, m.slots[k].key, 40);
                o[written].metrics = *m.slots[k].val;
                written++;
                free(m.slots[k].val);
            }
            qsort(o + at, written - at, 144, ref_cmp);
            batch_len[nb++] = written - at;
            memset(m.slots, 0, m.cap * sizeof(ref_slot)); m.len = 0;
            if (i == n) break;
            s = ref_find(&m, key);
        }
        if ((m.len + 1) * 2 > m.cap) { ref_grow(&m); s = ref_find(&m, key); }
        memcpy(s->key, key, 40);
        s->val = (flow_metrics*)malloc(sizeof(flow_metrics));
        memcpy(s->val, r + i * 144 + 40, 104);                  /* account.go:95: first record stored whole */
        memset((uint8_t*)s->val + 66, 0, 2); memset((uint8_t*)s->val + 100, 0, 4);   /* blank fields never reach Go */
        m.len++;
    }
    for (size_t k = 0; k < m.cap; k++) if (fail && m.slots[k].val) free(m.slots[k].val);
    free(m.slots);
    *n_batches = nb;
    return fail ? (size_t)-1 : written;
}
