/* nfagg_oracle_mt.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY: a partition-then-fold multi-core baseline for bench.py's
 * cpu_baseline.multicore ("a best-effort multi-threaded CPU variant … so the GPU is not compared to a
 * strawman"). Not a reference path — pkg/flow.Accounter is ONE goroutine (pkg/flow/account.go:58) — but what a CPU
 * implementation that wanted all cores would do, and the same decomposition the GPU's two-pass fold uses:
 *   phase 1 (partition)  T threads, thread t owns the contiguous slice t of the batch: hashes every key once, writes each
 *                        record's shard into a byte array and counts per (thread, shard); after a barrier a prefix sum over
 *                        (shard, thread) gives every (thread, shard) bucket its place in ONE index array, which the same
 *                        threads then fill — arrival order inside a shard is kept (slices are contiguous and in order);
 *   phase 2 (fold)       T threads, thread k folds shard k: the records its indices name, in arrival order, through the same
 *                        orc_acc_ingest as the single-core oracle (no record is looked at by more than one folder).
 * Round 3's variant let every thread scan the whole batch and skip (T-1)/T of it: 3.4x one core on 32 threads. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "nfagg_oracle.h"

typedef struct {
    const orc_flow_record* recs;
    size_t n;
    uint32_t T, t;
    uint8_t* shard_of;          /* n bytes */
    size_t* counts;             /* T x T: counts[t * T + s] = records of slice t that belong to shard s */
    size_t* start;              /* T x T: first position of bucket (t, s) in idx */
    uint32_t* idx;              /* n record indices grouped by shard, arrival order inside a shard */
    size_t* shard_begin;        /* T + 1 */
    uint64_t max_entries;
    int mode;
    size_t folded, flows;
} mt_job;

static uint32_t mt_mix(const orc_flow_id* id) {
    uint64_t w[5]; memcpy(w, id, 40); w[4] &= 0x00FFFFFFFFFFFFFFull;
    uint64_t h = (w[0] ^ (w[1] * 0x9E3779B97F4A7C15ull)) + (w[2] ^ (w[3] * 0xC2B2AE3D27D4EB4Full)) + w[4] * 0x165667B19E3779F9ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return (uint32_t)h;
}

static void slice_of(const mt_job* j, size_t* lo, size_t* hi) {
    const size_t per = (j->n + j->T - 1) / j->T;
    *lo = (size_t)j->t * per; *hi = *lo + per;
    if (*lo > j->n) *lo = j->n;
    if (*hi > j->n) *hi = j->n;
}

static void* mt_count(void* p) {
    mt_job* j = (mt_job*)p;
    size_t lo, hi; slice_of(j, &lo, &hi);
    size_t* c = j->counts + (size_t)j->t * j->T;
    for (size_t i = lo; i < hi; i++) {
        const uint8_t s = (uint8_t)(mt_mix(&j->recs[i].id) % j->T);
        j->shard_of[i] = s; c[s]++;
    }
    return 0;
}

static void* mt_fill(void* p) {
    mt_job* j = (mt_job*)p;
    size_t lo, hi; slice_of(j, &lo, &hi);
    size_t* at = j->start + (size_t)j->t * j->T;
    for (size_t i = lo; i < hi; i++) j->idx[at[j->shard_of[i]]++] = (uint32_t)i;
    return 0;
}

static void* mt_fold(void* p) {
    mt_job* j = (mt_job*)p;
    orc_accounter* a = orc_acc_new(j->max_entries, j->mode);
    const size_t lo = j->shard_begin[j->t], hi = j->shard_begin[j->t + 1];
    size_t k = lo;
    for (; k < hi; k++) if (orc_acc_ingest(a, &j->recs[j->idx[k]], 1) != 1) break;    /* a full shard stops (the bench's table never fills) */
    j->folded = k - lo; j->flows = orc_acc_len(a);
    orc_acc_free(a);
    return 0;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

static void run(mt_job* jobs, uint32_t T, void* (*fn)(void*)) {
    pthread_t* th = (pthread_t*)malloc(T * sizeof *th);
    for (uint32_t t = 0; t < T; t++) pthread_create(&th[t], 0, fn, &jobs[t]);
    for (uint32_t t = 0; t < T; t++) pthread_join(th[t], 0);
    free(th);
}

/* Returns the number of records folded (n unless a shard filled up); *flows = distinct flows over all shards;
 * seconds[0] = partition (count + prefix + fill), seconds[1] = fold, seconds[2] = the largest shard's share of the records (what
 * bounds the fold on a skewed stream: a key lives in ONE shard). T <= 256 threads = shards; n < 2^32. */
size_t orc_partition_fold_mt(const void* records, size_t n, uint32_t T, uint64_t max_entries, int mode, size_t* flows, double seconds[3]) {
    if (T == 0 || T > 256 || n >= 0xFFFFFFFFull) return 0;
    mt_job* jobs = (mt_job*)calloc(T, sizeof *jobs);
    uint8_t* shard_of = (uint8_t*)malloc(n ? n : 1);
    size_t* counts = (size_t*)calloc((size_t)T * T, sizeof *counts);
    size_t* start = (size_t*)calloc((size_t)T * T, sizeof *start);
    uint32_t* idx = (uint32_t*)malloc((n ? n : 1) * sizeof *idx);
    size_t* shard_begin = (size_t*)calloc(T + 1, sizeof *shard_begin);
    for (uint32_t t = 0; t < T; t++) {
        jobs[t].recs = (const orc_flow_record*)records; jobs[t].n = n; jobs[t].T = T; jobs[t].t = t;
        jobs[t].shard_of = shard_of; jobs[t].counts = counts; jobs[t].start = start; jobs[t].idx = idx; jobs[t].shard_begin = shard_begin;
        jobs[t].max_entries = max_entries; jobs[t].mode = mode;
    }
    const double t0 = now_s();
    run(jobs, T, mt_count);
    size_t at = 0;
    for (uint32_t s = 0; s < T; s++) {                          /* shard-major, slice order inside a shard = arrival order */
        shard_begin[s] = at;
        for (uint32_t t = 0; t < T; t++) { start[(size_t)t * T + s] = at; at += counts[(size_t)t * T + s]; }
    }
    shard_begin[T] = at;
    run(jobs, T, mt_fill);
    const double t1 = now_s();
    run(jobs, T, mt_fold);
    const double t2 = now_s();
    size_t folded = 0, fl = 0;
    for (uint32_t t = 0; t < T; t++) { folded += jobs[t].folded; fl += jobs[t].flows; }
    if (flows) *flows = fl;
    size_t biggest = 0;
    for (uint32_t s = 0; s < T; s++) if (shard_begin[s + 1] - shard_begin[s] > biggest) biggest = shard_begin[s + 1] - shard_begin[s];
    if (seconds) { seconds[0] = t1 - t0; seconds[1] = t2 - t1; seconds[2] = n ? (double)biggest / (double)n : 0.0; }
    free(jobs); free(shard_of); free(counts); free(start); free(idx); free(shard_begin);
    return folded;
}

/* ================================================================================================================== */
/* Local fold, then a key-sharded merge — the decomposition the GPU's own local fold uses (DESIGN.md §7 a'), on host    */
/* cores: bench.py's cpu_baseline.multicore_local_fold. Where partition-then-fold is bound by the shard of the hottest  */
/* flow (one folder gets 12 % of a Zipf(1.1) stream's records), here a hot flow costs every thread ONE table entry:     */
/*   phase 1 (fold)   thread t folds its contiguous slice, in arrival order, into a table of its own (AccumulateBase,    */
/*                    pkg/model/flow_content.go:28-61; first record stored whole, pkg/flow/account.go:95);               */
/*   phase 2 (merge)  the entries of every table bucketed by key shard; thread k merges shard k's entries from the       */
/*                    tables in slice order with the SAME AccumulateBase — it is its own ordered merge of partials: the  */
/*                    earlier slice's entry is `p`, the later one's `other` (min non-zero start, max end, sums, OR, last */
/*                    non-zero eth/dscp/sampling, first non-zero MACs, everything else from the earlier one).            */
/* Accounter mode only, no eviction on "full" (the bench's table never fills): returns 0 when a shard would exceed       */
/* max_entries. Not a reference path (pkg/flow.Accounter is ONE goroutine, account.go:58); bit-exact against the one-    */
/* core oracle in tests/test_oracle_mt.py.                                                                              */
/* ================================================================================================================== */
/* A table that never moves: 4-byte slots (index + 1 of a dense entry; sized once for the most items it can meet) over entries
 * appended in chunks — no rehash, no realloc, nothing freed while the threads run (every munmap is a TLB shoot-down for all of them:
 * the first version, which doubled its tables, got SLOWER with more threads). */
typedef struct { orc_flow_id key; orc_flow_metrics m; } lf_entry;                    /* 144 bytes */
#define LF_CHUNK_LOG2 14
#define LF_CHUNK (1u << LF_CHUNK_LOG2)
typedef struct { uint32_t* slots; size_t cap, len, n_chunks; lf_entry** chunks; } lf_table;

static uint64_t lf_hash(const orc_flow_id* id) {
    uint64_t w[5]; memcpy(w, id, 40); w[4] &= 0x00FFFFFFFFFFFFFFull;
    uint64_t h = (w[0] ^ (w[1] * 0x9E3779B97F4A7C15ull)) + (w[2] ^ (w[3] * 0xC2B2AE3D27D4EB4Full)) + w[4] * 0x165667B19E3779F9ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; h *= 0x94D049BB133111EBull; h ^= h >> 31;
    return h;
}
static uint32_t lf_shard(uint64_t h, uint32_t T) { return (uint32_t)(((h >> 40) * (uint64_t)T) >> 24); }   /* top 24 bits: not the slot bits */

static int lf_init(lf_table* t, size_t max_items) {
    size_t cap = 1024;
    while (cap < 2 * max_items) cap <<= 1;
    t->cap = cap; t->len = 0;
    t->n_chunks = max_items / LF_CHUNK + 2;
    t->slots = (uint32_t*)calloc(cap, sizeof(uint32_t));
    t->chunks = (lf_entry**)calloc(t->n_chunks, sizeof(lf_entry*));
    return t->slots && t->chunks;
}
static void lf_free(lf_table* t) {
    if (t->chunks) for (size_t c = 0; c < t->n_chunks; c++) free(t->chunks[c]);
    free(t->chunks); free(t->slots);
}
static lf_entry* lf_at(const lf_table* t, size_t i) { return &t->chunks[i >> LF_CHUNK_LOG2][i & (LF_CHUNK - 1)]; }
/* p = what the table holds for the key (earlier), other = what arrives (later): a record's metrics or a later slice's partial */
static int lf_upsert(lf_table* t, const orc_flow_id* key, const orc_flow_metrics* other, uint64_t h) {
    size_t m = t->cap - 1, i = (size_t)h & m;
    while (t->slots[i]) {
        lf_entry* e = lf_at(t, t->slots[i] - 1);
        if (memcmp(&e->key, key, 40) == 0) { orc_accumulate_base(&e->m, other); return 1; }
        i = (i + 1) & m;
    }
    const size_t idx = t->len;
    if ((idx >> LF_CHUNK_LOG2) >= t->n_chunks) return 0;
    if (!t->chunks[idx >> LF_CHUNK_LOG2] && !(t->chunks[idx >> LF_CHUNK_LOG2] = (lf_entry*)malloc(sizeof(lf_entry) * LF_CHUNK))) return 0;
    lf_entry* e = lf_at(t, idx);
    e->key = *key; e->m = *other;
    memset(e->m.pad2, 0, 2); memset(e->m.pad4, 0, 4);          /* binary.Read skips blank fields: padding never reaches Go */
    t->slots[i] = (uint32_t)idx + 1;
    t->len++;
    return 1;
}

/* Thread placement (bench.py's cpu_baseline; round-5 review item 7: the unpinned baseline varied 3.6 x by box and got slower with more
 * threads). orc_mt_set_pinning(1): thread t of the next runs binds itself to the t-th CPU of the list "node 0's CPUs, node 1's, ..."
 * (/sys/devices/system/node/node<k>/cpulist, restricted to the process's affinity mask): T threads fill one NUMA node before the next
 * is touched, and a thread stays where its table's pages were first touched. 0 (default): wherever the scheduler puts them. */
#include <sched.h>
static int g_pin = 0;
static int g_cpu_order[4096];
static int g_cpu_count = -1;
void orc_mt_set_pinning(int on) { g_pin = on; }
static void pin_build(void) {
    cpu_set_t allowed;
    g_cpu_count = 0;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    for (int node = 0; node < 64; node++) {
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        FILE* f = fopen(path, "r");
        if (!f) { if (node == 0) break; else continue; }
        char buf[4096];
        if (fgets(buf, sizeof buf, f)) {
            for (char* tok = strtok(buf, ",\n"); tok; tok = strtok(0, ",\n")) {
                int a, b;
                const int k = sscanf(tok, "%d-%d", &a, &b);
                if (k == 1) b = a;
                if (k >= 1) for (int c = a; c <= b && g_cpu_count < 4096; c++) if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) g_cpu_order[g_cpu_count++] = c;
            }
        }
        fclose(f);
    }
    if (g_cpu_count == 0)                                       /* no NUMA information: the allowed CPUs in numerical order */
        for (int c = 0; c < CPU_SETSIZE && g_cpu_count < 4096; c++) if (CPU_ISSET(c, &allowed)) g_cpu_order[g_cpu_count++] = c;
}
static void pin_self(uint32_t t) {
    if (!g_pin || g_cpu_count <= 0) return;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(g_cpu_order[t % (uint32_t)g_cpu_count], &one);
    (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
}

typedef struct lf_job_s {
    const orc_flow_record* recs;
    size_t n;
    uint32_t T, t;
    lf_table local, merged;
    uint32_t* items;            /* this thread's entries (indices) grouped by shard */
    size_t* shard_start;        /* T + 1 */
    struct lf_job_s* all;
    uint64_t max_entries;
    int overflow, failed;
} lf_job;

static void* lf_fold(void* p) {
    lf_job* j = (lf_job*)p;
    pin_self(j->t);
    const size_t per = (j->n + j->T - 1) / j->T;
    size_t lo = (size_t)j->t * per, hi = lo + per;
    if (lo > j->n) lo = j->n;
    if (hi > j->n) hi = j->n;
    if (!lf_init(&j->local, hi - lo)) { j->failed = 1; return 0; }
    for (size_t i = lo; i < hi; i++) {
        orc_flow_id key = j->recs[i].id;
        key.pad = 0;                                            /* Go's blank field: not part of map identity */
        if (!lf_upsert(&j->local, &key, &j->recs[i].metrics, lf_hash(&key))) { j->failed = 1; return 0; }
    }
    /* bucket the entries by key shard (counting sort of entry indices) */
    j->shard_start = (size_t*)calloc(j->T + 1, sizeof(size_t));
    j->items = (uint32_t*)malloc((j->local.len ? j->local.len : 1) * sizeof(uint32_t));
    size_t* at = (size_t*)malloc(j->T * sizeof(size_t));
    if (!j->shard_start || !j->items || !at) { j->failed = 1; free(at); return 0; }
    for (size_t i = 0; i < j->local.len; i++) j->shard_start[lf_shard(lf_hash(&lf_at(&j->local, i)->key), j->T) + 1]++;
    for (uint32_t s = 0; s < j->T; s++) j->shard_start[s + 1] += j->shard_start[s];
    memcpy(at, j->shard_start, j->T * sizeof(size_t));
    for (size_t i = 0; i < j->local.len; i++) j->items[at[lf_shard(lf_hash(&lf_at(&j->local, i)->key), j->T)]++] = (uint32_t)i;
    free(at);
    return 0;
}

static void* lf_merge(void* p) {
    lf_job* j = (lf_job*)p;
    pin_self(j->t);
    size_t mine = 0;
    for (uint32_t t = 0; t < j->T; t++) mine += j->all[t].shard_start[j->t + 1] - j->all[t].shard_start[j->t];
    if (!lf_init(&j->merged, mine)) { j->failed = 1; return 0; }
    for (uint32_t t = 0; t < j->T; t++) {                       /* slice order = arrival order between the tables */
        const lf_job* src = &j->all[t];
        for (size_t k = src->shard_start[j->t]; k < src->shard_start[j->t + 1]; k++) {
            const lf_entry* e = lf_at(&src->local, src->items[k]);
            if (!lf_upsert(&j->merged, &e->key, &e->m, lf_hash(&e->key))) { j->failed = 1; return 0; }
        }
    }
    if (j->merged.len > j->max_entries) j->overflow = 1;
    return 0;
}

static int lf_rec_cmp(const void* x, const void* y) { return memcmp(x, y, 40); }

/* Returns n (0 on overflow / bad arguments); *flows = distinct flows; seconds[0] = local folds, seconds[1] = bucket + merge,
 * seconds[2] = the largest shard's share of the merged entries. out (may be NULL): room for out_cap records; the merged flows,
 * sorted by key, when they fit. T <= 256; the local tables hold what their slices hold (memory: ~300 B per distinct flow and slice). */
size_t orc_local_fold_mt(const void* records, size_t n, uint32_t T, uint64_t max_entries, size_t* flows, double seconds[3], void* out, size_t out_cap) {
    if (T == 0 || T > 256 || n >= 0xFFFFFFFFull) return 0;
    if (g_pin && g_cpu_count < 0) pin_build();
    lf_job* jobs = (lf_job*)calloc(T, sizeof *jobs);
    for (uint32_t t = 0; t < T; t++) {
        jobs[t].recs = (const orc_flow_record*)records; jobs[t].n = n; jobs[t].T = T; jobs[t].t = t;
        jobs[t].all = jobs; jobs[t].max_entries = max_entries;
    }
    pthread_t* th = (pthread_t*)malloc(T * sizeof *th);
    const double t0 = now_s();
    for (uint32_t t = 0; t < T; t++) pthread_create(&th[t], 0, lf_fold, &jobs[t]);
    for (uint32_t t = 0; t < T; t++) pthread_join(th[t], 0);
    int failed = 0;
    for (uint32_t t = 0; t < T; t++) failed |= jobs[t].failed;
    const double t1 = now_s();
    if (!failed) {
        for (uint32_t t = 0; t < T; t++) pthread_create(&th[t], 0, lf_merge, &jobs[t]);
        for (uint32_t t = 0; t < T; t++) pthread_join(th[t], 0);
    }
    const double t2 = now_s();
    size_t fl = 0, entries = 0, biggest = 0;
    int overflow = failed;
    for (uint32_t t = 0; t < T && !failed; t++) {
        fl += jobs[t].merged.len; overflow |= jobs[t].overflow | jobs[t].failed;
        size_t mine = 0;
        for (uint32_t s = 0; s < T; s++) mine += jobs[s].shard_start[t + 1] - jobs[s].shard_start[t];
        entries += mine;
        if (mine > biggest) biggest = mine;
    }
    if (fl > max_entries) overflow = 1;
    if (flows) *flows = fl;
    if (seconds) { seconds[0] = t1 - t0; seconds[1] = t2 - t1; seconds[2] = entries ? (double)biggest / (double)entries : 0.0; }
    if (out && fl <= out_cap && !overflow) {
        orc_flow_record* o = (orc_flow_record*)out;
        size_t k = 0;
        for (uint32_t t = 0; t < T; t++)
            for (size_t i = 0; i < jobs[t].merged.len; i++) { const lf_entry* e = lf_at(&jobs[t].merged, i); o[k].id = e->key; o[k].metrics = e->m; k++; }
        qsort(o, k, sizeof(orc_flow_record), lf_rec_cmp);
    }
    for (uint32_t t = 0; t < T; t++) { lf_free(&jobs[t].local); lf_free(&jobs[t].merged); free(jobs[t].items); free(jobs[t].shard_start); }
    free(jobs); free(th);
    return overflow ? 0 : n;
}
