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
