/* nfagg_group_cdriver.c — the multi-GPU group of libnfagg driven from plain C, the way a cgo shim in the one-process
 * Go agent drives it (pkg/agent/agent.go:387-442): nfagg_group_create over the listed devices, nfagg_group_ingest with the
 * Accounter's evict-on-full loop (pkg/flow/account.go:81-96), the per-tick sketch merge (RCCL all-reduce when the members
 * sit on distinct devices — also at N = 1, where it degenerates but still goes through ncclCommInitAll / ncclAllReduce),
 * nfagg_group_evict. Writes <out>.records (every evicted batch back to back) and prints one line per eviction
 * "reason n_flows", then "hll_src <estimate of member 0 after the merge>".
 * usage: nfagg_group_cdriver <records.bin> <out-prefix> <max_entries> <batch_records> <devices e.g. 0 or 0,0,0> */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nfagg.h"

static void die(nfagg_group* g, const char* what, int rc) {
    fprintf(stderr, "%s failed: %d: %s\n", what, rc, nfagg_group_last_error(g) ? nfagg_group_last_error(g) : "");
    exit(2);
}

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s records.bin out-prefix max_entries batch devices\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t n = (size_t)ftell(f) / sizeof(nfagg_flow_record);
    fseek(f, 0, SEEK_SET);
    nfagg_flow_record* recs = malloc(n ? n * sizeof *recs : 1);
    if (fread(recs, sizeof *recs, n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    const uint64_t max_entries = strtoull(argv[3], 0, 10);
    const size_t batch = (size_t)strtoull(argv[4], 0, 10);
    int32_t devices[64]; uint32_t n_dev = 0;
    for (char* tok = strtok(argv[5], ","); tok && n_dev < 64; tok = strtok(0, ",")) devices[n_dev++] = atoi(tok);

    nfagg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.max_entries = max_entries;
    cfg.sketch_flags = NFAGG_SKETCH_CM | NFAGG_SKETCH_HLL;
    nfagg_group* g = 0;
    int rc = nfagg_group_create(&cfg, devices, n_dev, &g);
    if (rc != NFAGG_OK) die(0, "nfagg_group_create", rc);

    char path[4096];
    snprintf(path, sizeof path, "%s.records", argv[2]);
    FILE* fo = fopen(path, "wb");
    const size_t cap = (size_t)max_entries + n_dev;                   /* every shard holds at most ceil(max_entries / N) */
    nfagg_flow_record* out = malloc(cap * sizeof *out);
    size_t n_out = 0, off = 0;
    while (off < n) {
        const size_t m = n - off < batch ? n - off : batch;
        size_t consumed = 0;
        rc = nfagg_group_ingest(g, recs + off, m, &consumed);
        if (rc < 0) die(g, "nfagg_group_ingest", rc);
        off += consumed;
        if (rc == NFAGG_FULL) {
            if ((rc = nfagg_group_evict(g, NFAGG_REASON_FULL, out, cap, &n_out)) != NFAGG_OK) die(g, "nfagg_group_evict(full)", rc);
            fwrite(out, sizeof *out, n_out, fo);
            printf("full %zu\n", n_out);
        }
    }
    if ((rc = nfagg_group_merge_sketches(g)) != NFAGG_OK) die(g, "nfagg_group_merge_sketches", rc);
    double est = 0;
    if ((rc = nfagg_hll_estimate(nfagg_group_member(g, 0), NFAGG_HLL_SRC, &est)) != NFAGG_OK) die(g, "nfagg_hll_estimate", rc);
    if ((rc = nfagg_group_evict(g, NFAGG_REASON_CLOSING, out, cap, &n_out)) != NFAGG_OK) die(g, "nfagg_group_evict(closing)", rc);
    fwrite(out, sizeof *out, n_out, fo);
    fclose(fo);
    printf("closing %zu\n", n_out);
    printf("hll_src %.17g\n", est);
    nfagg_group_destroy(g);
    return 0;
}
