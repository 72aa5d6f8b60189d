/* nfagg_cdriver.c — libnfagg driven from plain C, the way the cgo shim of INTEGRATION.md drives it: no Python, no torch,
 * only include/nfagg.h and lib/libnfagg.so. Reads 144-byte flow_record_t from a file, feeds them through
 * nfagg_ingest in ring-sized batches with the Accounter's evict-on-full loop (pkg/flow/account.go:81-96), evicts on
 * close, serialises the evicted flows with nfagg_encode_pb, and writes
 *   <out>.records : every evicted batch back to back (144-byte records)
 *   <out>.pb      : the pbflow.Records frames of the LAST eviction
 *   stdout        : one line per eviction "reason n_flows", then "hll_src <estimate>" when sketches are on.
 * With a sixth argument "account" the batches go through nfagg_account instead — the evict-on-full loop runs inside the library —
 * from / into page-locked buffers (nfagg_host_alloc): the control flow of INTEGRATION.md section 3's GPUAccounter.flush.
 * With "ring" instead, the records reach the library the way the agent's RingBufTracer would hand them over in batches
 * (pkg/flow/tracer_ringbuf.go:112-134): a producer puts them into a BPF-style ring buffer — with discarded and wrong-length
 * samples in between, wrapping around the data area — and the consumer loop is nfagg_staging_acquire -> nfagg_ringbuf_drain
 * (straight into the pinned staging buffer) -> nfagg_staging_commit, with the same evict-on-full handling.
 * usage: nfagg_cdriver <records.bin> <out-prefix> <max_entries> <batch_records> <sketches 0|1> [account|ring]
 *   cc -std=c11 -O2 -I include tools/c/nfagg_cdriver.c -o nfagg_cdriver -L <libdir> -lnfagg -Wl,-rpath,<libdir> */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nfagg.h"

static void die(nfagg_handle* h, const char* what, int rc) {
    fprintf(stderr, "%s failed: %d: %s\n", what, rc, nfagg_last_error(h) ? nfagg_last_error(h) : "");
    exit(2);
}

/* ---- a BPF_MAP_TYPE_RINGBUF as the kernel leaves it in memory (kernel/bpf/ringbuf.c): 8-byte header {len | flags, pad}, data
 * padded to 8 bytes; the producer side of the "ring" mode */
#define RB_SIZE (1u << 16)
#define RB_DISCARD 0x40000000u
static uint8_t rb_data[RB_SIZE];
static volatile uint64_t rb_prod, rb_cons;
static int rb_push(const void* payload, uint32_t len, uint32_t flags) {
    const uint32_t total = 8 + ((len + 7u) & ~7u);
    if (rb_prod - rb_cons + total > RB_SIZE) return 0;           /* no room: the consumer must drain first */
    uint8_t blob[8 + 160];
    memset(blob, 0, sizeof blob);
    const uint32_t hdr = len | flags;
    memcpy(blob, &hdr, 4);
    memcpy(blob + 8, payload, len);
    for (uint32_t k = 0; k < total; k++) rb_data[(rb_prod + k) & (RB_SIZE - 1)] = blob[k];
    rb_prod += total;
    return 1;
}

int main(int argc, char** argv) {
    if (argc != 6 && argc != 7) { fprintf(stderr, "usage: %s records.bin out-prefix max_entries batch sketches [account|ring]\n", argv[0]); return 1; }
    const int account = argc == 7 && strcmp(argv[6], "account") == 0;
    const int ring = argc == 7 && strcmp(argv[6], "ring") == 0;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t n = (size_t)ftell(f) / sizeof(nfagg_flow_record);
    fseek(f, 0, SEEK_SET);
    nfagg_flow_record* recs = 0;
    if (account) { if (nfagg_host_alloc((n ? n : 1) * sizeof *recs, (void**)&recs) != NFAGG_OK) { fprintf(stderr, "nfagg_host_alloc failed\n"); return 1; } }
    else recs = malloc(n ? n * sizeof *recs : 1);
    if (fread(recs, sizeof *recs, n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    const uint64_t max_entries = strtoull(argv[3], 0, 10);
    const size_t batch = (size_t)strtoull(argv[4], 0, 10);
    const int sketches = atoi(argv[5]);

    nfagg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.max_entries = max_entries;
    cfg.sketch_flags = sketches ? (NFAGG_SKETCH_CM | NFAGG_SKETCH_HLL) : 0;
    if (ring) cfg.staging_records = batch;                       /* the pinned staging buffer takes one drain */
    nfagg_handle* h = 0;
    int rc = nfagg_create(&cfg, &h);
    if (rc != NFAGG_OK) die(0, "nfagg_create", rc);

    char path[4096];
    snprintf(path, sizeof path, "%s.records", argv[2]);
    FILE* fo = fopen(path, "wb");
    const size_t out_cap = (size_t)(max_entries ? max_entries : 1) + (account ? batch : 0);   /* account: a batch's evictions on full, too */
    nfagg_flow_record* out = 0;
    if (account) { if (nfagg_host_alloc(out_cap * sizeof *out, (void**)&out) != NFAGG_OK) die(h, "nfagg_host_alloc", NFAGG_ENOMEM); }
    else out = malloc(out_cap * sizeof *out);
    size_t n_out = 0, off = 0;
    const size_t max_epochs = batch / (size_t)(max_entries ? max_entries : 1) + 2;
    uint64_t* epoch_end = malloc(max_epochs * sizeof *epoch_end);
    while (account && off < n) {                                 /* GPUAccounter.flush: one call per batch, every eviction on full delivered */
        const size_t m = n - off < batch ? n - off : batch;
        size_t consumed = 0, n_epochs = 0;
        rc = nfagg_account(h, recs + off, m, out, out_cap, epoch_end, max_epochs, &n_epochs, &consumed);
        if (rc < 0) die(h, "nfagg_account", rc);
        off += consumed;
        uint64_t lo = 0;
        for (size_t e = 0; e < n_epochs; e++) {
            fwrite(out + lo, sizeof *out, (size_t)(epoch_end[e] - lo), fo);
            printf("full %zu\n", (size_t)(epoch_end[e] - lo));
            lo = epoch_end[e];
        }
    }
    if (ring) {
        rb_prod = rb_cons = (uint64_t)RB_SIZE * 5 - 200;         /* the first samples wrap around the data area */
        const nfagg_ringbuf rb = {rb_data, RB_SIZE - 1, &rb_prod, &rb_cons};
        size_t produced = 0, skipped_total = 0, injected = 0;
        nfagg_flow_record* rest = malloc((batch ? batch : 1) * sizeof *rest);
        while (produced < n || rb_prod != rb_cons) {
            while (produced < n) {                               /* the kernel side: fill the ring */
                const uint8_t junk[24] = {1, 2, 3};
                if (produced % 11 == 5 && (injected & 1) == 0) { if (!rb_push(junk, 24, 0)) break; injected |= 1; }               /* wrong length */
                if (produced % 13 == 7 && (injected & 2) == 0) { if (!rb_push(recs + produced, 144, RB_DISCARD)) break; injected |= 2; }  /* discarded */
                if (!rb_push(recs + produced, 144, 0)) break;
                produced++; injected = 0;
            }
            for (;;) {                                           /* the agent side: drain straight into the staging buffer */
                void* buf = 0; size_t cap = 0, got = 0, skipped = 0, consumed = 0;
                if ((rc = nfagg_staging_acquire(h, &buf, &cap)) != NFAGG_OK) die(h, "nfagg_staging_acquire", rc);
                if ((rc = nfagg_ringbuf_drain(&rb, buf, cap, &got, &skipped, 0)) != NFAGG_OK) die(h, "nfagg_ringbuf_drain", rc);
                skipped_total += skipped;
                memcpy(rest, buf, got * sizeof *rest);            /* only needed when the commit stops on full */
                rc = nfagg_staging_commit(h, got, &consumed);
                if (rc < 0) die(h, "nfagg_staging_commit", rc);
                size_t at = consumed;
                while (rc == NFAGG_FULL) {                       /* account.go:85-94: evict, then resubmit the rest */
                    if ((rc = nfagg_evict(h, NFAGG_REASON_FULL, out, (size_t)max_entries, &n_out)) != NFAGG_OK) die(h, "nfagg_evict(full)", rc);
                    fwrite(out, sizeof *out, n_out, fo);
                    printf("full %zu\n", n_out);
                    rc = nfagg_ingest(h, rest + at, got - at, &consumed);
                    if (rc < 0) die(h, "nfagg_ingest", rc);
                    at += consumed;
                }
                if (got < cap) break;                            /* ring empty */
            }
        }
        fprintf(stderr, "ring: %zu samples skipped (discarded / wrong length)\n", skipped_total);
        free(rest);
        off = n;
    }
    while (off < n) {                                            /* the record arm of Accounter.Account, batched */
        const size_t m = n - off < batch ? n - off : batch;
        size_t consumed = 0;
        rc = nfagg_ingest(h, recs + off, m, &consumed);
        if (rc < 0) die(h, "nfagg_ingest", rc);
        off += consumed;
        if (rc == NFAGG_FULL) {                                  /* account.go:85-94: evict, then resubmit the rest */
            if ((rc = nfagg_evict(h, NFAGG_REASON_FULL, out, (size_t)max_entries, &n_out)) != NFAGG_OK) die(h, "nfagg_evict(full)", rc);
            fwrite(out, sizeof *out, n_out, fo);
            printf("full %zu\n", n_out);
        }
    }
    if ((rc = nfagg_evict(h, NFAGG_REASON_CLOSING, out, (size_t)max_entries, &n_out)) != NFAGG_OK) die(h, "nfagg_evict(closing)", rc);
    fwrite(out, sizeof *out, n_out, fo);
    fclose(fo);
    printf("closing %zu\n", n_out);

    nfagg_intf_name names[2];
    memset(names, 0, sizeof names);
    names[0].if_index = 2; names[0].name_len = 4; memcpy(names[0].name, "eth0", 4);
    names[1].if_index = 3; names[1].name_len = 4; memcpy(names[1].name, "eth1", 4); names[1].udn_len = 7; memcpy(names[1].udn, "default", 7);
    nfagg_pb_options opt;
    memset(&opt, 0, sizeof opt);
    opt.struct_size = sizeof opt;
    opt.n_names = 2; opt.names = names;
    opt.now_unix_ns = 1700000000000000000ll; opt.mono_now_ns = 3000000;
    opt.agent_ip[10] = 0xff; opt.agent_ip[11] = 0xff; opt.agent_ip[12] = 10; opt.agent_ip[15] = 1;
    memcpy(opt.unknown_name, "unknown", 7); opt.unknown_len = 7;
    uint64_t* offs = malloc((n_out + 1) * sizeof *offs);
    uint32_t* lens = malloc((n_out ? n_out : 1) * sizeof *lens);
    size_t need = 0, cap = 64;
    uint8_t* pb = malloc(cap);
    while ((rc = nfagg_encode_pb(h, out, n_out, &opt, pb, cap, offs, lens, 0, &need)) == NFAGG_TRUNCATED) { cap = need; pb = realloc(pb, cap); }
    if (rc != NFAGG_OK) die(h, "nfagg_encode_pb", rc);
    snprintf(path, sizeof path, "%s.pb", argv[2]);
    fo = fopen(path, "wb"); fwrite(pb, 1, need, fo); fclose(fo);
    if (sketches) {
        double est = 0;
        if ((rc = nfagg_hll_estimate(h, NFAGG_HLL_SRC, &est)) != NFAGG_OK) die(h, "nfagg_hll_estimate", rc);
        printf("hll_src %.17g\n", est);
    }
    nfagg_destroy(h);
    if (account) { nfagg_host_free(recs); nfagg_host_free(out); }
    return 0;
}
