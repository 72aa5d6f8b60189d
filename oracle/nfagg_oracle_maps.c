/* nfagg_oracle_maps.c — CPU ORACLE (test infrastructure only): the merge of the drained eBPF maps,
 * restating FlowFetcher.LookupAndDeleteMap (pkg/tracer/tracer.go:1022-1116) and
 * lookupAndDeletePerCPUMap (:1118-1146):
 *   flows[id] = NewBpfFlowContent(base) for every entry of the main map (aggregated_flows), then, map by
 *   map in the order DNS, packet drops, network events, xlat, additional (RTT/IPsec), QUIC, for every id of
 *   that per-CPU map: flow = flows[id], or a content with zero base metrics when absent (:1136-1139);
 *   the n_cpu partials are accumulated in CPU order with the matching model.Accumulate*
 *   (pkg/model/flow_content.go); flows[id] = flow.
 * A key listed twice in one map models an id the iterator returned twice: the second LookupAndDelete
 * fails (the entry is gone) and the loop continues (:1048-1052, :1130-1134) — first occurrence wins.
 * Go map order is random: the result is returned sorted by the key bytes. Byte 39 of the id is a blank
 * field of the Go struct (pkg/ebpf/bpf_x86_bpfel.go:119): it takes no part in map-key equality and is zeroed. */
#include <stdlib.h>
#include <string.h>
#include "nfagg_oracle.h"

typedef struct node { orc_flow_id id; orc_content c; uint32_t seen; struct node* next; } node;

static uint64_t fnv(const void* k) {
    const uint8_t* p = (const uint8_t*)k; uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 39; i++) { h ^= p[i]; h *= 1099511628211ull; }   /* byte 39: blank field, not part of the Go key */
    return h;
}

static int node_cmp(const void* a, const void* b) { return memcmp(&(*(node* const*)a)->id, &(*(node* const*)b)->id, 39); }

size_t orc_map_merge(const orc_flow_id* main_ids, const orc_flow_metrics* main_vals, size_t n_main,
                     const orc_flow_id* const feat_ids[6], const void* const feat_vals[6], const size_t feat_n[6],
                     size_t n_cpu, orc_flow_id* out_ids, orc_content* out) {
    size_t total = n_main, nb = 64, n_nodes = 0;
    for (int k = 0; k < 6; k++) total += feat_n[k];
    while (nb < 2 * total) nb <<= 1;
    node** bucket = (node**)calloc(nb, sizeof *bucket);
    node** all = (node**)calloc(total ? total : 1, sizeof *all);
    /* main map: flows[id] = model.NewBpfFlowContent(baseMetrics) */
    for (size_t i = 0; i < n_main; i++) {
        uint64_t h = fnv(&main_ids[i]) & (nb - 1);
        node* p = bucket[h];
        while (p && memcmp(&p->id, &main_ids[i], 39)) p = p->next;
        if (p) continue;                                   /* listed twice: LookupAndDelete fails, continue */
        p = (node*)calloc(1, sizeof *p);
        p->id = main_ids[i]; p->id.pad = 0; p->c.base = main_vals[i]; p->seen = 1u << 6;
        p->next = bucket[h]; bucket[h] = p; all[n_nodes++] = p;
    }
    static const int order[6] = {1, 2, 3, 4, 0, 5};        /* tracer.go:1057-1110: dns, drops, netev, xlat, additional, quic */
    for (int q = 0; q < 6; q++) {
        const int k = order[q];
        for (size_t i = 0; i < feat_n[k]; i++) {
            const orc_flow_id* id = &feat_ids[k][i];
            uint64_t h = fnv(id) & (nb - 1);
            node* p = bucket[h];
            while (p && memcmp(&p->id, id, 39)) p = p->next;
            if (!p) {                                      /* not found: BpfFlowContent{BpfFlowMetrics: &ebpf.BpfFlowMetrics{}} */
                p = (node*)calloc(1, sizeof *p);
                p->id = *id; p->id.pad = 0;
                p->next = bucket[h]; bucket[h] = p; all[n_nodes++] = p;
            }
            if (p->seen & (1u << k)) continue;
            p->seen |= 1u << k;
            for (size_t c = 0; c < n_cpu; c++) {
                const size_t at = i * n_cpu + c;
                switch (k) {
                case 0: orc_accumulate_additional(&p->c, (const orc_additional*)feat_vals[k] + at); break;
                case 1: orc_accumulate_dns(&p->c, (const orc_dns*)feat_vals[k] + at); break;
                case 2: orc_accumulate_drops(&p->c, (const orc_drops*)feat_vals[k] + at); break;
                case 3: orc_accumulate_netev(&p->c, (const orc_netev*)feat_vals[k] + at); break;
                case 4: orc_accumulate_xlat(&p->c, (const orc_xlat*)feat_vals[k] + at); break;
                case 5: orc_accumulate_quic(&p->c, (const orc_quic*)feat_vals[k] + at); break;
                }
            }
        }
    }
    qsort(all, n_nodes, sizeof *all, node_cmp);
    for (size_t i = 0; i < n_nodes; i++) { out_ids[i] = all[i]->id; out[i] = all[i]->c; free(all[i]); }
    free(all); free(bucket);
    return n_nodes;
}
