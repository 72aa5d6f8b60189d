// nfagg_ringbuf.hip — host-only: bulk drain of a BPF ring buffer into a staging buffer
// (SURVEY.md §8(f) rank 2). Restates ringReader.readRecord
// (vendor/github.com/cilium/ebpf/ringbuf/ring.go:44-101) as a batch loop: same order of
// checks per sample (empty -> stop; short header -> error; busy -> stop without consuming;
// discard -> skip; copy), one consumer-position store per batch instead of one per sample.
#include <string.h>
#include "../../include/nfagg.h"

namespace {
constexpr uint32_t kBusy = 0x80000000u;      // BPF_RINGBUF_BUSY_BIT
constexpr uint32_t kDiscard = 0x40000000u;   // BPF_RINGBUF_DISCARD_BIT
constexpr uint64_t kHdr = 8;                 // BPF_RINGBUF_HDR_SZ
constexpr uint32_t kRecord = 144;            // sizeof(flow_record_t), bpf/types.h:212-215
}

extern "C" int nfagg_ringbuf_drain(const nfagg_ringbuf* rb, void* dst, size_t cap_records,
                                   size_t* n_records, size_t* n_skipped, uint64_t* errno_counts) {
    if (!rb || !rb->data || !rb->producer_pos || !rb->consumer_pos || (!dst && cap_records) || !n_records) return NFAGG_EINVAL;
    if ((rb->mask & (rb->mask + 1)) != 0) return NFAGG_EINVAL;
    const uint64_t prod = __atomic_load_n(rb->producer_pos, __ATOMIC_ACQUIRE);
    uint64_t cons = __atomic_load_n(rb->consumer_pos, __ATOMIC_RELAXED);
    uint8_t* out = static_cast<uint8_t*>(dst);
    size_t n = 0, skipped = 0;
    int rc = NFAGG_OK;
    while (n < cap_records) {
        const uint64_t remaining = prod - cons;
        if (remaining == 0) break;                                   // errEOR
        if (remaining < kHdr) { rc = NFAGG_EINVAL; break; }          // "read record header": io.ErrUnexpectedEOF
        const uint64_t start = cons & rb->mask;
        // atomic (acquire) read of len: happens-before with the kernel's xchg at commit (ring.go:60-63)
        const uint32_t len = __atomic_load_n(reinterpret_cast<const uint32_t*>(rb->data + start), __ATOMIC_ACQUIRE);
        if (len & kBusy) break;                                      // errBusy: not committed yet, position not stored
        const uint32_t data_len = len & ~(kBusy | kDiscard);
        const uint64_t aligned = ((uint64_t)data_len + 7) & ~7ull;   // data is padded to 8 bytes
        if (prod - (cons + kHdr) < aligned) { rc = NFAGG_EINVAL; break; }   // "read sample data": io.ErrUnexpectedEOF
        const uint64_t dstart = (cons + kHdr) & rb->mask;
        cons += kHdr + aligned;
        if (len & kDiscard) { skipped++; continue; }
        if (data_len != kRecord) { skipped++; continue; }            // model.ReadFrom would fail on it
        const uint64_t size = rb->mask + 1;
        uint8_t* o = out + n * kRecord;
        if (dstart + kRecord <= size) memcpy(o, rb->data + dstart, kRecord);
        else {                                                       // wraps: the reference reads through the second mapping
            const uint64_t first = size - dstart;
            memcpy(o, rb->data + dstart, first);
            memcpy(o + first, rb->data, kRecord - first);
        }
        if (errno_counts) errno_counts[o[40 + 57]]++;                // metrics.errno @57
        n++;
    }
    __atomic_store_n(rb->consumer_pos, cons, __ATOMIC_RELEASE);
    *n_records = n;
    if (n_skipped) *n_skipped = skipped;
    return rc;
}
