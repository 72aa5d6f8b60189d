// nfagg_ringbuf.hip — host-only: bulk drain of a BPF ring buffer into a staging buffer
// (SURVEY.md §8(f) rank 2). Restates ringReader.readRecord
// (vendor/github.com/cilium/ebpf/ringbuf/ring.go:44-101) as a batch loop: same order of
// checks per sample (empty -> stop; short header -> error; busy -> stop without consuming;
// discard -> skip; copy), one consumer-position store per batch instead of one per sample.
#include <string.h>
#include "nfagg_hostpool.h"
#include "../../include/nfagg.h"

namespace {
constexpr uint32_t kBusy = 0x80000000u;      // BPF_RINGBUF_BUSY_BIT
constexpr uint32_t kDiscard = 0x40000000u;   // BPF_RINGBUF_DISCARD_BIT
constexpr uint64_t kHdr = 8;                 // BPF_RINGBUF_HDR_SZ
constexpr uint32_t kRecord = 144;            // sizeof(flow_record_t), bpf/types.h:212-215
constexpr uint64_t kStride = kHdr + kRecord; // a committed flow sample: header + 144 bytes (already a multiple of 8)
constexpr size_t kBulkMin = 32768;           // samples from which a drain is worth splitting over the copy workers (nfagg_hostpool.h)
constexpr unsigned kBulkMaxParts = 64;

inline void copy_sample(const nfagg_ringbuf* rb, uint64_t dstart, uint8_t* o) {
    const uint64_t size = rb->mask + 1;
    if (dstart + kRecord <= size) memcpy(o, rb->data + dstart, kRecord);
    else {                                                           // wraps: the reference reads through the second mapping
        const uint64_t first = size - dstart;
        memcpy(o, rb->data + dstart, first);
        memcpy(o + first, rb->data, kRecord - first);
    }
}
// the same with non-temporal stores (o 16-byte aligned: the staging buffer is, and 144 is a multiple of 16): the sample is on its
// way to the DMA engine, not to this core's cache
inline void copy_sample_stream(const nfagg_ringbuf* rb, uint64_t dstart, uint8_t* o) {
    if (dstart + kRecord > rb->mask + 1 || ((uintptr_t)o & 15)) { copy_sample(rb, dstart, o); return; }
    const uint8_t* s = rb->data + dstart;
    __m128i v[9];
    for (int k = 0; k < 9; k++) v[k] = _mm_loadu_si128((const __m128i*)(s + 16 * k));
    for (int k = 0; k < 9; k++) _mm_stream_si128((__m128i*)(o + 16 * k), v[k]);
}

// The ring of a busy agent holds nothing but committed 144-byte flow samples, 152 bytes apart. Under that assumption sample i of
// a run starts at cons + 152 i, so a run can be split over threads: each verifies as it copies that every header in its range
// says exactly "144 bytes, committed, not discarded" and stops at the first that does not. By induction the run is what the
// per-sample reader (ring.go:44-101) would have delivered up to the first such header of the first thread that met one; what later
// threads copied beyond it is discarded, and the sequential loop takes over at that sample. Returns the samples delivered.
size_t drain_uniform_run(const nfagg_ringbuf* rb, uint64_t cons, size_t want, uint8_t* out, uint64_t* errno_counts) {
    nfagg::HostPool& pool = nfagg::HostPool::get();
    unsigned T = pool.best_parts();
    if (T > want / (kBulkMin / 4)) T = (unsigned)(want / (kBulkMin / 4));
    if (T > kBulkMaxParts) T = kBulkMaxParts;
    if (T < 1) T = 1;
    const size_t per = (want + T - 1) / T;
    size_t ok[kBulkMaxParts];
    // (per-part errno histograms only when asked for: 2 KiB each)
    uint64_t* err = errno_counts ? static_cast<uint64_t*>(calloc((size_t)T * 256, sizeof(uint64_t))) : nullptr;
    const bool count_errno = errno_counts && err;
    auto work = [&](unsigned t) {
        const size_t lo = per * t, hi = lo + per < want ? lo + per : want;
        size_t i = lo;
        for (; i < hi; i++) {
            const uint64_t at = cons + kStride * i;
            const uint32_t len = __atomic_load_n(reinterpret_cast<const uint32_t*>(rb->data + (at & rb->mask)), __ATOMIC_ACQUIRE);
            if (len != kRecord) break;                               // busy, discarded or another length: the sequential reader decides
            uint8_t* o = out + i * kRecord;
            const uint64_t dstart = (at + kHdr) & rb->mask;
            copy_sample_stream(rb, dstart, o);
            if (count_errno) err[(size_t)t * 256 + rb->data[(dstart + 40 + 57) & rb->mask]]++;
        }
        _mm_sfence();
        ok[t] = i - lo;
    };
    pool.parallel(T, work);
    size_t n = 0;
    bool ended = false;
    for (unsigned t = 0; t < T; t++) {
        const size_t lo = per * t, hi = lo + per < want ? lo + per : want;
        if (!ended) {
            n += ok[t];
            if (count_errno) for (int e = 0; e < 256; e++) errno_counts[e] += err[(size_t)t * 256 + e];
            if (ok[t] < hi - lo) ended = true;                       // the run ends inside this part's range
        }
    }
    if (errno_counts && !err) {                                      // no memory for the per-part counts: count what was delivered, after the fact
        for (size_t i = 0; i < n; i++) errno_counts[out[i * kRecord + 40 + 57]]++;
    }
    free(err);
    return n;
}
}

// pkg/flow/limiter.go:28-38 for the burst of evictions one nfagg_account call delivers (include/nfagg.h).
extern "C" size_t nfagg_limit_batches(const uint64_t* epoch_end, size_t n_epochs, size_t queue_len, size_t queue_cap, uint8_t* keep,
                                      uint64_t* dropped_flows) {
    size_t kept = 0;
    uint64_t dropped = 0, lo = 0;
    for (size_t e = 0; e < n_epochs && epoch_end; e++) {
        const uint64_t flows = epoch_end[e] - lo;
        lo = epoch_end[e];
        const bool forward = queue_cap == 0 || queue_len < queue_cap;        // limiter.go:30
        if (forward) { kept++; if (queue_cap) queue_len++; }                 // `out <- i`: one more batch waits in the channel
        else dropped += flows;                                               // :33-34  Add(float64(len(i))); droppedFlows += len(i)
        if (keep) keep[e] = forward ? 1 : 0;
    }
    if (dropped_flows) *dropped_flows = dropped;
    return kept;
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
    size_t bulk_holdoff = 0;                                        // samples to take one by one after a run ended early
    while (n < cap_records) {
        uint64_t remaining = prod - cons;
        if (remaining == 0) break;                                   // errEOR
        if (bulk_holdoff) bulk_holdoff--;
        else {
            // bulk path: a long run of plain flow samples is copied by several threads (drain_uniform_run); whatever ends the run
            // — a busy, discarded or odd-length sample, the end of the ring content — is looked at by the code below, as before
            const size_t avail = (size_t)(remaining / kStride), room = cap_records - n;
            const size_t want = avail < room ? avail : room;
            if (want >= kBulkMin) {
                const size_t got = drain_uniform_run(rb, cons, want, out + n * kRecord, errno_counts);
                n += got; cons += kStride * got;
                if (got == want) continue;
                bulk_holdoff = 1024;                                 // a ring full of odd samples is not worth waking the workers each
                remaining = prod - cons;                             // > 0: the run ended at a sample, not at the end of the content
            }
        }
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
        uint8_t* o = out + n * kRecord;
        copy_sample(rb, dstart, o);
        if (errno_counts) errno_counts[o[40 + 57]]++;                // metrics.errno @57
        n++;
    }
    __atomic_store_n(rb->consumer_pos, cons, __ATOMIC_RELEASE);
    *n_records = n;
    if (n_skipped) *n_skipped = skipped;
    return rc;
}
