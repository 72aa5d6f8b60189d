// nfagg_hostpool.h — the host side's copy workers (one pool per process), shared by everything in libnfagg that moves records
// with host cores: the caller's buffer -> the pinned staging ring (nfagg_ingest / nfagg_account from pageable memory), the pinned
// bounce buffers -> the caller's eviction buffer, and the BPF ring buffer -> the staging buffer (nfagg_ringbuf_drain: what replaces
// RingBufTracer's one-record-per-read loop, pkg/flow/tracer_ringbuf.go:112-134, vendor/github.com/cilium/ebpf/ringbuf/ring.go:44-101).
//
// Why a pool and not std::thread per copy (rounds 2-4): a freshly spawned thread lands where the scheduler puts it — on the other
// socket as easily as next to the GPU — and the ring drain was bimodal from box to box and run to run (48-457 M records/s,
// profiles/r04_ring_drain_threads.txt); thread construction can also throw inside an extern "C" entry point. Here the workers
//   * are created once (nfagg_create of the first handle; nfagg_host_threads re-shapes the pool), creation failures leave a
//     smaller pool or none (the caller then copies inline: every job is also worked on by the thread that submits it),
//   * are bound to the CPUs of the NUMA node the GPU hangs off (/sys/bus/pci/devices/<bdf>/numa_node ->
//     /sys/devices/system/node/node<N>/cpulist), when that can be read,
//   * copy with non-temporal stores (the destination is a pinned buffer the DMA engine reads next, or a caller buffer nobody
//     reads soon: no read-for-ownership, no cache pollution),
//   * and how many PARTS a large copy is cut into is measured, not assumed: a calibration of a few milliseconds at pool creation
//     times a 32 MiB copy with 2, 4, 8, 12, 16 parts and keeps the fastest (round 4's constant, 4, was right on one box and
//     half the rate on the next).
#pragma once
#include <emmintrin.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace nfagg {

// dst and bytes multiples of 16 are streamed; anything else falls back to memcpy for the ragged ends
inline void copy_nt(void* dst, const void* src, size_t bytes) {
    uint8_t* d = static_cast<uint8_t*>(dst);
    const uint8_t* s = static_cast<const uint8_t*>(src);
    const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
    if (head) { const size_t h = head < bytes ? head : bytes; memcpy(d, s, h); d += h; s += h; bytes -= h; }
    size_t k = 0;
    for (; k + 64 <= bytes; k += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(s + k)), b = _mm_loadu_si128((const __m128i*)(s + k + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(s + k + 32)), e = _mm_loadu_si128((const __m128i*)(s + k + 48));
        _mm_stream_si128((__m128i*)(d + k), a); _mm_stream_si128((__m128i*)(d + k + 16), b);
        _mm_stream_si128((__m128i*)(d + k + 32), c); _mm_stream_si128((__m128i*)(d + k + 48), e);
    }
    for (; k + 16 <= bytes; k += 16) _mm_stream_si128((__m128i*)(d + k), _mm_loadu_si128((const __m128i*)(s + k)));
    if (k < bytes) memcpy(d + k, s + k, bytes - k);
    _mm_sfence();
}

class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }

    // (Re)shape the pool: `threads` workers (0 = keep / default 16; the submitting thread always works too), bound to NUMA node
    // `node` (-1 = leave them where they are). Returns the workers running.
    // explicit = the caller asked for this shape (nfagg_host_threads): the implicit default of the first nfagg_create leaves it alone
    // afterwards — an agent that keeps the pool off its own cores (threads = 2, node = -1) must not find it re-bound behind its back.
    unsigned configure(unsigned threads, int node, bool explicit_call = true) {
        std::lock_guard<std::mutex> cfg(cfg_mu_);
        if (!explicit_call && explicit_) return (unsigned)workers_.size();
        if (explicit_call) explicit_ = true;
        if (threads == 0) threads = workers_.empty() ? 16u : (unsigned)workers_.size();
        if (threads > 64) threads = 64;
        if (threads == workers_.size() && node == node_) return threads;
        stop_workers();
        node_ = node;
        cpu_set_t set;
        const bool bind = node >= 0 && node_cpus(node, &set);
        bool all_bound = bind;
        for (unsigned t = 0; t < threads; t++) {
            try {
                workers_.emplace_back([this] { work(); });
            } catch (...) { break; }                                  // thread limit reached: a smaller pool
            // (fails, or is intersected with the allowed set, under a cgroup cpuset: then the pool is NOT bound, and says so)
            if (bind && pthread_setaffinity_np(workers_.back().native_handle(), sizeof set, &set) != 0) all_bound = false;
        }
        bound_ = all_bound && !workers_.empty();
        n_workers_.store((unsigned)workers_.size(), std::memory_order_release);
        calibrate();
        return (unsigned)workers_.size();
    }
    unsigned workers() { std::lock_guard<std::mutex> cfg(cfg_mu_); return (unsigned)workers_.size(); }
    unsigned best_parts() const { return best_parts_.load(std::memory_order_relaxed); }
    int node() const { return node_; }
    bool bound() const { return bound_; }
    double calibrated_gbs() const { return calib_gbs_; }

    // fn(part) for part in [0, parts): by the workers and by the caller; returns when all are done. Several jobs may be in flight
    // (nfagg_account copies up and down at the same time): workers serve whichever has parts left. A job lives on its submitter's
    // stack: a worker's last touch of it is the `done` count, under the mutex the submitter waits under.
    void parallel(unsigned parts, const std::function<void(unsigned)>& fn) {
        // (the count, not the vector: configure() may be re-shaping it on another thread)
        if (parts <= 1 || n_workers_.load(std::memory_order_acquire) == 0) { for (unsigned p = 0; p < parts; p++) fn(p); return; }
        Job job; job.parts = parts; job.fn = &fn;
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        unsigned mine = 0;
        for (;;) {                                                   // the submitting thread takes parts of its own job
            const unsigned p = job.next.fetch_add(1, std::memory_order_relaxed);
            if (p >= parts) break;
            fn(p);
            mine++;
        }
        std::unique_lock<std::mutex> lk(mu_);
        for (size_t k = 0; k < jobs_.size(); k++) if (jobs_[k] == &job) { jobs_.erase(jobs_.begin() + k); break; }   // nothing left to take
        job.done += mine;
        done_cv_.wait(lk, [&] { return job.done == parts; });
    }

    // bytes from src to dst, cut into `parts` pieces on 4 KiB boundaries (0 = the calibrated number), streamed
    void copy(void* dst, const void* src, size_t bytes, unsigned parts = 0) {
        constexpr size_t kMinPerPart = 2u << 20;
        if (parts == 0) parts = best_parts();
        if (parts > bytes / kMinPerPart) parts = (unsigned)(bytes / kMinPerPart);
        if (parts <= 1) { copy_nt(dst, src, bytes); return; }
        const size_t per = ((bytes / parts) + 4095) & ~(size_t)4095;
        parallel(parts, [&](unsigned p) {
            const size_t lo = per * p;
            if (lo >= bytes) return;
            const size_t len = (lo + per < bytes && p + 1 < parts) ? per : bytes - lo;
            copy_nt((char*)dst + lo, (const char*)src + lo, len);
        });
    }

    ~HostPool() { std::lock_guard<std::mutex> cfg(cfg_mu_); stop_workers(); }

private:
    struct Job {
        unsigned parts = 0;
        const std::function<void(unsigned)>* fn = nullptr;
        std::atomic<unsigned> next{0};
        unsigned done = 0;                                           // under mu_
    };
    std::mutex mu_, cfg_mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<Job*> jobs_;
    std::vector<std::thread> workers_;
    bool stop_ = false, bound_ = false, explicit_ = false;
    std::atomic<unsigned> n_workers_{0};
    int node_ = -1;
    std::atomic<unsigned> best_parts_{4};
    double calib_gbs_ = 0.0;

    void work() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            Job* j = nullptr;
            unsigned p = 0;
            for (Job* c : jobs_) {
                const unsigned q = c->next.fetch_add(1, std::memory_order_relaxed);
                if (q < c->parts) { j = c; p = q; break; }
            }
            if (!j) {
                if (stop_) return;
                cv_.wait(lk);
                continue;
            }
            lk.unlock();
            (*j->fn)(p);
            lk.lock();
            if (++j->done == j->parts) done_cv_.notify_all();        // (the submitter may return as soon as the mutex is released)
        }
    }
    void stop_workers() {
        n_workers_.store(0, std::memory_order_release);              // parallel() copies inline while the pool is re-shaped
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) if (t.joinable()) t.join();
        workers_.clear();
        { std::lock_guard<std::mutex> lk(mu_); stop_ = false; }
    }
    static bool node_cpus(int node, cpu_set_t* set) {
        char path[96], buf[4096];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        FILE* f = fopen(path, "r");
        if (!f) return false;
        const bool ok = fgets(buf, sizeof buf, f) != nullptr;
        fclose(f);
        if (!ok) return false;
        CPU_ZERO(set);
        int n = 0;
        for (char* p = buf; *p && *p != '\n';) {                     // "0-63,128-191"
            char* e = nullptr;
            const long a = strtol(p, &e, 10);
            if (e == p) break;
            long b = a;
            p = e;
            if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
            for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, set); n++; }
            if (*p == ',') p++;
        }
        return n > 0;
    }
    // a few milliseconds, once per configure: into how many parts is a large streamed copy best cut on THIS host?
    void calibrate() {
        best_parts_.store(workers_.empty() ? 1u : 4u);
        if (workers_.empty()) return;
        constexpr size_t kBytes = 32u << 20;
        void *a = nullptr, *b = nullptr;
        if (posix_memalign(&a, 4096, kBytes) != 0 || posix_memalign(&b, 4096, kBytes) != 0) { free(a); free(b); return; }
        memset(a, 1, kBytes); memset(b, 2, kBytes);
        double best = 0.0;
        unsigned best_p = 4;
        const unsigned cand[] = {2, 4, 8, 12, 16, 24, 32};
        for (unsigned parts : cand) {
            if (parts > workers_.size() + 1) break;
            double t_best = 1e9;
            for (int rep = 0; rep < 2; rep++) {
                const auto t0 = std::chrono::steady_clock::now();
                copy(b, a, kBytes, parts);
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (dt < t_best) t_best = dt;
            }
            const double gbs = kBytes / t_best / 1e9;
            if (gbs > best * 1.05) { best = gbs; best_p = parts; }    // more parts only for a real gain: they are cores taken from the agent
        }
        best_parts_.store(best_p);
        calib_gbs_ = best;
        free(a); free(b);
    }
};

}  // namespace nfagg
