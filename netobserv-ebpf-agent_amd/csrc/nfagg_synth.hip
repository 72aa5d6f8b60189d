// nfagg_synth.hip — synthetic flow_record_t streams generated ON THE DEVICE
// (bench/test support; not part of the drop-in ABI). 100 M records are
// 14.4 GB: they are produced directly in HBM instead of crossing PCIe.
//
// Population and per-record metrics follow the reference's benchmark fixtures
// (pkg/model/bench_fixtures_test.go:19-50 benchFlowID / benchFlowMetrics) as
// fixed in SURVEY.md §8(d); ranks are drawn Zipf(s) by inverse CDF over a
// 64-bit fixed-point threshold table with a counter-based RNG, so that the CPU
// generator in oracle/ reproduces every byte (tests/test_device_path_gpu.py::test_device_generator_matches_host_generator).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "nfagg_hash.h"

namespace {

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// 36 dwords of one record
struct R36 { uint32_t d[36]; };

__host__ __device__ inline void put8(R36& r, int byte, uint32_t v) {
    r.d[byte >> 2] = (r.d[byte >> 2] & ~(0xffu << (8 * (byte & 3)))) | ((v & 0xffu) << (8 * (byte & 3)));
}
__host__ __device__ inline void put16(R36& r, int byte, uint32_t v) { put8(r, byte, v); put8(r, byte + 1, v >> 8); }
__host__ __device__ inline void put32(R36& r, int byte, uint32_t v) { r.d[byte >> 2] = v; }
__host__ __device__ inline void put64(R36& r, int byte, uint64_t v) { r.d[byte >> 2] = (uint32_t)v; r.d[(byte >> 2) + 1] = (uint32_t)(v >> 32); }

// benchFlowID(i): src 10.(i>>16).(i>>8).i, dst ...(i+1), sport 1024+i%60000, dport 443, TCP;
// the high bits of i go to the ICMP bytes so populations above 2^24 stay unique.
__host__ __device__ inline void flow_id(R36& r, uint64_t i) {
    put16(r, 10, 0xffffu); put16(r, 26, 0xffffu);
    put8(r, 12, 10); put8(r, 13, (uint32_t)(i >> 16)); put8(r, 14, (uint32_t)(i >> 8)); put8(r, 15, (uint32_t)i);
    put8(r, 28, 10); put8(r, 29, (uint32_t)(i >> 16)); put8(r, 30, (uint32_t)(i >> 8)); put8(r, 31, (uint32_t)(i + 1));
    put16(r, 32, (uint32_t)(1024 + (i % 60000)));
    put16(r, 34, 443);
    put8(r, 36, 6);
    put8(r, 37, (uint32_t)(i >> 24));
    put8(r, 38, (uint32_t)(i >> 32));
}

constexpr int M = 40;  // metrics offset in the record

__host__ __device__ inline void bench_record(R36& r, uint64_t i, uint64_t j) {
    for (int k = 0; k < 36; k++) r.d[k] = 0;
    flow_id(r, i);
    put64(r, M + 0, 1000000ull + j);
    put64(r, M + 8, 2000000ull + j);
    put64(r, M + 16, 1500ull * (1 + j % 10));
    put32(r, M + 24, (uint32_t)(1 + j % 10));
    put16(r, M + 28, 0x0800);
    put16(r, M + 30, 0x10);
    put8(r, M + 32, 0x02); put8(r, M + 37, 0x01);
    put8(r, M + 38, 0x02); put8(r, M + 43, 0x02);
    put32(r, M + 44, (uint32_t)(2 + i % 4));
    put8(r, M + 56, (uint32_t)(i % 2));
}

// variant 1: scramble every order-dependent field (zeros included), identity
// fields and padding, so parity tests exercise first/last resolution.
__host__ __device__ inline void scramble(R36& r, uint64_t seed, uint64_t j) {
    const uint64_t a = splitmix64(seed ^ (j * 0xD1B54A32D192ED03ull) ^ 0x5bd1e995ull);
    const uint64_t b = splitmix64(a);
    const uint64_t c = splitmix64(b);
    if ((a & 7) == 0) put64(r, M + 0, 0);
    if (((a >> 3) & 15) == 0) put64(r, M + 8, 0);
    put16(r, M + 30, (uint32_t)((1u << ((a >> 8) & 15)) | (((a >> 12) & 1) ? 0x10u : 0u)) & 0xffffu);
    switch ((a >> 16) & 3) { case 0: put16(r, M + 28, 0); break; case 1: put16(r, M + 28, 0x86DD); break; default: break; }
    if (((a >> 18) & 3) == 0) put8(r, M + 58, (uint32_t)((a >> 20) & 0x3f)); else put8(r, M + 58, 0);
    switch ((a >> 26) & 3) { case 0: put32(r, M + 52, (uint32_t)(b & 0xffffffffu)); break; case 1: put32(r, M + 52, 50); break; default: put32(r, M + 52, 0); }
    if (((a >> 28) & 3) == 0) { for (int k = 0; k < 6; k++) put8(r, M + 32 + k, 0); } else put8(r, M + 36, (uint32_t)(a >> 32));
    if (((a >> 30) & 3) == 0) { for (int k = 0; k < 6; k++) put8(r, M + 38 + k, 0); } else put8(r, M + 41, (uint32_t)(a >> 40));
    put32(r, M + 44, (uint32_t)(1 + ((b >> 32) & 7)));
    put8(r, M + 56, (uint32_t)((b >> 35) & 1));
    put8(r, M + 57, ((b >> 36) & 1) ? 16u : 7u);
    put32(r, M + 48, (uint32_t)((b >> 37) & 1));
    put8(r, M + 59, (uint32_t)((b >> 38) % 7));
    for (int k = 0; k < 6; k++) {
        put8(r, M + 60 + k, (uint32_t)((c >> (2 * k)) & 3));
        put32(r, M + 68 + 4 * k, (uint32_t)((c >> (12 + 4 * k)) & 15));
    }
    put16(r, M + 92, ((c >> 40) & 1) ? 0x0303u : 0x0304u);
    put16(r, M + 94, (uint32_t)(c >> 44) & 0xffffu);
    put16(r, M + 96, (uint32_t)(c >> 28) & 0xffffu);
    put8(r, M + 98, 1u << ((c >> 60) & 3));
    put8(r, M + 99, (uint32_t)((c >> 63) & 1));
    if (((b >> 41) & 31) == 0) { put64(r, M + 16, ~0ull - (b & 0xffff)); put32(r, M + 24, 0xffffff00u + (uint32_t)(c & 0xff)); }
    put8(r, 39, (uint32_t)(b >> 48));            // dirty padding on purpose
    put8(r, M + 66, (uint32_t)(b >> 50));
    put8(r, M + 101, (uint32_t)(b >> 52));
}

__host__ __device__ inline uint64_t key_index(uint64_t seed, uint64_t j, uint64_t n_keys, const uint64_t* th, uint32_t hot_permille) {
    const uint64_t u = splitmix64(seed + j * 0x9E3779B97F4A7C15ull);
    if (hot_permille) {
        const uint64_t v = splitmix64(u ^ 0xA5A5A5A5A5A5A5A5ull);
        if (v % 1000 < hot_permille) return 0;
    }
    if (!th) return mulhi64(u, n_keys);
    uint64_t lo = 0, hi = n_keys - 1;
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (u <= th[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

__global__ __launch_bounds__(256) void k_synth(uint4* __restrict__ out, uint64_t n, uint64_t j0, uint64_t seed, uint64_t n_keys,
                                               const uint64_t* __restrict__ th, uint32_t hot_permille, uint32_t variant,
                                               const uint64_t* __restrict__ pop_index) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const uint64_t j = j0 + t;
        uint64_t i = key_index(seed, j, n_keys, th, hot_permille);
        if (pop_index) i = pop_index[i];
        R36 r;
        bench_record(r, i, j);
        if (variant == 1) scramble(r, seed, j);
        if (variant == 2) { put32(r, M + 44, (uint32_t)(2 + (j & 1))); put8(r, M + 56, (uint32_t)((j >> 1) & 1)); }  // configs[4]: two interfaces, both directions
        uint4* o = out + t * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = make_uint4(r.d[4 * k], r.d[4 * k + 1], r.d[4 * k + 2], r.d[4 * k + 3]);
    }
}

}  // namespace

// ---- the chip's own yardsticks for bench.py ("confirm with a stream microbench in the same run"): what HBM delivers
// to kernels shaped like the fold's, measured with HIP events in the run that reports them. Round 5 used torch.Tensor.copy_ (a
// library blit: 4.8 TB/s) where MI355X_MICROARCH.md measures 6.3 TB/s for a float4 copy kernel.
//   y_read_records   144-byte records, one per lane and trip, the first 112 bytes as seven 16-byte loads — what pass 1 of the fold
//                    issues (tools/pmc_calib.hip calib_read_records); whole lines cross the fabric: n x 144 bytes
//   y_read_stream    a plain 16-byte grid-stride read of the same bytes
//   y_copy           16 bytes read and 16 written per lane and trip
__global__ __launch_bounds__(1024) void y_read_records(const uint4* __restrict__ in, uint64_t n, uint64_t* sink) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4* p = in + i * 9;
#pragma unroll
        for (int k = 0; k < 7; k++) { const uint4 v = p[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x123456789abcdefull) *sink = acc;
}
__global__ __launch_bounds__(256) void y_read_stream(const uint4* __restrict__ in, uint64_t n16, uint64_t* sink) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x123456789abcdefull) *sink = acc;
}
__global__ __launch_bounds__(256) void y_copy(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}

extern "C" {

// which: 0 = y_read_records over `bytes` / 144 records at d_a, 1 = y_read_stream over bytes / 16 units at d_a, 2 = y_copy of bytes / 16 units
// from d_a to d_b. One untimed pass, then `reps` timed ones between two HIP events on the null stream. Returns the mean milliseconds
// per pass (< 0: a HIP error). d_sink: 8 bytes of device memory.
double nfagg_synth_yardstick(int which, const void* d_a, void* d_b, uint64_t bytes, void* d_sink, int reps) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
    auto pass = [&]() {
        if (which == 0) hipLaunchKernelGGL(y_read_records, dim3(256 * 2), dim3(1024), 0, 0, (const uint4*)d_a, bytes / 144, (uint64_t*)d_sink);
        else if (which == 1) hipLaunchKernelGGL(y_read_stream, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)d_a, bytes / 16, (uint64_t*)d_sink);
        else hipLaunchKernelGGL(y_copy, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)d_a, (uint4*)d_b, bytes / 16);
    };
    (void)hipGetLastError();
    pass();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) pass();
    hipEventRecord(e1, 0);
    float ms = -1.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) ms = -1.f;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms < 0.f ? -1.0 : (double)ms / (reps > 0 ? reps : 1);
}

// Fill d_out[0..n) (device, 16-byte aligned) with records j0..j0+n of the stream.
// d_thresholds: device table from nfagg_synth_zipf_thresholds (NULL = uniform);
// d_pop_index: optional device rank -> population-member table. stream: hipStream_t or NULL.
int nfagg_synth_stream(void* d_out, uint64_t n, uint64_t j0, uint64_t seed, uint64_t n_keys,
                       const uint64_t* d_thresholds, uint32_t hot_permille, uint32_t variant,
                       const uint64_t* d_pop_index, void* stream) {
    if (n == 0) return 0;
    uint64_t g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    (void)hipGetLastError(); hipLaunchKernelGGL(k_synth, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (uint4*)d_out, n, j0, seed, n_keys,
                       d_thresholds, hot_permille, variant, d_pop_index);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Host: thresholds[k] = floor(2^64 * CDF_zipf(k+1)), last = 2^64-1.
void nfagg_synth_zipf_thresholds(uint64_t n_keys, double s, uint64_t* out) {
    long double total = 0.0L;
    for (uint64_t k = 1; k <= n_keys; k++) total += powl((long double)k, -(long double)s);
    long double acc = 0.0L;
    const long double two64 = 18446744073709551616.0L;
    for (uint64_t k = 1; k <= n_keys; k++) {
        acc += powl((long double)k, -(long double)s);
        const long double v = acc / total * two64;
        out[k - 1] = (v >= two64) ? ~0ull : (uint64_t)v;
    }
    out[n_keys - 1] = ~0ull;
}

// Host: the first n_keys population members i (ascending) whose key hashes to
// `shard` of `n_shards`: the population of one GPU's shard.
void nfagg_synth_shard_population(uint64_t n_keys, uint32_t n_shards, uint32_t shard, uint64_t* out) {
    uint64_t found = 0;
    for (uint64_t i = 0; found < n_keys; i++) {
        R36 r;
        for (int k = 0; k < 36; k++) r.d[k] = 0;
        flow_id(r, i);
        uint64_t w[5];
        for (int k = 0; k < 5; k++) w[k] = (uint64_t)r.d[2 * k] | ((uint64_t)r.d[2 * k + 1] << 32);
        if (nfagg::shard_of_hash(nfagg::key_hash(w), n_shards) == shard) out[found++] = i;
    }
}

// Host mirror of the device generator (used to cross-check device output).
void nfagg_synth_stream_host(void* out, uint64_t n, uint64_t j0, uint64_t seed, uint64_t n_keys,
                             const uint64_t* thresholds, uint32_t hot_permille, uint32_t variant,
                             const uint64_t* pop_index) {
    for (uint64_t t = 0; t < n; t++) {
        const uint64_t j = j0 + t;
        uint64_t i = key_index(seed, j, n_keys, thresholds, hot_permille);
        if (pop_index) i = pop_index[i];
        R36 r;
        bench_record(r, i, j);
        if (variant == 1) scramble(r, seed, j);
        if (variant == 2) { put32(r, M + 44, (uint32_t)(2 + (j & 1))); put8(r, M + 56, (uint32_t)((j >> 1) & 1)); }  // configs[4]: two interfaces, both directions
        memcpy((char*)out + t * 144, r.d, 144);
    }
}

}  // extern "C"
