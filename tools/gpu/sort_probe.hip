// Probe (round 5): rocprim::radix_sort_keys on 64-bit keys = (40 hash bits << 24 | index) over partial bit ranges, default config and
// the one-sweep-only config of csrc/nfagg_epoch_par.hip. Build: hipcc --offload-arch=gfx950 -O2 -o tools/gpu/sort_probe tools/gpu/sort_probe.hip
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
using OneSweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; return x ^ (x >> 33); }
template <class Config>
static int run(const char* name, size_t n, unsigned begin, unsigned end, unsigned flows) {
    std::vector<uint64_t> h(n), want(n), got(n);
    for (size_t i = 0; i < n; i++) h[i] = (mix(i % flows + 1) & ~0xFFFFFFull) | i;
    want = h;
    const uint64_t m = (end == 64 ? ~0ull : ((1ull << end) - 1)) & ~((1ull << begin) - 1);
    std::stable_sort(want.begin(), want.end(), [m](uint64_t a, uint64_t b) { return (a & m) < (b & m); });
    uint64_t *in, *out; void* tmp = nullptr; size_t bytes = 0;
    hipMalloc(&in, n * 8); hipMalloc(&out, n * 8);
    hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipError_t e = rocprim::radix_sort_keys<Config>(nullptr, bytes, (const uint64_t*)in, out, n, begin, end, s);
    hipMalloc(&tmp, bytes + 16);
    if (e == hipSuccess) e = rocprim::radix_sort_keys<Config>(tmp, bytes, (const uint64_t*)in, out, n, begin, end, s);
    hipError_t e2 = hipStreamSynchronize(s);
    hipMemcpy(got.data(), out, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < n; i++) bad += got[i] != want[i];
    printf("%-10s n %9zu bits [%2u, %2u) flows %8u: launch %d sync %d temp %zu bytes, %zu positions differ from std::stable_sort\n", name, n, begin, end, flows, (int)e, (int)e2, bytes, bad);
    fflush(stdout);
    hipFree(in); hipFree(out); hipFree(tmp); hipStreamDestroy(s);
    return bad != 0;
}
int main() {
    int bad = 0;
    for (size_t n : {600000ul, 2000000ul, 8000000ul}) {
        for (unsigned flows : {1000u, 1000000u}) {
            bad += run<rocprim::default_config>("default", n, 32, 64, flows);
            bad += run<OneSweep>("one-sweep", n, 32, 64, flows);
            bad += run<OneSweep>("one-sweep", n, 24, 32, flows);
            bad += run<OneSweep>("one-sweep", n, 24, 64, flows);
            bad += run<OneSweep>("one-sweep", n, 0, 64, flows);
            bad += run<rocprim::default_config>("default", n, 24, 64, flows);
        }
    }
    return bad ? 1 : 0;
}
