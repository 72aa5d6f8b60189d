// tools/ubench.hip — micro-benchmarks that decide the ingest kernel's structure on gfx950:
// throughput of agent-scope (cross-XCD coherent) atomics vs plain L2 atomics on random
// addresses, same-address serialisation, sc1 loads, and record-stream read shapes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

template <int SCOPE, bool RET, int PER>
__global__ __launch_bounds__(256) void k_atomic_rand(uint64_t* tab, uint64_t mask, uint64_t n, uint64_t* sink) {
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t a = (mix(i) & mask) & ~(uint64_t)(PER * 2 - 1);   // PER atomics inside one 128-B line when PER>1
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (RET) acc += __hip_atomic_fetch_add(&tab[a + k], 1ull, __ATOMIC_RELAXED, SCOPE);
            else __hip_atomic_fetch_add(&tab[a + k], 1ull, __ATOMIC_RELAXED, SCOPE);
        }
    }
    if (RET && acc == 0x1234567) *sink = acc;
}

template <int SCOPE>
__global__ __launch_bounds__(256) void k_atomic_same(uint64_t* tab, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        __hip_atomic_fetch_add(&tab[0], 1ull, __ATOMIC_RELAXED, SCOPE);
}

template <int SCOPE>
__global__ __launch_bounds__(256) void k_load_rand(const uint64_t* tab, uint64_t mask, uint64_t n, uint64_t* sink) {
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        acc += __hip_atomic_load(&tab[mix(i) & mask], __ATOMIC_RELAXED, SCOPE);
    if (acc == 0x1234567) *sink = acc;
}

__global__ __launch_bounds__(256) void k_stream(const uint4* in, uint64_t n16, uint64_t* sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x1234567) *sink = acc;
}

// one 144-byte record per lane, 9 strided 16-byte loads
__global__ __launch_bounds__(256) void k_rec_strided(const uint4* in, uint64_t n, uint64_t* sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4* p = in + i * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) { uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x1234567) *sink = acc;
}

// the same records, read coalesced by the workgroup into LDS, then each lane reads its record from LDS
__global__ __launch_bounds__(256) void k_rec_lds(const uint4* in, uint64_t n, uint64_t* sink) {
    __shared__ uint4 tile[256 * 9];
    uint32_t acc = 0;
    const uint64_t tiles = n / 256;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint4* p = in + t * 256 * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) tile[k * 256 + threadIdx.x] = p[k * 256 + threadIdx.x];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; k++) { uint4 v = tile[threadIdx.x * 9 + k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        __syncthreads();
    }
    if (acc == 0x1234567) *sink = acc;
}

template <typename F>
static double time_ms(F f, int reps = 3) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

#define AG __HIP_MEMORY_SCOPE_AGENT
#define WG __HIP_MEMORY_SCOPE_WORKGROUP

int main() {
    const uint64_t max_words = 1ull << 28;   // 2 GiB
    uint64_t *tab, *sink;
    CK(hipMalloc(&tab, max_words * 8)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(tab, 0, max_words * 8));
    const uint64_t n = 1ull << 27;
    const dim3 g(2048), b(256);
    printf("# random-address atomic add u64, %llu ops per launch\n", (unsigned long long)n);
    for (uint64_t words : {1ull << 21, 1ull << 25, 1ull << 28}) {
        const uint64_t mask = words - 1;
        double t;
        t = time_ms([&] { hipLaunchKernelGGL((k_atomic_rand<AG, false, 1>), g, b, 0, 0, tab, mask, n, sink); });
        printf("table %5llu MiB  agent noret : %7.2f G atomics/s\n", (unsigned long long)(words * 8 >> 20), n / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((k_atomic_rand<AG, true, 1>), g, b, 0, 0, tab, mask, n, sink); });
        printf("table %5llu MiB  agent ret   : %7.2f G atomics/s\n", (unsigned long long)(words * 8 >> 20), n / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((k_atomic_rand<WG, false, 1>), g, b, 0, 0, tab, mask, n, sink); });
        printf("table %5llu MiB  wg    noret : %7.2f G atomics/s (not cross-XCD coherent)\n", (unsigned long long)(words * 8 >> 20), n / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((k_atomic_rand<AG, false, 4>), g, b, 0, 0, tab, mask, n / 4, sink); });
        printf("table %5llu MiB  agent noret, 4 per line : %7.2f G atomics/s (%.2f G lines/s)\n", (unsigned long long)(words * 8 >> 20), n / t / 1e6, n / 4 / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((k_atomic_rand<AG, false, 8>), g, b, 0, 0, tab, mask, n / 8, sink); });
        printf("table %5llu MiB  agent noret, 8 per line : %7.2f G atomics/s (%.2f G lines/s)\n", (unsigned long long)(words * 8 >> 20), n / t / 1e6, n / 8 / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((k_load_rand<AG>), g, b, 0, 0, tab, mask, n, sink); });
        printf("table %5llu MiB  agent load 8B : %7.2f G loads/s\n", (unsigned long long)(words * 8 >> 20), n / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((k_load_rand<WG>), g, b, 0, 0, tab, mask, n, sink); });
        printf("table %5llu MiB  plain load 8B : %7.2f G loads/s\n", (unsigned long long)(words * 8 >> 20), n / t / 1e6);
    }
    {
        const uint64_t m = 1ull << 22;
        double t = time_ms([&] { hipLaunchKernelGGL((k_atomic_same<AG>), g, b, 0, 0, tab, m); });
        printf("same address agent : %7.2f M atomics/s (%.1f ns each)\n", m / t / 1e3, t * 1e6 / m);
        t = time_ms([&] { hipLaunchKernelGGL((k_atomic_same<WG>), g, b, 0, 0, tab, m); });
        printf("same address wg    : %7.2f M atomics/s\n", m / t / 1e3);
    }
    {
        const uint64_t bytes = max_words * 8;
        double t = time_ms([&] { hipLaunchKernelGGL(k_stream, g, b, 0, 0, (const uint4*)tab, bytes / 16, sink); });
        printf("stream read 2 GiB uint4      : %7.1f GB/s\n", bytes / t / 1e6);
        const uint64_t nrec = bytes / 144 / 256 * 256;
        t = time_ms([&] { hipLaunchKernelGGL(k_rec_strided, g, b, 0, 0, (const uint4*)tab, nrec, sink); });
        printf("144-B records, lane-strided  : %7.1f GB/s (%.2f G rec/s)\n", nrec * 144 / t / 1e6, nrec / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL(k_rec_lds, g, b, 0, 0, (const uint4*)tab, nrec, sink); });
        printf("144-B records, LDS-staged    : %7.1f GB/s (%.2f G rec/s)\n", nrec * 144 / t / 1e6, nrec / t / 1e6);
    }
    return 0;
}
