// tools/gpu/gather_flavours.hip — round 5 probe for pass 2's index gather: does ANY load flavour of gfx950 fetch less
// than two whole 128-byte lines for the 112-byte head of a 144-byte record at a random index? One lane per record, seven 16-byte loads
// (bytes 0..111), 35 M random records out of 100 M (14.4 GB): the gather of the headline call. Flavours = the cache-policy bits of
// global_load (sc0, sc1, nt and their combinations) and a six-load form that leaves out the seventh unit.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gpu/gather_flavours.hip -o tools/gpu/gather_flavours
// Run:   tools/gpu/gather_flavours            (times, HIP events)
//        rocprofv3 --pmc FETCH_SIZE -d out -- tools/gpu/gather_flavours     (bytes per kernel; x2 as calibrated on this part)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

#define LD(FL) asm volatile("global_load_dwordx4 %0, %1, off " FL : "=v"(v[k]) : "v"(p + k) : "memory")

template <int FLAV, int UNITS>
__global__ __launch_bounds__(1024) void k_gather(const uint4* in, uint64_t n_rec, uint64_t n_acc, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_acc; i += stride) {
        const uint4* p = in + (mix(i) % n_rec) * 9;
        uint4 v[7];
#pragma unroll
        for (int k = 0; k < UNITS; k++) {
            if (FLAV == 0) LD("");
            else if (FLAV == 1) LD("nt");
            else if (FLAV == 2) LD("sc0");
            else if (FLAV == 3) LD("sc1");
            else if (FLAV == 4) LD("sc0 sc1");
            else if (FLAV == 5) LD("sc0 nt");
            else if (FLAV == 6) LD("sc1 nt");
            else LD("sc0 sc1 nt");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < UNITS; k++) acc ^= fold(v[k]);
    }
    if (acc == 0x1234567) *sink = acc;
}

template <int FLAV, int UNITS>
static void run(const char* name, const uint4* d, uint64_t n_rec, uint64_t n_acc, uint32_t* sink) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_gather<FLAV, UNITS>), dim3(1024), dim3(1024), 0, 0, d, n_rec, n_acc, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-22s %d units  %7.3f ms   %6.1f G records/s   needed bytes at %5.2f TB/s\n", name, UNITS, best, n_acc / best / 1e6, n_acc * 16.0 * UNITS / best / 1e9);
}

int main() {
    const uint64_t n_rec = 100000000ull, n_acc = 35000000ull;
    uint4* d; uint32_t* sink;
    CK(hipMalloc(&d, n_rec * 144)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 1, n_rec * 144)); CK(hipDeviceSynchronize());
    run<0, 7>("plain", d, n_rec, n_acc, sink);
    run<1, 7>("nt", d, n_rec, n_acc, sink);
    run<2, 7>("sc0", d, n_rec, n_acc, sink);
    run<3, 7>("sc1", d, n_rec, n_acc, sink);
    run<4, 7>("sc0 sc1", d, n_rec, n_acc, sink);
    run<5, 7>("sc0 nt", d, n_rec, n_acc, sink);
    run<6, 7>("sc1 nt", d, n_rec, n_acc, sink);
    run<7, 7>("sc0 sc1 nt", d, n_rec, n_acc, sink);
    run<0, 6>("plain, six units", d, n_rec, n_acc, sink);
    run<1, 6>("nt, six units", d, n_rec, n_acc, sink);
    return 0;
}
