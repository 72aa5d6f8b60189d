// pmc_calib.hip — kernels with an exactly known HBM byte count, in the access patterns the
// ingest path uses, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced read; other widths and
// WRITE_SIZE must be calibrated on a known byte count in one's own pattern).
//   hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -- tools/pmc_calib ; rocprofv3 --pmc WRITE_SIZE -- tools/pmc_calib
// Buffers are 4.3 GB: far beyond the 256 MiB Infinity Cache, every byte comes from / goes to HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// 144-byte records, one per lane, the first 112 bytes as 16-byte loads (what part::k_fold pass 1 issues);
// the stream is contiguous, so all n*144 bytes cross the fabric.
__global__ __launch_bounds__(1024) void calib_read_records(const uint4* __restrict__ in, uint64_t n, uint64_t* sink) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4* p = in + i * 9;
#pragma unroll
        for (int k = 0; k < 7; k++) { const uint4 v = p[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x123456789abcdefull) *sink = acc;
}

// 144-byte records written one per lane as nine 16-byte stores (k_evict, k_synth).
__global__ __launch_bounds__(256) void calib_write_records(uint4* __restrict__ out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4* p = out + i * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) p[k] = make_uint4((uint32_t)i, k, 3, 4);
    }
}

// plain wide copy for reference: n16 x 16 bytes read and written.
__global__ __launch_bounds__(256) void calib_copy(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
    const uint64_t n = 30000000;            // records: 4.32 GB
    uint4 *a, *b; uint64_t* sink;
    CK(hipMalloc(&a, n * 144)); CK(hipMalloc(&b, n * 144)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 1, n * 144)); CK(hipMemset(b, 2, n * 144));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(calib_read_records, dim3(256), dim3(1024), 0, 0, a, n, sink);
        hipLaunchKernelGGL(calib_write_records, dim3(2048), dim3(256), 0, 0, b, n);
        hipLaunchKernelGGL(calib_copy, dim3(2048), dim3(256), 0, 0, a, b, n * 9);
        CK(hipDeviceSynchronize());
    }
    printf("calib_read_records: %llu bytes read\ncalib_write_records: %llu bytes written\ncalib_copy: %llu bytes read, %llu written\n",
           (unsigned long long)(n * 144), (unsigned long long)(n * 144), (unsigned long long)(n * 144), (unsigned long long)(n * 144));
    return 0;
}
