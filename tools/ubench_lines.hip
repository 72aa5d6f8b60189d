// tools/ubench_lines.hip — what the memory system of an MI355X gives RANDOM 128-byte-line accesses, the access pattern of every
// kernel of the fold that is not the record stream itself: pass 2's index gather (112 bytes of a 144-byte record at a random
// index: 1.75 lines), the cache flushes (home slot line read, slot lines written), k_finalize and k_evict (slot lines by
// live-list order = random). DESIGN.md §9 compares the kernels' line rates with these ceilings.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lines.hip -o tools/ubench_lines     Run: tools/ubench_lines [GiB of buffer]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// the record stream: 16 bytes per lane, consecutive
__global__ __launch_bounds__(256) void k_stream(const uint4* in, uint64_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) acc ^= fold(in[i]);
    if (acc == 0x1234567) *sink = acc;
}

// UNITS x 16 bytes of a 144-byte record at a random index per lane; PER records in flight per lane (pass 2: UNITS = 7, PER = 1 + the
// software pipeline's next tile)
template <int UNITS, int PER>
__global__ __launch_bounds__(1024) void k_gather_record(const uint4* in, uint64_t n_rec, uint64_t n_acc, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * PER;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * PER; i < n_acc; i += stride) {
        uint4 v[PER][UNITS];
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const uint4* r = in + (mix(i + p) % n_rec) * 9;
#pragma unroll
            for (int k = 0; k < UNITS; k++) v[p][k] = r[k];
        }
#pragma unroll
        for (int p = 0; p < PER; p++)
#pragma unroll
            for (int k = 0; k < UNITS; k++) acc ^= fold(v[p][k]);
    }
    if (acc == 0x1234567) *sink = acc;
}

// one 16-byte load in a random 128-byte line per lane (the least a lane can ask of a line)
template <int PER>
__global__ __launch_bounds__(1024) void k_line_touch(const uint4* in, uint64_t n_lines, uint64_t n_acc, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * PER;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * PER; i < n_acc; i += stride) {
        uint4 v[PER];
#pragma unroll
        for (int p = 0; p < PER; p++) v[p] = in[(mix(i + p) % n_lines) * 8];
#pragma unroll
        for (int p = 0; p < PER; p++) acc ^= fold(v[p]);
    }
    if (acc == 0x1234567) *sink = acc;
}

// a whole random 128-byte line by eight lanes (k_evict's cooperative fetch of a slot's hot line); WRITE: and written back
template <bool WRITE>
__global__ __launch_bounds__(1024) void k_line_whole(uint4* in, uint64_t n_lines, uint64_t n_acc, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t lane8 = threadIdx.x & 7;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n_acc; i += stride) {
        uint4* p = in + (mix(i) % n_lines) * 8 + lane8;
        uint4 v = *p;
        acc ^= fold(v);
        if (WRITE) { v.x += 1; *p = v; }
    }
    if (acc == 0x1234567) *sink = acc;
}


// ---- pass 2's own pattern: workgroup b gathers the records of "partition" b in ASCENDING index order (one record out of every
// `gap`, jittered): every workgroup sweeps the whole batch once, all of them at about the same pace.
__device__ __forceinline__ uint64_t sweep_index(uint64_t k, uint64_t b, uint64_t gap, uint64_t n_rec) {
    const uint64_t i = k * gap + (mix(k * 4099 + b) % gap);
    return i < n_rec ? i : n_rec - 1;
}

// one record per lane, UNITS strided 16-byte loads (what pass2_round does)
template <int UNITS>
__global__ __launch_bounds__(1024) void k_sweep_per_lane(const uint4* in, uint64_t n_rec, uint64_t per_block, uint64_t gap, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t k = threadIdx.x; k < per_block; k += blockDim.x) {
        const uint4* r = in + sweep_index(k, blockIdx.x, gap, n_rec) * 9;
        uint4 v[UNITS];
#pragma unroll
        for (int u = 0; u < UNITS; u++) v[u] = r[u];
#pragma unroll
        for (int u = 0; u < UNITS; u++) acc ^= fold(v[u]);
    }
    if (acc == 0x1234567) *sink = acc;
}

// eight lanes per record: lane u of a group loads unit u (and unit 8 by lane 0 when UNITS = 9): consecutive 16-byte units of
// one record by consecutive lanes; LDS_TRANSPOSE: the units go through LDS so that one lane ends up with its record's units
template <int UNITS, bool LDS_TRANSPOSE>
__global__ __launch_bounds__(1024) void k_sweep_coop(const uint4* in, uint64_t n_rec, uint64_t per_block, uint64_t gap, uint32_t* sink) {
    __shared__ uint4 T[LDS_TRANSPOSE ? 1024 * 9 : 1];
    uint32_t acc = 0;
    const int u = threadIdx.x & 7, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint64_t k0 = 0; k0 < per_block; k0 += blockDim.x) {
        // the wave's 64 records of this tile, eight per instruction
        uint4 v[8], v8 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t k = k0 + (uint64_t)wv * 64 + j * 8 + (lane >> 3);
            const bool ok = k < per_block;
            const uint4* r = in + sweep_index(ok ? k : 0, blockIdx.x, gap, n_rec) * 9;
            v[j] = (ok && u < UNITS) ? r[u] : make_uint4(0, 0, 0, 0);
            if (UNITS == 9 && j == u) { }      // (unit 8: below)
        }
        if (UNITS == 9) {                       // the ninth unit of the wave's 64 records: one per lane
            const uint64_t k = k0 + (uint64_t)wv * 64 + lane;
            if (k < per_block) v8 = in[sweep_index(k, blockIdx.x, gap, n_rec) * 9 + 8];
        }
        if (LDS_TRANSPOSE) {
#pragma unroll
            for (int j = 0; j < 8; j++) T[(wv * 64 + j * 8 + (lane >> 3)) * 9 + u] = v[j];
            __builtin_amdgcn_wave_barrier();
            uint4 w[8];
#pragma unroll
            for (int q = 0; q < 8; q++) w[q] = T[(wv * 64 + lane) * 9 + q];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 8; q++) acc ^= fold(w[q]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= fold(v[j]);
        }
        acc ^= fold(v8);
    }
    if (acc == 0x1234567) *sink = acc;
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 13.4;            // 14.4 GB = the 100 M-record batch
    const uint64_t bytes = ((uint64_t)(gib * (1ull << 30)) / 1152) * 1152;      // whole records and whole lines
    const uint64_t n16 = bytes / 16, n_rec = bytes / 144, n_lines = bytes / 128;
    uint4* buf; uint32_t* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    const uint64_t n_acc = 35000000;                               // the spilled records of a configs[1] call
    printf("buffer %.2f GB (%llu records of 144 bytes, %llu lines of 128 bytes); %llu random accesses per launch\n", bytes / 1e9,
           (unsigned long long)n_rec, (unsigned long long)n_lines, (unsigned long long)n_acc);
    double ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, buf, n16, sink); }, 3);
    printf("stream read, 16 bytes per lane                     : %7.3f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
#define GATHER(U, P, LINES) do { \
        ms = time_ms([&] { hipLaunchKernelGGL((k_gather_record<U, P>), dim3(2048), dim3(1024), 0, 0, buf, n_rec, n_acc, sink); }, 3); \
        printf("gather %3d bytes of a random record, %d in flight     : %7.3f ms  %6.2f G records/s  %6.2f G lines/s (%.3f lines each)  %5.2f TB/s of lines\n", \
               16 * U, P, ms, n_acc / ms / 1e6, n_acc * LINES / ms / 1e6, LINES, n_acc * LINES * 128 / ms / 1e9); } while (0)
    // lines touched by [16 a, 16 a + 16 U) for a = 0..7 (a record starts at a multiple of 16 bytes inside a line; 144 = 128 + 16)
    GATHER(7, 1, 1.75); GATHER(7, 2, 1.75); GATHER(7, 4, 1.75);
    GATHER(9, 1, 2.0); GATHER(9, 2, 2.0);
    GATHER(3, 2, 1.25); GATHER(1, 4, 1.0);
#define TOUCH(P) do { \
        ms = time_ms([&] { hipLaunchKernelGGL((k_line_touch<P>), dim3(2048), dim3(1024), 0, 0, buf, n_lines, n_acc, sink); }, 3); \
        printf("16 bytes of a random line, %d in flight                : %7.3f ms  %6.2f G lines/s  %5.2f TB/s of lines\n", P, ms, n_acc / ms / 1e6, n_acc * 128 / ms / 1e9); } while (0)
    TOUCH(1); TOUCH(4); TOUCH(8);
    ms = time_ms([&] { hipLaunchKernelGGL((k_line_whole<false>), dim3(2048), dim3(1024), 0, 0, buf, n_lines, n_acc, sink); }, 3);
    printf("a whole random line by eight lanes, read             : %7.3f ms  %6.2f G lines/s  %5.2f TB/s\n", ms, n_acc / ms / 1e6, n_acc * 128 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_line_whole<true>), dim3(2048), dim3(1024), 0, 0, buf, n_lines, n_acc, sink); }, 3);
    printf("a whole random line by eight lanes, read and written : %7.3f ms  %6.2f G lines/s  %5.2f TB/s (read + write)\n", ms, n_acc / ms / 1e6, 2.0 * n_acc * 128 / ms / 1e9);

    // pass 2's pattern: 2048 partitions x 17 090 records each (35 M), each partition sweeping the batch
    {
        const uint64_t parts = 2048, per_block = n_acc / parts, gap = n_rec / per_block;
        printf("pass 2's pattern: %llu workgroups x %llu records in ascending order (one of every %llu)\n", (unsigned long long)parts, (unsigned long long)per_block, (unsigned long long)gap);
#define SWEEP(NAME, KERNEL, LINES) do { \
            ms = time_ms([&] { hipLaunchKernelGGL(KERNEL, dim3(parts), dim3(1024), 0, 0, buf, n_rec, per_block, gap, sink); }, 3); \
            printf("  %-58s: %7.3f ms  %6.2f G records/s  %6.2f G lines/s\n", NAME, ms, parts * per_block / ms / 1e6, parts * per_block * LINES / ms / 1e6); } while (0)
        SWEEP("112 bytes, one record per lane (7 strided loads)", (k_sweep_per_lane<7>), 1.75);
        SWEEP("112 bytes, eight lanes per record", (k_sweep_coop<7, false>), 1.75);
        SWEEP("112 bytes, eight lanes per record + LDS transpose", (k_sweep_coop<7, true>), 1.75);
        SWEEP("128 bytes, eight lanes per record + LDS transpose", (k_sweep_coop<8, true>), 1.875);
        SWEEP("144 bytes, one record per lane (9 strided loads)", (k_sweep_per_lane<9>), 2.0);
        SWEEP("144 bytes, eight lanes per record + one, LDS transpose", (k_sweep_coop<9, true>), 2.0);
    }
    return 0;
}
