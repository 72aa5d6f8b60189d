// Probe (round 5): k_par_hash -> sort -> k_par_links of csrc/nfagg_epoch_par.hip on synthetic records, prev[] checked on the host.
// Build: hipcc --offload-arch=gfx950 -O2 -I netobserv-ebpf-agent_amd/csrc -o tools/gpu/links_probe tools/gpu/links_probe.hip
#include "../../netobserv-ebpf-agent_amd/csrc/nfagg_epoch_par.hip"
#include <cstdio>
#include <vector>
#include <unordered_map>
using namespace nfagg;
static uint64_t mixh(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; return x ^ (x >> 33); }
int main() {
    for (size_t n : {600000ul, 8000000ul}) {
        const unsigned flows = 100000;
        std::vector<uint8_t> recs(n * 144, 0);
        std::vector<uint32_t> flow_of(n);
        for (size_t i = 0; i < n; i++) {
            const uint32_t f = (uint32_t)(mixh(i * 7919 + 1) % flows) / ((mixh(i) & 3) + 1);      // skewed
            flow_of[i] = f;
            uint64_t w[5] = {mixh(f + 1), mixh(f + 77), f, ~(uint64_t)f, f * 3ull};
            w[4] &= 0x00FFFFFFFFFFFFFFull;
            memcpy(&recs[i * 144], w, 40);
            recs[i * 144 + 39] = (uint8_t)i;                                                     // Go's blank byte: not part of the key
        }
        void *d; uint64_t *k0, *k1, *k2; int32_t* prev; uint32_t* ctl; void* tmp; size_t tb = 0;
        hipMalloc(&d, n * 144); hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&k2, n * 8); hipMalloc(&prev, n * 4); hipMalloc(&ctl, 4096);
        hipMemcpy(d, recs.data(), n * 144, hipMemcpyHostToDevice); hipMemset(ctl, 0, 4096);
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        hipError_t e = launch_par_sort(nullptr, &tb, k0, k1, n, s);
        hipMalloc(&tmp, tb + 16);
        if (e == hipSuccess) e = launch_par_hash(d, n, k0, s);
        hipError_t e1 = hipStreamSynchronize(s);
        if (e == hipSuccess) e = launch_par_sort(tmp, &tb, k0, k1, n, s);
        hipError_t e2 = hipStreamSynchronize(s);
        std::vector<uint64_t> ks(n);
        hipMemcpy(ks.data(), k1, n * 8, hipMemcpyDeviceToHost);
        size_t unsorted = 0, badidx = 0;
        for (size_t i = 1; i < n; i++) unsorted += ks[i] <= ks[i - 1];
        for (size_t i = 0; i < n; i++) badidx += (ks[i] & 0xFFFFFF) >= n;
        printf("n %zu: launch %d, sync after hash %d, after sort %d; sorted keys: %zu out of order, %zu indices out of range\n", n, (int)e, (int)e1, (int)e2, unsorted, badidx);
        fflush(stdout);
        if (e == hipSuccess) e = launch_par_links(d, k1, n, prev, ctl + 1, s);
        hipError_t e3 = hipStreamSynchronize(s);
        std::vector<int32_t> got(n);
        hipMemcpy(got.data(), prev, n * 4, hipMemcpyDeviceToHost);
        std::unordered_map<uint32_t, int32_t> last;
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) { auto it = last.find(flow_of[i]); const int32_t want = it == last.end() ? -1 : it->second; bad += got[i] != want; last[flow_of[i]] = (int32_t)i; }
        printf("n %zu: links launch %d sync %d: %zu of %zu prev[] wrong\n", n, (int)e, (int)e3, bad, n);
        fflush(stdout);
    }
    return 0;
}
