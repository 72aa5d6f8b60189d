// Minimal form of a hipcc (ROCm 7.2, gfx950) miscompile met in round 5: (key & 0xFFFFFF) * 144 is taken for a 24-bit multiply, the mask is
// dropped as not demanded, and v_mad_u64_u32 — which reads all 32 bits — is formed from it. hipcc --offload-arch=gfx950 -O3 -c; objdump:
//     v_mad_u64_u32 v[10:11], s[16:17], v8, s18, v[10:11]      <- address from the UNMASKED low dword of the key
//     v_and_b32_e32 v8, 0xffffff, v8                            <- the mask, afterwards (for the value that is stored)
// csrc/nfagg_epoch_par.hip key_index() keeps the mask with an opaque register; tests/test_isa_pins.py looks for this signature.
#include <hip/hip_runtime.h>
#include <stdint.h>
// minimal form of what k_par_links did: index = key & 0xFFFFFF, address = base + index * 144
__global__ void k(const uint64_t* __restrict__ ks, const char* __restrict__ recs, uint64_t n, uint64_t* out) {
    const uint64_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0 || p >= n) return;
    uint64_t q = p, acc = 0;
    const uint64_t hb = ks[p] & ~0xFFFFFFull;
    while (q > 0) {
        const uint64_t k2 = ks[--q];
        if ((k2 & ~0xFFFFFFull) != hb) break;
        const uint4* r = reinterpret_cast<const uint4*>(recs + (k2 & 0xFFFFFFull) * 144);
        const uint4 a = r[0];
        if (a.x == 7) { acc = (uint32_t)(k2 & 0xFFFFFFull); break; }
    }
    out[p] = acc;
}
