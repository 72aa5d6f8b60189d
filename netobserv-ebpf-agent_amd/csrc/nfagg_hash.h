// nfagg_hash.h — the key / IP hash specification of libnfagg (host + device).
//
// One 64-bit hash of the 40 key bytes (bpf/types.h:191-204 flow_id, byte 39
// forced to zero) drives everything: table slot (low bits), fingerprint (high
// 62 bits), shard (multiply-shift of the high 32 bits). Sketches hash the
// 16-byte IPs with four fixed seeds. The spec is frozen in DESIGN.md
// §"Hash specification"; oracle/ restates it independently.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NF_HD __host__ __device__ __forceinline__
#else
#define NF_HD inline
#endif

namespace nfagg {

constexpr uint64_t kMul = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kKeySeed = 0x6E66616767206B31ull;  // "nfagg k1"

NF_HD uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

NF_HD uint64_t fmix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// w[0..5): the key as five little-endian 64-bit words, byte 39 already zero.
NF_HD uint64_t key_hash(const uint64_t w[5]) {
    uint64_t h = kKeySeed;
#pragma unroll
    for (int i = 0; i < 5; i++) h = (rotl64(h, 27) ^ w[i]) * kMul;
    return fmix64(h);
}

NF_HD uint64_t ip_seed(uint32_t i) {
    switch (i & 3) {
        case 0: return 0x243F6A8885A308D3ull;
        case 1: return 0x13198A2E03707344ull;
        case 2: return 0xA4093822299F31D0ull;
        default: return 0x082EFA98EC4E6C89ull;
    }
}

// ip as two little-endian 64-bit words.
NF_HD uint64_t ip_hash(uint64_t lo, uint64_t hi, uint32_t seed_index) {
    uint64_t h = ip_seed(seed_index);
    h = (rotl64(h, 27) ^ lo) * kMul;
    h = (rotl64(h, 27) ^ hi) * kMul;
    return fmix64(h);
}

NF_HD uint32_t shard_of_hash(uint64_t h, uint32_t n_shards) {
    if (n_shards <= 1) return 0;
    return (uint32_t)(((h >> 32) * (uint64_t)n_shards) >> 32);
}

// Count-Min row index: Kirsch–Mitzenmacher double hashing on two IP hashes,
// top log2w bits.
NF_HD uint64_t cm_index(uint64_t ha, uint64_t hb_odd, uint32_t row, uint32_t log2w) {
    return (ha + (uint64_t)row * hb_odd) >> (64 - log2w);
}

}  // namespace nfagg
