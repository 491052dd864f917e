// xxh64.h — XXH64 (public xxHash specification), streaming form specialised for the block
// hash of BlockManager::compute_hash (reference src/engine/block_manager.rs:109-123): the byte
// string is [prefix_hash u64 LE]? ++ token ids as i64 LE, i.e. a sequence of 64-bit words, so the
// hash is computed word-wise with no staging buffer (the reference builds a Vec<u8> per call).
#pragma once
#include <cstddef>
#include <cstdint>

namespace nvr {

struct Xxh64Words {
    static constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL,
                              P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL,
                              P5 = 0x27D4EB2F165667C5ULL;
    static inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
    static inline uint64_t lane(uint64_t acc, uint64_t w) { return rotl(acc + w * P2, 31) * P1; }
    static inline uint64_t fold(uint64_t h, uint64_t v) { return (h ^ lane(0, v)) * P1 + P4; }

    // hash of `nwords` little-endian 64-bit words, word i supplied by get(i); seed 0
    template <class Get>
    static uint64_t hash(size_t nwords, Get get) {
        const uint64_t seed = 0;
        uint64_t h;
        size_t i = 0;
        if (nwords >= 4) {
            uint64_t v[4] = {seed + P1 + P2, seed + P2, seed, seed - P1};
            for (; i + 4 <= nwords; i += 4)
                for (int j = 0; j < 4; ++j) v[j] = lane(v[j], get(i + j));
            h = rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18);
            for (int j = 0; j < 4; ++j) h = fold(h, v[j]);
        } else {
            h = seed + P5;
        }
        h += (uint64_t)nwords * 8;
        for (; i < nwords; ++i) h = rotl(h ^ lane(0, get(i)), 27) * P1 + P4;
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        return h;
    }
};

inline uint64_t block_hash(const int64_t *tokens, size_t n, bool has_prefix, uint64_t prefix) {
    if (has_prefix)
        return Xxh64Words::hash(n + 1, [&](size_t i) { return i == 0 ? prefix : (uint64_t)tokens[i - 1]; });
    return Xxh64Words::hash(n, [&](size_t i) { return (uint64_t)tokens[i]; });
}

}  // namespace nvr
