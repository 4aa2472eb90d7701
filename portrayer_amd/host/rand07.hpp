// rand 0.7.0 `StdRng` as the reference's scene scripts use it (examples/big-scene.rs:27-67):
// `StdRng::seed_from_u64`, `gen::<f64>()`, `SliceRandom::choose`, `SliceRandom::shuffle` (examples/graphics-castle.rs:452).
//
// rand / rand_chacha / rand_core are third-party crates (Cargo.lock pins rand 0.7.0, rand_chacha
// 0.2.0, rand_core 0.5.0); this follows their published algorithms: seed_from_u64 fills the 32-byte
// seed with eight PCG32 outputs; StdRng is ChaCha with 20 rounds, a 64-bit block counter in state
// words 12-13 and stream 0; the output is the key-stream as little-endian 32-bit words.
#pragma once

#include <array>
#include <cstdint>
#include <utility>
#include <vector>

namespace portrayer {
namespace rand07 {

class StdRng {
   public:
    static StdRng seed_from_u64(uint64_t state) {
        const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
        StdRng r;
        for (int i = 0; i < 8; i++) {
            state = state * MUL + INC;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            r.key_[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
        return r;
    }
    uint32_t next_u32() {
        if (pos_ == 16) refill();
        return buf_[pos_++];
    }
    uint64_t next_u64() {
        uint64_t lo = next_u32();
        uint64_t hi = next_u32();
        return lo | (hi << 32);
    }
    double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }  // Standard: 53 random bits
    // SliceRandom::choose -> gen_range(0, len as u32): widening-multiply rejection sampling
    uint32_t gen_index(uint32_t len) {
        int lz = __builtin_clz(len);
        uint32_t zone = (len << lz) - 1u;
        for (;;) {
            uint64_t m = (uint64_t)next_u32() * (uint64_t)len;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }
    template <class T>
    const T& choose(const std::vector<T>& v) { return v[gen_index((uint32_t)v.size())]; }
    // SliceRandom::shuffle (rand 0.7.0 seq/mod.rs): for i in (1..len).rev() { swap(i, gen_index(rng, i + 1)) }
    template <class T, size_t N>
    void shuffle(std::array<T, N>& a) {
        for (size_t i = N - 1; i >= 1; i--) std::swap(a[i], a[gen_index((uint32_t)(i + 1))]);
    }
    const uint32_t* key() const { return key_; }

   private:
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    static void qr(uint32_t* s, int a, int b, int c, int d) {
        s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
        s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
        s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
        s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
    }
    void refill() {
        uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) init[4 + i] = key_[i];
        init[12] = (uint32_t)counter_; init[13] = (uint32_t)(counter_ >> 32); init[14] = 0; init[15] = 0;
        uint32_t s[16];
        for (int i = 0; i < 16; i++) s[i] = init[i];
        for (int r = 0; r < 10; r++) {
            qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15);
            qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) buf_[i] = s[i] + init[i];
        counter_++;
        pos_ = 0;
    }
    uint32_t key_[8] = {0};
    uint64_t counter_ = 0;
    uint32_t buf_[16];
    int pos_ = 16;
};

}  // namespace rand07
}  // namespace portrayer
