"""rand 0.7.0 `StdRng` restated for the tests (test infrastructure).

examples/big-scene.rs:27-67 builds its scene from `StdRng::seed_from_u64(1234939301)`;
rand / rand_chacha / rand_core are NOT under /root/reference (Cargo.lock pins rand 0.7.0,
rand_chacha 0.2.0, rand_core 0.5.0), so this follows their published algorithms (SURVEY App.B.3):
  * seed_from_u64: eight PCG32 outputs fill the 32-byte seed (rand_core 0.5),
  * StdRng = ChaCha, 20 rounds, 64-bit block counter in words 12-13, stream 0, output = the
    key-stream as little-endian u32 words in order,
  * gen::<f64>() = (next_u64() >> 11) * 2^-53, next_u64 = lo | hi << 32 of consecutive words,
  * slice.choose = widening-multiply rejection sampling on next_u32.
Pinned by the known answers of SURVEY §8(c)6 (tests/test_rand07.py) and, at image level, by
render/09a_kdtree.png.
"""
from __future__ import annotations

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def _rotl(x, n):
    return ((x << n) & M32) | (x >> (32 - n))


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & M32; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & M32; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & M32; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & M32; s[b] = _rotl(s[b] ^ s[c], 7)


class StdRng:
    def __init__(self, key_words):
        self.key = list(key_words)
        self.counter = 0
        self.buf = []

    @staticmethod
    def seed_from_u64(state: int) -> "StdRng":
        MUL, INC = 6364136223846793005, 11634580027462260723
        words = []
        for _ in range(8):
            state = (state * MUL + INC) & M64
            xorshifted = (((state >> 18) ^ state) >> 27) & M32
            rot = state >> 59
            words.append(((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32)
        return StdRng(words)

    def _block(self):
        init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + self.key + \
               [self.counter & M32, (self.counter >> 32) & M32, 0, 0]
        s = list(init)
        for _ in range(10):
            _qr(s, 0, 4, 8, 12); _qr(s, 1, 5, 9, 13); _qr(s, 2, 6, 10, 14); _qr(s, 3, 7, 11, 15)
            _qr(s, 0, 5, 10, 15); _qr(s, 1, 6, 11, 12); _qr(s, 2, 7, 8, 13); _qr(s, 3, 4, 9, 14)
        self.counter += 1
        return [(a + b) & M32 for a, b in zip(s, init)]

    def next_u32(self) -> int:
        if not self.buf:
            self.buf = self._block()
        return self.buf.pop(0)

    def next_u64(self) -> int:
        lo = self.next_u32()
        hi = self.next_u32()
        return lo | (hi << 32)

    def gen_f64(self) -> float:
        return float(self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    def gen_index(self, n: int) -> int:
        lz = 32 - n.bit_length()
        zone = ((n << lz) & M32) - 1 & M32
        while True:
            v = self.next_u32()
            m = v * n
            hi, lo = m >> 32, m & M32
            if lo <= zone:
                return hi

    def choose(self, seq):
        return seq[self.gen_index(len(seq))]
