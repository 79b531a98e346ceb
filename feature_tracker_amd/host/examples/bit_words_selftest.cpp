// bit_words_selftest — CPU check of host/compat/bit_words.h: Pack / Unpack of std::vector<bool> against the per-bit definition for
// every length 0 .. 300 (whole and partial 32- and 64-bit words), random contents, dirty padding bits in the container's last word.
// Prints "ok <cases> fast=<0|1>" and exits 0, or the first mismatch and 1.  Needs no device.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "bit_words.h"

int main() {
    std::mt19937 rng(2027);
    size_t cases = 0;
    for (size_t n_bits = 0; n_bits <= 300; ++n_bits) {
        for (int trial = 0; trial < 8; ++trial) {
            std::vector<bool> bits(n_bits);
            for (size_t b = 0; b < n_bits; ++b) {
                bits[b] = (rng() & 1u) != 0;
            }
            // leave garbage in the container's padding: grow with ones, then shrink (the storage keeps the bits)
            bits.resize(n_bits + 70, true);
            bits.resize(n_bits);
            const size_t n_words = (n_bits + 31) / 32 + 2;
            std::vector<uint32_t> words(n_words, 0xDEADBEEFu), expect(n_words, 0u);
            for (size_t b = 0; b < n_bits; ++b) {
                if (bits[b]) {
                    expect[b >> 5] |= 1u << (b & 31);
                }
            }
            feature_tracker::bit_words::Pack(bits, words.data(), n_words);
            if (words != expect) {
                printf("Pack mismatch at %zu bits\n", n_bits);
                return 1;
            }
            std::vector<bool> back(5, true);
            std::vector<uint32_t> dirty = expect;
            if (n_bits & 31) {
                dirty[(n_bits - 1) >> 5] |= ~((1u << (n_bits & 31)) - 1u);  // bits beyond n_bits in the source words must be ignored
            }
            feature_tracker::bit_words::Unpack(dirty.data(), n_bits, back);
            if (back != bits) {
                printf("Unpack mismatch at %zu bits\n", n_bits);
                return 1;
            }
            ++cases;
        }
    }
    printf("ok %zu fast=%d\n", cases, FTK_BIT_WORDS_FAST);
    return 0;
}
