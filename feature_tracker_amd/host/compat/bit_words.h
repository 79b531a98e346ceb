// bit_words.h — std::vector<bool> <-> packed 32-bit words (bit b of a descriptor = bit b & 31 of word b >> 5), the layout of the
// device's Hamming matcher and BRIEF kernel.  The reference's API hands descriptors around as per-bit containers
// (feature_detector::BriefType); converting them bit by bit cost more host time than the device calls they frame (0.6 of the
// 0.76 ms test_descriptor_matcher_brief.cpp:69-88 spans).  libstdc++ keeps the bits of a vector<bool> packed LSB-first in
// unsigned long words behind _Bit_iterator::_M_p, which on a little-endian machine IS the layout above: whole descriptors move
// with memcpy.  Any other standard library takes the portable per-bit loops.
#ifndef _FTK_HOST_BIT_WORDS_H_
#define _FTK_HOST_BIT_WORDS_H_

#include <cstdint>
#include <cstring>
#include <vector>

namespace feature_tracker {
namespace bit_words {

#if defined(__GLIBCXX__) && defined(__BYTE_ORDER__) && (__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__)
#define FTK_BIT_WORDS_FAST 1
#else
#define FTK_BIT_WORDS_FAST 0
#endif

// words[0 .. n_words) <- bits (bits beyond bits.size() in the last word: 0).  n_words >= ceil(bits.size() / 32).
inline void Pack(const std::vector<bool> &bits, uint32_t *words, size_t n_words) {
    const size_t n_bits = bits.size();
    const size_t used = (n_bits + 31) / 32;
    for (size_t w = used; w < n_words; ++w) {
        words[w] = 0u;
    }
    if (n_bits == 0) {
        return;
    }
#if FTK_BIT_WORDS_FAST
    std::memcpy(words, bits.begin()._M_p, used * sizeof(uint32_t));  // the allocation is whole unsigned longs: never read past it
    if (n_bits & 31) {
        words[used - 1] &= (1u << (n_bits & 31)) - 1u;  // padding bits of the container are unspecified
    }
#else
    for (size_t w = 0; w < used; ++w) {
        words[w] = 0u;
    }
    for (size_t b = 0; b < n_bits; ++b) {
        if (bits[b]) {
            words[b >> 5] |= 1u << (b & 31);
        }
    }
#endif
}

// bits <- the first n_bits of words
inline void Unpack(const uint32_t *words, size_t n_bits, std::vector<bool> &bits) {
    bits.assign(n_bits, false);
    if (n_bits == 0) {
        return;
    }
#if FTK_BIT_WORDS_FAST
    const size_t used = (n_bits + 31) / 32;
    uint32_t last = words[used - 1];
    if (n_bits & 31) {
        last &= (1u << (n_bits & 31)) - 1u;
    }
    char *dst = reinterpret_cast<char *>(bits.begin()._M_p);
    std::memcpy(dst, words, (used - 1) * sizeof(uint32_t));
    std::memcpy(dst + (used - 1) * sizeof(uint32_t), &last, sizeof(uint32_t));
#else
    for (size_t b = 0; b < n_bits; ++b) {
        bits[b] = ((words[b >> 5] >> (b & 31)) & 1u) != 0;
    }
#endif
}

}  // namespace bit_words
}  // namespace feature_tracker

#endif  // _FTK_HOST_BIT_WORDS_H_
