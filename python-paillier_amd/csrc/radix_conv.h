// radix_conv.h — ciphertext limbs <-> decimal text, one number per thread.
//
// The reference's wire formats carry ciphertexts as DECIMAL STRINGS: docs/serialisation.rst:24-43
// ({"public_key": {"n": ...}, "values": [[str(ciphertext), exponent], ...]}) and the CLI's {"v": str(ciphertext), "e": ...}
// (phe/command_line.py:120-131, :267-276).  str(int) / int(str) on a 4096-bit number is quadratic work on one core
// (~21 us / ~9 us in CPython) — for a vector that is more time than encrypting it on the GPU — so the batch form of the
// wire format converts on the device: little-endian 32-bit words <-> fixed-width ASCII digits, left-padded with '0'
// (the host strips / adds the padding; the digits themselves are exactly str(int)'s).
//
// Both directions are schoolbook base conversions in base 10^9: to_decimal divides the number by 10^9 once per nine
// digits (the quotient shrinks by one word every ~1.07 rounds), from_decimal is Horner's rule x = x * 10^9 + chunk.
// The working number lives in memory the caller provides through an accessor — on the GPU a padded LDS tile indexed
// [word][thread] (conflict-free: the threads of a wave touch consecutive banks), in the CPU test build a plain array.
// Plain C++ without wave primitives: tests/emu/emu_driver.cpp compiles these very functions for the host.
#pragma once
#include <stdint.h>

#ifndef PHE_DEV
#define PHE_DEV inline
#endif

namespace phe {

constexpr uint32_t kDecChunk = 1000000000u;  // 10^9 < 2^32
constexpr int kDecChunkDigits = 9;

// digits needed for any number below 2^(32*words): floor(32*words*log10(2)) + 1, with log10(2) < 30103/100000
constexpr int decimal_width(int words) { return (int)(((int64_t)32 * words * 30103) / 100000) + 1; }

// x: `words` little-endian 32-bit words (destroyed); out[0..width): ASCII digits, most significant first, '0'-padded.
// Returns false if the number needs more than `width` digits.
template <class Words>
PHE_DEV bool limbs_to_decimal(Words x, int words, char* out, int width) {
    int top = words;
    while (top > 0 && x(top - 1) == 0u) --top;
    int pos = width;
    bool ok = true;
    while (top > 0) {
        uint32_t rem = 0;
        for (int j = top - 1; j >= 0; --j) {
            const uint64_t v = ((uint64_t)rem << 32) | x(j);
            const uint32_t q = (uint32_t)(v / kDecChunk);  // < 2^32 because rem < 10^9
            rem = (uint32_t)(v - (uint64_t)q * kDecChunk);
            x(j) = q;
        }
        if (x(top - 1) == 0u) --top;
        for (int d = 0; d < kDecChunkDigits; ++d) {
            const uint32_t t = rem / 10u;
            const char ch = (char)('0' + (rem - t * 10u));
            rem = t;
            if (pos > 0) out[--pos] = ch;
            else if (ch != '0') ok = false;
        }
    }
    while (pos > 0) out[--pos] = '0';
    return ok;
}

// in[0..width): ASCII digits, most significant first (leading '0's allowed); x: `words` words, written completely.
// Returns 0 on success, 1 for a character that is not a digit, 2 if the value does not fit `words` words.
template <class Words>
PHE_DEV int decimal_to_limbs(const char* in, int width, Words x, int words) {
    for (int j = 0; j < words; ++j) x(j) = 0u;
    int top = 0, status = 0;
    int first = width % kDecChunkDigits;  // the leading chunk takes the odd digits
    if (first == 0) first = kDecChunkDigits;
    for (int pos = 0; pos < width;) {
        const int len = (pos == 0) ? (first < width ? first : width) : kDecChunkDigits;
        uint32_t chunk = 0, mult = 1;
        for (int d = 0; d < len; ++d) {
            const uint32_t c = (uint32_t)(unsigned char)in[pos + d] - (uint32_t)'0';
            if (c > 9u) status = 1;
            chunk = chunk * 10u + (c > 9u ? 0u : c);
            mult *= 10u;
        }
        pos += len;
        uint32_t carry = chunk;  // x = x * 10^len + chunk
        for (int j = 0; j < top; ++j) {
            const uint64_t v = (uint64_t)x(j) * mult + carry;
            x(j) = (uint32_t)v;
            carry = (uint32_t)(v >> 32);
        }
        if (carry) {
            if (top < words) x(top++) = carry;
            else if (status == 0) status = 2;
        }
    }
    return status;
}

}  // namespace phe
