// CRC-32 (gzip, reflected 0xEDB88320) by carry-less multiplication: sixty-four bytes are folded per step with PCLMULQDQ, the way
// Intel's "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ" describes it, then reduced to 32 bits (Barrett).  zlib
// 1.2.11's table-driven crc32 does 1 GB/s a thread on the hosts measured -- half the rate of this library's deflate decoder, and the
// bound of the chunk-parallel decoder's conversion pass (profiles/r06/gzip_stream_reader.txt); this one runs at memory speed.  Chosen
// at run time (__builtin_cpu_supports); every other host keeps zlib's.  Checked against zlib on every length and alignment
// (tests/test_inflate.py).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>

#if defined(__x86_64__)
#include <immintrin.h>

namespace pgcrc {

// crc: the register's value (zlib's crc xor 0xFFFFFFFF); len >= 64 and a multiple of 16
__attribute__((target("pclmul,sse4.1"))) static uint32_t fold(const uint8_t *buf, size_t len, uint32_t crc) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x00));
    x2 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x10));
    x3 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x20));
    x4 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(k1k2));
    buf += 64;
    len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x00));
        y6 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x10));
        y7 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x20));
        y8 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    // four registers into one
    x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(k3k4));
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    // sixteen bytes at a time
    while (len >= 16) {
        x2 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf));
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(k5k0));
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // 64 -> 32 bits (Barrett)
    x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(poly));
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

static inline bool have_clmul() {
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return ok;
}

}  // namespace pgcrc
#endif

// zlib's crc32_z(crc, buf, len), faster where the CPU can
static inline uint32_t pg_crc32(uint32_t crc, const uint8_t *buf, size_t len) {
#if defined(__x86_64__)
    if (len >= 64 && pgcrc::have_clmul()) {
        const size_t body = len & ~(size_t)15;
        crc = ~pgcrc::fold(buf, body, ~crc);
        buf += body;
        len -= body;
    }
#endif
    return len ? (uint32_t)crc32_z(crc, buf, len) : crc;
}
