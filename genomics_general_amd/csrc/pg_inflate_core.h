// DEFLATE (RFC 1951) decoder of ONE gzip member by ONE wavefront -- the core of k_inflate (pg_inflate.hip).
//
// `-g input.geno.gz` is the reference's normal input (popgenWindows.py:313, genomics.py:1917 gzip.open; the producer is
// `parseVCF.py ... | bgzip > out.geno.gz`, VCF_processing/README.md:33): bgzip writes BGZF, a gzip file made of independent members of
// at most 64 KiB of text each.  The members of a block of the input are copied to the device as they are and inflated there, a
// wavefront per member, straight into the text slot of the device tokenizer (pg_tokenize.hip) -- the text never exists on the host.
//
// How a wavefront decodes a serial bit stream:
//   * control flow is uniform: bit buffer, symbol, length, distance live in SGPRs (values come out of readlane / readfirstlane);
//   * the compressed bytes sit in two VGPRs (lane i holds dword i of the current / next 256-byte piece): a refill is a v_readlane,
//     a new piece one coalesced load, issued a whole piece ahead of its first use;
//   * Huffman codes are decoded canonically WITHOUT tables: lane L holds the left-aligned limit of code length L, the next 15
//     stream bits are bit-reversed (s_brev) and compared against all limits at once (v_cmp + ballot), the lowest set bit is the code
//     length, one LDS read of the symbol list sorted by (length, symbol) gives the symbol.  Building that for a dynamic block is a
//     histogram, a 15-step scan and a ranking by ballots -- no 2^k-entry tables to fill per block;
//   * length / distance bases and extra-bit counts are lane constants (readlane);
//   * a match is copied by the whole wavefront (64 bytes per load / store pair); a distance shorter than the wavefront becomes a
//     pattern fill.  The output window is the output itself (global memory): a wavefront's vector memory operations execute in
//     order, so a load sees the bytes an earlier store instruction of the same wavefront wrote.
//
// The same source is compiled twice: for the device (pg_inflate.hip), and -- with PG_INFLATE_EMULATE -- as a lockstep emulation of
// the 64 lanes on the host, which tests/inflate_emul.cpp holds against zlib in the CPU suite (the per-lane statements are written
// one memory operation per LANES block so that the emulation runs them in the order the hardware does).  The emulation is test
// infrastructure; no product path runs it.
#pragma once
#include <stdint.h>

#include "pg_inflate.h"

#ifndef PGI_LUT
#define PGI_LUT 1                 // a direct table for codes of up to eight bits beside the table-free canonical decode
#endif

#ifdef PG_INFLATE_EMULATE
#define PGI_DEV static inline
#define PL(type, name) type name[64]
#define PLREF(type, name) type *name
#define V(name) name[lane]
#define LANES for (int lane = 0; lane < 64; ++lane)
#define READLANE(name, l) (name[(l)])
#define WRITELANE(name, l, val) (name[(l)] = (val))
#define BALLOT(mask, expr)                                    \
    do {                                                      \
        mask = 0;                                             \
        for (int lane = 0; lane < 64; ++lane)                 \
            if (expr) mask |= 1ull << lane;                   \
    } while (0)
#define UNI(x) (x)
#define UNI64(x) (x)
#define PGI_IN_VGPR(x) (x)
#define PGI_SYNC
#define PGI_ATOMIC_INC(p) (++*(p))
#define PGI_LANE_PARAM
#define PGI_LANE_ARG
#define PGI_NOUNROLL
static inline uint32_t pgi_brev32(uint32_t x) {
    x = (x >> 16) | (x << 16);
    x = ((x & 0xFF00FF00u) >> 8) | ((x & 0x00FF00FFu) << 8);
    x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
    return x;
}
#define PGI_CTZ64(x) __builtin_ctzll(x)
#define PGI_POPC64(x) __builtin_popcountll(x)
#else
#define PGI_DEV __device__ __forceinline__ static
#define PL(type, name) type name
#define PLREF(type, name) type &name
#define V(name) name
#define LANES
#define READLANE(name, l) __builtin_amdgcn_readlane((int)(name), (int)(l))
// (a select, not a branch: a divergent branch around it would make every value the compiler sinks into its arms look divergent)
template <class T, class U>
__device__ __forceinline__ void pgi_setlane(T &name, int lane, int l, U val) {
    name = lane == l ? (T)val : name;
}
#define WRITELANE(name, l, val) pgi_setlane(name, lane, (int)(l), (val))
#define BALLOT(mask, expr) mask = __ballot(expr)
#define UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
#define UNI64(x) (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((x) >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x)))
// the bit buffer lives in VECTOR registers (every lane the same value): the scalar unit issues one instruction per SIMD every four cycles and is
// what bounds the kernel, the vector unit has room -- so the shifts and masks of the bit buffer run there, and only what steers the
// control flow (a code length, a symbol, extra bits) comes back through v_readfirstlane.  The empty asm makes the value "not known to be
// uniform" to the compiler.
__device__ __forceinline__ uint64_t pgi_in_vgpr(uint64_t x) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return ((uint64_t)hi << 32) | lo;
}
#define PGI_IN_VGPR(x) pgi_in_vgpr(x)
#define PGI_SYNC __syncthreads()
#define PGI_ATOMIC_INC(p) atomicAdd((p), 1u)
#define PGI_LANE_PARAM , const int lane
#define PGI_LANE_ARG , lane
#define PGI_NOUNROLL _Pragma("nounroll")
#define pgi_brev32(x) __builtin_bitreverse32(x)
#define PGI_CTZ64(x) __builtin_ctzll(x)
#define PGI_POPC64(x) __popcll(x)
#endif

// LDS of one wavefront
// Every per-lane statement below is written WITHOUT a branch on the lane (a lane that has nothing to store stores into a dump
// area behind the array, a lane that has nothing to load loads a valid address and drops the value): the compiler threads jumps
// through divergent branches, and every uniform value it then merges behind them (bit buffer, bit count, ...) turns "divergent",
// i.e. moves from the scalar unit into vector registers -- the decode loop then runs at a fraction of its speed.
struct PgiShared {
    uint32_t hist[64];            // [0] collects the lanes without a symbol
    uint16_t sorted_ll[288 + 64]; // literal / length symbols sorted by (code length, symbol) | dump
    uint16_t sorted_d[32 + 64];   // distance symbols (and, while a dynamic header is read, the 19 symbols of the code-length code) | dump
    uint8_t lens[320 + 64];       // code lengths: literal / length symbols, then the distance symbols | dump
#if PGI_LUT
    uint16_t lut_ll[256];         // codes of up to eight bits, by the next eight stream bits: symbol | length << 12 (0: a longer code)
    uint16_t lut_d[256];
#endif
    alignas(16) uint8_t ring[4096];   // the last PGI_RING bytes of the output
};

#if PGI_LUT
// The direct table of a code pgi_build has just described (lim, bas, sorted): entry i = what the canonical decode gives for the
// eight stream bits i when the code is at most eight bits long (those bits alone decide it then), else 0.
PGI_DEV void pgi_build_lut(PLREF(uint32_t, lim), PLREF(int32_t, bas), const uint16_t *sorted, int n_sorted, uint16_t *lut PGI_LANE_PARAM) {
    for (int g = 0; g < 256; g += 64) {
        PL(uint32_t, c15);
        PL(uint32_t, lsel);
        PL(int32_t, bsel);
        LANES {
            V(c15) = pgi_brev32((uint32_t)(g + lane)) >> 17;
            V(lsel) = 0u;
            V(bsel) = 0;
        }
        PGI_NOUNROLL
        for (int L = 1; L <= 8; ++L) {
            const uint32_t limL = (uint32_t)READLANE(lim, L);
            const int32_t basL = (int32_t)READLANE(bas, L);
            LANES {
                const bool hit = V(lsel) == 0u && V(c15) < limL;
                V(lsel) = hit ? (uint32_t)L : V(lsel);
                V(bsel) = hit ? basL : V(bsel);
            }
        }
        LANES {
            const uint32_t l = V(lsel) ? V(lsel) : 1u;
            const uint32_t ix = (uint32_t)((int32_t)(V(c15) >> (15u - l)) + V(bsel));
            const bool ok = V(lsel) != 0u && ix < (uint32_t)n_sorted;
            const uint32_t sym = sorted[ok ? ix : 0u];
            lut[g + lane] = (uint16_t)(ok ? (sym | (V(lsel) << 12)) : 0u);
        }
    }
    PGI_SYNC;
}
#endif

// The output window.  The text a member inflates to is written into a ring in LDS and leaves for global memory in pieces of 1 KiB
// (64 lanes x 16 bytes, aligned stores), so
//   * a match whose source lies in the ring -- nearly all of them: the line above, or the one above that -- is an LDS read and an LDS
//     write: no trip to the L2, and no wait for the acknowledgement of earlier stores (gfx9 counts loads and stores in ONE counter:
//     with byte stores per symbol every load of a match waited for every store before it -- 3700 cycles per symbol, measured);
//   * a match from further back reads global memory, which by then holds those bytes (everything older than the ring minus the
//     longest match has been flushed), and finds at most two flush stores in flight;
//   * the stores are few and wide (~70 per member instead of ~1500 byte-wide ones).
// The ring index of output byte p is (p + A) mod PGI_RING with A = the misalignment of the member's first byte in global memory, so
// that 16-byte aligned pieces of the destination are 16-byte aligned in the ring.  Lanes past the end of a match write into ring
// slots ahead of the output (no select): those are rewritten before they are read, and what they overwrite lies further back than
// any source the ring is asked for (PGI_NEAR).
#define PGI_RING 4096u
#define PGI_NEAR (PGI_RING - 320u)          // sources up to this distance are read from the ring
#define PGI_FLUSH_AT 2048u                  // bytes not yet in global memory that start a flush
struct alignas(16) PgiU4 {
    uint32_t x, y, z, w;
};

// The canonical code of `n` symbols with lengths lens[] (0 = unused): lane L of lim holds (first code of length L + their number),
// left-aligned to 15 bits; lane L of bas the index of the first symbol of length L in sorted[] minus its first code.
// kind 0: code-length code, 1: literal / length, 2: distance (what zlib's inflate_table accepts: an incomplete set only when it is
// a single code of one bit -- never for the code-length code --, no code at all: every decode then fails).
PGI_DEV int pgi_build(const uint8_t *lens, int n, uint16_t *sorted, uint32_t *hist, PLREF(uint32_t, lim), PLREF(int32_t, bas),
                      int dump, int kind PGI_LANE_PARAM) {
    LANES { hist[lane] = 0; }
    PGI_SYNC;
    for (int g = 0; g < n; g += 64) {
        LANES {
            const int s = g + lane;                      // (< 320 + 64: inside lens[] whatever n is)
            const int l = lens[s];
            PGI_ATOMIC_INC(&hist[s < n ? l : 0]);
        }
    }
    PGI_SYNC;
    PL(uint32_t, cnt);
    PL(uint32_t, nextpos);
    LANES {
        V(cnt) = lane < 16 ? hist[lane] : 0u;
        V(lim) = 0u;
        V(bas) = 0;
        V(nextpos) = 0u;
    }
    uint32_t code = 0, off = 0;
    int over = 0, maxlen = 0;
    PGI_NOUNROLL
    for (int L = 1; L <= 15; ++L) {
        const uint32_t c = (uint32_t)READLANE(cnt, L);
        const uint32_t first = code;
        code += c;
        if (code > (1u << L)) over = 1;
        if (c) maxlen = L;
        WRITELANE(lim, L, code << (15 - L));
        WRITELANE(bas, L, (int32_t)off - (int32_t)first);
        WRITELANE(nextpos, L, off);
        off += c;
        code <<= 1;
    }
    if (over) return PGI_ERR_CODE;
    if (code != (1u << 16) && (kind == 0 || maxlen > 1)) return PGI_ERR_CODE;          // incomplete
    for (int g = 0; g < n; g += 64) {
        PL(int, l);
        LANES {
            const int s = g + lane;
            const int x = (int)lens[s];
            V(l) = s < n ? x : 0;
        }
        uint64_t any;
        BALLOT(any, V(l) != 0);
        if (!any) continue;
        for (int L = 1; L <= maxlen; ++L) {
            uint64_t m;
            BALLOT(m, V(l) == L);
            if (!m) continue;
            const uint32_t p0 = (uint32_t)READLANE(nextpos, L);
            LANES {
                const uint32_t at = V(l) == L ? p0 + (uint32_t)PGI_POPC64(m & ((1ull << lane) - 1ull)) : (uint32_t)(dump + lane);
                sorted[at] = (uint16_t)(g + lane);
            }
            WRITELANE(nextpos, L, p0 + (uint32_t)PGI_POPC64(m));
        }
    }
    PGI_SYNC;
    return 0;
}

// the bit reader (see the header comment); `comp` as dwords, n_dw of them may be read.  A piece that reaches past them repeats the
// last dword (no select on the loaded value: the load must stay in flight until the piece is needed); those bits are never consumed by
// a valid stream, and a damaged one is stopped by the check against the member's last bit.
#define PGI_FILL1()                                                                  \
    do {                                                                             \
        const uint32_t w_ = (uint32_t)READLANE(cur, widx & 63u);                     \
        buf = PGI_IN_VGPR(buf | ((uint64_t)w_ << cnt));                              \
        cnt += 32;                                                                   \
        ++widx;                                                                      \
        if ((widx & 63u) == 0u) {                                                    \
            LANES { V(cur) = V(nxt); }                                               \
            LANES {                                                                  \
                const uint32_t a_ = widx + 64u + (uint32_t)lane;                     \
                V(nxt) = comp[a_ < n_dw ? a_ : n_dw - 1u];                           \
            }                                                                        \
        }                                                                            \
    } while (0)
#define PGI_NEED(n)                     \
    do {                                \
        while (cnt < (n)) PGI_FILL1();  \
    } while (0)
#define PGI_DROP(n)        \
    do {                   \
        buf >>= (n);       \
        cnt -= (int)(n);   \
    } while (0)
#define PGI_SEEK(byte_off)                                                  \
    do {                                                                    \
        const uint32_t d_ = (uint32_t)((byte_off) >> 2);                    \
        const uint32_t c0_ = d_ & ~63u;                                     \
        LANES {                                                             \
            const uint32_t a_ = c0_ + (uint32_t)lane;                       \
            V(cur) = comp[a_ < n_dw ? a_ : n_dw - 1u];                      \
        }                                                                   \
        LANES {                                                             \
            const uint32_t a_ = c0_ + 64u + (uint32_t)lane;                 \
            V(nxt) = comp[a_ < n_dw ? a_ : n_dw - 1u];                      \
        }                                                                   \
        widx = d_;                                                          \
        buf = PGI_IN_VGPR(0);                                               \
        cnt = 0;                                                            \
        const int drop_ = (int)((byte_off) & 3u) * 8;                       \
        if (drop_) {                                                        \
            PGI_FILL1();                                                    \
            PGI_DROP(drop_);                                                \
        }                                                                   \
    } while (0)
// the next symbol of the code (lim, bas, sorted): its length in L_, the symbol in sym_
#define PGI_DECODE(sym_, L_, lim, bas, sorted)                                                   \
    do {                                                                                         \
        const uint32_t c15_ = pgi_brev32((uint32_t)buf) >> 17;                                   \
        uint64_t mm_;                                                                            \
        BALLOT(mm_, c15_ < V(lim));                                                              \
        if (!mm_) return PGI_ERR_CODE;                                                           \
        L_ = (int)PGI_CTZ64(mm_);                                                                \
        sym_ = (int)UNI(sorted[(int)(c15_ >> (15 - L_)) + (int)READLANE(bas, L_)]);              \
        PGI_DROP(L_);                                                                            \
    } while (0)

// The same inside the symbol loop, where nothing branches out: a bit pattern that is no code of an (incomplete) code yields the
// symbol `invalid_` (which the loop's one accumulated check turns into PGI_ERR_CODE) and takes 15 bits.  Every early exit of the loop
// costs the compiler a copy of the loop's live registers on each path; with none the loop is a fifth shorter.
#define PGI_DECODE_NOEXIT(sym_, L_, lim, bas, sorted, n_sorted_, invalid_)                       \
    do {                                                                                         \
        const uint32_t c15_ = pgi_brev32((uint32_t)buf) >> 17;                                   \
        uint64_t mm_;                                                                            \
        BALLOT(mm_, c15_ < V(lim));                                                              \
        L_ = (int)PGI_CTZ64(mm_ | (1ull << 15));                                                 \
        const uint32_t ix_ = (uint32_t)((int)(c15_ >> (15 - L_)) + (int)READLANE(bas, L_));     \
        const int s_ = (int)UNI(sorted[ix_ < (uint32_t)(n_sorted_) ? ix_ : 0u]);                 \
        sym_ = mm_ ? s_ : (invalid_);                                                            \
        PGI_DROP(L_);                                                                            \
    } while (0)

// ---- CRC-32 of the member's text, taken where the text leaves for global memory (the flush holds it in registers) -----------------
// Lane i checksums bytes [16 i, 16 i + 16) of every aligned 1 KiB piece: four table steps for its sixteen bytes (multiplication by
// x^32, sliced by the four bytes of the register) and, in front of every piece but the first, one more for the 1008 bytes of the other
// lanes in between (multiplication by x^8064) -- CRCs are linear, so the lanes' registers can run apart and be XORed at the end, each
// first moved to the end of the aligned text (x^(128 m), m = the 16-byte chunks still to go: one 32-step product per lane and member).
// The bytes in front of the first aligned piece and behind the last whole chunk go through the byte table.  Tables (global memory,
// 9.5 KB: they stay in the L1): [0, 256) the byte table, [256, 1280) x^32 sliced, [1280, 2304) x^8064 sliced, [2304, 2368) x^(128 m).
#define PGI_CRC_POLY 0xEDB88320u
#define PGI_CRC_TAB 2368
static inline uint32_t pgi_crc_mul_host(uint32_t a, uint32_t b) {          // a * b mod P, reflected (zlib's multmodp)
    uint32_t p = 0;
    for (int k = 31; k >= 0; --k) {
        if ((a >> k) & 1u) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? PGI_CRC_POLY : 0u);
    }
    return p;
}
static inline void pgi_make_crc_tables(uint32_t *t) {
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t c = b;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? PGI_CRC_POLY : 0u);
        t[b] = c;
    }
    uint32_t x8 = 0x80000000u;                                              // x^0 ...
    for (int k = 0; k < 8; ++k) x8 = (x8 >> 1) ^ ((x8 & 1u) ? PGI_CRC_POLY : 0u);       // ... x^8
    auto power = [&](uint32_t bytes) {                                      // x^(8 * bytes)
        uint32_t r = 0x80000000u, sq = x8;
        for (uint32_t e = bytes; e; e >>= 1) {
            if (e & 1u) r = pgi_crc_mul_host(r, sq);
            sq = pgi_crc_mul_host(sq, sq);
        }
        return r;
    };
    const uint32_t x32 = power(4), x8064 = power(1008), x128 = power(16);
    for (int q = 0; q < 4; ++q)
        for (uint32_t b = 0; b < 256; ++b) {
            t[256 + 256 * q + b] = pgi_crc_mul_host(x32, b << (8 * q));
            t[1280 + 256 * q + b] = pgi_crc_mul_host(x8064, b << (8 * q));
        }
    t[2304] = 0x80000000u;
    for (int m = 1; m < 64; ++m) t[2304 + m] = pgi_crc_mul_host(t[2304 + m - 1], x128);
}

#if PGI_LUT
// ... with the direct table in front: one LDS read gives symbol and length of a code of up to eight bits
#define PGI_DECODE_LUT(sym_, L_, lut_, lim, bas, sorted, n_sorted_, invalid_)                     \
    do {                                                                                         \
        const uint32_t e_ = (uint32_t)UNI(lut_[(uint32_t)buf & 255u]);                           \
        if (e_) {                                                                                \
            sym_ = (int)(e_ & 0xFFFu);                                                           \
            L_ = (int)(e_ >> 12);                                                                \
            PGI_DROP(L_);                                                                        \
        } else {                                                                                 \
            PGI_DECODE_NOEXIT(sym_, L_, lim, bas, sorted, n_sorted_, invalid_);                  \
        }                                                                                        \
    } while (0)
#else
#define PGI_DECODE_LUT(sym_, L_, lut_, lim, bas, sorted, n_sorted_, invalid_) PGI_DECODE_NOEXIT(sym_, L_, lim, bas, sorted, n_sorted_, invalid_)
#endif

// One member: in_len bytes of deflate stream at byte in_off of comp -> out_len bytes at dst.  0, or PGI_ERR_* bits.
// sink: 128 bytes of the member's own where lanes without a byte store.  nl_list (may be null): the offsets, in the member's text,
// of its line feeds in front of offset nl_lim, in order -- found in the registers of the flush, so that the tokenizer needs no pass
// over the text for them (k_nl_count / k_nl_write); at most nl_cap are stored, *nl_n_out counts them all.
PGI_DEV int pgi_member(const uint32_t *__restrict__ comp, uint32_t n_dw, uint32_t in_off, uint32_t in_len, uint8_t *dst,
                       uint32_t out_len, uint8_t *sink, PgiShared *sh, uint16_t *nl_list, uint32_t nl_cap, uint32_t nl_lim,
                       uint32_t *nl_n_out, const uint32_t *__restrict__ crc_tab, uint32_t want_crc PGI_LANE_PARAM) {
    uint32_t nl_n = 0;
    // crc_tab (may be null: no check): pgi_make_crc_tables; want_crc: the CRC-32 of the member's trailer
    PL(uint32_t, cs);                                // the lanes' CRC registers
    LANES { V(cs) = lane == 0 ? 0xFFFFFFFFu : 0u; }
    uint32_t crc_pieces = 0;                         // aligned 1 KiB pieces checksummed so far
// v * x^k mod P by the sliced table at off_ (x^32: 256, x^8064: 1280)
#define PGI_CRC_SLICED(v, off_) (crc_tab[(off_) + ((v) & 255u)] ^ crc_tab[(off_) + 256u + (((v) >> 8) & 255u)] ^ \
                                 crc_tab[(off_) + 512u + (((v) >> 16) & 255u)] ^ crc_tab[(off_) + 768u + ((v) >> 24)])
// one byte into a register (the byte table)
#define PGI_CRC_BYTE(s_, b_) (crc_tab[((s_) ^ (uint32_t)(b_)) & 255u] ^ ((s_) >> 8))
    uint16_t *const nl_dump = reinterpret_cast<uint16_t *>(sink);
// a line feed at offset p_ of the member's text
#define PGI_NL_PUT(p)                                                                                          \
    do {                                                                                                       \
        const uint32_t p_ = (p);                                                                               \
        if (p_ < nl_lim) {                                                                                     \
            LANES { *((lane == 0 && nl_n < nl_cap) ? nl_list + nl_n : nl_dump + lane) = (uint16_t)p_; }        \
            ++nl_n;                                                                                            \
        }                                                                                                      \
    } while (0)
// 0x80 in every byte of x that is a line feed (exact: no borrow between the bytes)
#define PGI_NL_FLAGS(x) (~((((x) ^ 0x0A0A0A0Au) & 0x7F7F7F7Fu) + 0x7F7F7F7Fu | ((x) ^ 0x0A0A0A0Au) | 0x7F7F7F7Fu))
    // length / distance bases and extra bits (RFC 1951 3.2.5) as lane constants: base | extra << 16
    PL(uint32_t, lconst);
    PL(uint32_t, dconst);
    LANES {
        const uint32_t i = (uint32_t)lane;
        uint32_t lb, le, db, de;
        if (i < 8u) {
            lb = 3u + i;
            le = 0u;
        } else if (i < 28u) {
            le = (i >> 2) - 1u;
            lb = 3u + ((4u + (i & 3u)) << le);
        } else {
            lb = 258u;
            le = 0u;
        }
        if (i < 4u) {
            db = 1u + i;
            de = 0u;
        } else {
            de = ((i >> 1) - 1u) & 15u;
            db = 1u + ((2u + (i & 1u)) << de);
        }
        V(lconst) = lb | (le << 16);
        V(dconst) = (db & 0xFFFFu) | (de << 16);     // (bases up to 24577 fit 16 bits)
    }
    if (n_dw == 0u) return PGI_ERR_IN;
    PL(uint32_t, cur);
    PL(uint32_t, nxt);
    uint32_t widx;
    uint64_t buf;
    int cnt;
    PGI_SEEK(in_off);
    const uint64_t end_bits = ((uint64_t)in_off + in_len) * 8u;
    PL(uint32_t, lim_ll);
    PL(int32_t, bas_ll);
    PL(uint32_t, lim_d);
    PL(int32_t, bas_d);
    uint32_t pos = 0, fl = 0;                        // output bytes produced / of them in global memory already
    const uint32_t A = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
    uint8_t *const ring = sh->ring;
#define PGI_RIX(p) (((uint32_t)(p) + A) & (PGI_RING - 1u))
// n < 64 bytes from the ring to global memory, a byte per lane (the head in front of the first aligned piece, the last bytes
// behind the last whole 16-byte chunk); creg_: the CRC register these bytes go into (uniform)
#define PGI_FLUSH_BYTES(n, creg_)                                                 \
    do {                                                                          \
        PL(uint8_t, fb_);                                                         \
        LANES { V(fb_) = ring[PGI_RIX(fl + (uint32_t)lane)]; }                    \
        LANES { *((uint32_t)lane < (n) ? dst + fl + lane : sink + lane) = V(fb_); } \
        if (nl_list) {                                                            \
            uint64_t nm_;                                                         \
            BALLOT(nm_, V(fb_) == 10 && (uint32_t)lane < (n));                    \
            while (nm_) {                                                         \
                const uint32_t b_ = (uint32_t)PGI_CTZ64(nm_);                     \
                nm_ &= nm_ - 1ull;                                                \
                PGI_NL_PUT(fl + b_);                                              \
            }                                                                     \
        }                                                                         \
        if (crc_tab) {                                                            \
            for (uint32_t k_ = 0; k_ < (n); ++k_) {                               \
                const uint32_t by_ = (uint32_t)READLANE(fb_, k_) & 255u;          \
                creg_ = (uint32_t)UNI(PGI_CRC_BYTE(creg_, by_));                  \
            }                                                                     \
        }                                                                         \
        fl += (n);                                                                \
    } while (0)
// nl_ (1 .. 64) lanes x 16 bytes from the ring (16-byte aligned there and in global memory): stored, searched for line feeds, checksummed
#define PGI_PIECE(nl_)                                                                           \
    do {                                                                                         \
        PL(PgiU4, fq_);                                                                          \
        LANES { V(fq_) = *reinterpret_cast<const PgiU4 *>(ring + PGI_RIX(fl + 16u * (uint32_t)lane)); } \
        LANES { *reinterpret_cast<PgiU4 *>((uint32_t)lane < (nl_) ? dst + fl + 16u * (uint32_t)lane : sink) = V(fq_); } \
        if (nl_list) {                                                                           \
            PL(uint32_t, z0_);                                                                   \
            PL(uint32_t, z1_);                                                                   \
            PL(uint32_t, z2_);                                                                   \
            PL(uint32_t, z3_);                                                                   \
            LANES {                                                                              \
                V(z0_) = PGI_NL_FLAGS(V(fq_).x);                                                 \
                V(z1_) = PGI_NL_FLAGS(V(fq_).y);                                                 \
                V(z2_) = PGI_NL_FLAGS(V(fq_).z);                                                 \
                V(z3_) = PGI_NL_FLAGS(V(fq_).w);                                                 \
            }                                                                                    \
            uint64_t nm_;                                                                        \
            BALLOT(nm_, (V(z0_) | V(z1_) | V(z2_) | V(z3_)) != 0u && (uint32_t)lane < (nl_));    \
            while (nm_) {                                                                        \
                const uint32_t l_ = (uint32_t)PGI_CTZ64(nm_);                                    \
                nm_ &= nm_ - 1ull;                                                               \
                for (uint32_t d_ = 0; d_ < 4u; ++d_) {                                           \
                    uint32_t zz_ = (uint32_t)(d_ == 0u ? READLANE(z0_, l_) : d_ == 1u ? READLANE(z1_, l_) : d_ == 2u ? READLANE(z2_, l_) : READLANE(z3_, l_)); \
                    while (zz_) {                                                                \
                        const uint32_t b_ = (uint32_t)__builtin_ctz(zz_);                        \
                        zz_ &= zz_ - 1u;                                                         \
                        PGI_NL_PUT(fl + 16u * l_ + 4u * d_ + (b_ >> 3));                         \
                    }                                                                            \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        if (crc_tab) {                                                                           \
            LANES {                                                                              \
                uint32_t c_ = V(cs);                                                             \
                if (crc_pieces) c_ = PGI_CRC_SLICED(c_, 1280u);          /* (uniform) the 1008 bytes of the other lanes */ \
                c_ ^= V(fq_).x;                                                                  \
                c_ = PGI_CRC_SLICED(c_, 256u);                                                   \
                c_ ^= V(fq_).y;                                                                  \
                c_ = PGI_CRC_SLICED(c_, 256u);                                                   \
                c_ ^= V(fq_).z;                                                                  \
                c_ = PGI_CRC_SLICED(c_, 256u);                                                   \
                c_ ^= V(fq_).w;                                                                  \
                c_ = PGI_CRC_SLICED(c_, 256u);                                                   \
                V(cs) = (uint32_t)lane < (nl_) ? c_ : V(cs);                                     \
            }                                                                                    \
        }                                                                                        \
        fl += 16u * (nl_);                                                                       \
    } while (0)
#define PGI_FLUSH(final)                                                                         \
    do {                                                                                         \
        if (((fl + A) & 15u) != 0u) {                                                            \
            const uint32_t h_ = 16u - ((fl + A) & 15u), n_ = h_ < pos - fl ? h_ : pos - fl;      \
            uint32_t c0_ = (uint32_t)READLANE(cs, 0);                                            \
            PGI_FLUSH_BYTES(n_, c0_);                                                            \
            WRITELANE(cs, 0, c0_);                                                               \
        }                                                                                        \
        while (pos - fl >= 1024u) {                                                              \
            PGI_PIECE(64u);                                                                      \
            ++crc_pieces;                                                                        \
        }                                                                                        \
        if (final) {                                                                             \
            const uint32_t q_ = (pos - fl) >> 4;                 /* whole 16-byte chunks of the rest: lanes 0 .. q_ - 1 */ \
            if (q_) PGI_PIECE(q_);                                                               \
            uint32_t total_ = 0u;                                                                \
            if (crc_tab) {                                                                       \
                /* every lane's register to the end of the aligned text, then all of them XORed */ \
                PL(uint32_t, ca_);                                                               \
                PL(uint32_t, cb_);                                                               \
                PL(uint32_t, cp_);                                                               \
                LANES {                                                                          \
                    const uint32_t m_ = (uint32_t)lane < q_ ? q_ - (uint32_t)lane - 1u : (crc_pieces ? 63u + q_ - (uint32_t)lane : 0u); \
                    V(ca_) = crc_tab[2304u + (m_ & 63u)];                                        \
                    V(cb_) = V(cs);                                                              \
                    V(cp_) = 0u;                                                                 \
                }                                                                                \
                for (int k_ = 31; k_ >= 0; --k_) {                                               \
                    LANES {                                                                      \
                        V(cp_) ^= ((V(ca_) >> k_) & 1u) ? V(cb_) : 0u;                           \
                        V(cb_) = (V(cb_) >> 1) ^ ((V(cb_) & 1u) ? PGI_CRC_POLY : 0u);            \
                    }                                                                            \
                }                                                                                \
                for (int l_ = 0; l_ < 64; ++l_) total_ ^= (uint32_t)READLANE(cp_, l_);           \
            }                                                                                    \
            if (fl < pos) {                                                                      \
                const uint32_t r_ = pos - fl;                                                    \
                PGI_FLUSH_BYTES(r_, total_);                                                     \
            }                                                                                    \
            if (crc_tab && (total_ ^ 0xFFFFFFFFu) != want_crc) return PGI_ERR_CRC;               \
        }                                                                                        \
    } while (0)
    int fixed_built = 0;
    for (;;) {
        if ((uint64_t)widx * 32u - (uint64_t)cnt > end_bits) return PGI_ERR_IN;
        PGI_NEED(3);
        const uint32_t hdr = (uint32_t)UNI((uint32_t)buf & 7u);
        const int bfinal = (int)(hdr & 1u), btype = (int)(hdr >> 1);
        PGI_DROP(3);
        if (btype == 3) return PGI_ERR_BTYPE;
        if (btype == 0) {
            PGI_DROP(cnt & 7);
            PGI_NEED(32);
            const uint32_t lw = (uint32_t)UNI((uint32_t)buf);
            const uint32_t len = lw & 0xFFFFu, nlen = lw >> 16;
            PGI_DROP(32);
            if ((len ^ 0xFFFFu) != nlen) return PGI_ERR_STORED;
            const uint64_t bp = ((uint64_t)widx * 32u - (uint64_t)cnt) >> 3;
            if (bp + len > (uint64_t)in_off + in_len) return PGI_ERR_IN;
            if ((uint64_t)pos + len > out_len) return PGI_ERR_OUT;
            const uint8_t *src = reinterpret_cast<const uint8_t *>(comp) + bp;
            for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
                PL(uint8_t, v);
                LANES {
                    const uint32_t i = i0 + (uint32_t)lane;
                    V(v) = src[i < len ? i : 0u];
                }
                LANES { ring[PGI_RIX(pos + (uint32_t)lane)] = V(v); }
                pos += len - i0 < 64u ? len - i0 : 64u;
                if (pos - fl >= PGI_FLUSH_AT) PGI_FLUSH(0);
            }
            const uint64_t np = bp + len;
            PGI_SEEK(np);
            fixed_built = 0;            // (nothing lost, but keep the flag honest: the tables below are per block)
        } else {
            int hlit = 288, hdist = 32;                      // the fixed code (3.2.6)
            const int build = btype == 2 || !fixed_built;
            if (btype == 1) {
                if (!fixed_built) {
                    for (int g = 0; g < 320; g += 64) {
                        LANES {
                            const int s = g + lane;
                            sh->lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                        }
                    }
                    PGI_SYNC;
                }
            } else {
                PGI_NEED(14);
                const uint32_t hw = (uint32_t)UNI((uint32_t)buf & 0x3FFFu);
                hlit = (int)(hw & 31u) + 257;
                hdist = (int)((hw >> 5) & 31u) + 1;
                const int hclen = (int)((hw >> 10) & 15u) + 4;
                PGI_DROP(14);
                if (hlit > 286 || hdist > 30) return PGI_ERR_CODE;
                // the code-length code: 3 bits each, in the order of RFC 1951 3.2.7
                LANES { sh->lens[lane] = 0; }
                PGI_SYNC;
                for (int i = 0; i < hclen; ++i) {
                    PGI_NEED(3);
                    const int v = (int)UNI((uint32_t)buf & 7u);
                    PGI_DROP(3);
                    // order: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                    int sym;
                    if (i < 3) sym = 16 + i;
                    else if (i == 3) sym = 0;
                    else if ((i & 1) == 0) sym = 8 + ((i - 4) >> 1);        // i = 4, 6, 8 ... 18 -> 8, 9, 10 ... 15
                    else sym = 7 - ((i - 5) >> 1);                           // i = 5, 7, 9 ... 17 -> 7, 6, 5 ... 1
                    LANES { sh->lens[sym] = (uint8_t)v; }                  // (every lane the same byte)
                }
                PGI_SYNC;
                PL(uint32_t, lim_p);
                PL(int32_t, bas_p);
                int rc = pgi_build(sh->lens, 19, sh->sorted_d, sh->hist, lim_p, bas_p, 32, 0 PGI_LANE_ARG);
                if (rc) return rc;
                // the code lengths of the literal / length and distance codes, run-length coded (3.2.7)
                const int total = hlit + hdist;
                int i = 0, prev = 0;
                while (i < total) {
                    PGI_NEED(22);
                    int sym, L;
                    PGI_DECODE(sym, L, lim_p, bas_p, sh->sorted_d);
                    if (sym < 16) {
                        LANES { sh->lens[i] = (uint8_t)sym; }
                        prev = sym;
                        ++i;
                        continue;
                    }
                    int rep, val;
                    if (sym == 16) {
                        if (i == 0) return PGI_ERR_CODE;
                        val = prev;
                        rep = 3 + (int)UNI((uint32_t)buf & 3u);
                        PGI_DROP(2);
                    } else if (sym == 17) {
                        val = 0;
                        rep = 3 + (int)UNI((uint32_t)buf & 7u);
                        PGI_DROP(3);
                    } else {
                        val = 0;
                        rep = 11 + (int)UNI((uint32_t)buf & 127u);
                        PGI_DROP(7);
                    }
                    if (i + rep > total) return PGI_ERR_CODE;
                    for (int r0 = 0; r0 < rep; r0 += 64) {
                        LANES { sh->lens[r0 + lane < rep ? i + r0 + lane : 320 + lane] = (uint8_t)val; }
                    }
                    prev = val;
                    i += rep;
                }
                PGI_SYNC;
                if ((int)UNI(sh->lens[256]) == 0) return PGI_ERR_CODE;               // no end-of-block code
                // the distance lengths follow the hlit literal / length lengths: move them where pgi_build of the fixed code has them
                PL(uint8_t, dl);
                LANES {
                    const uint8_t x = sh->lens[hlit + lane];               // (hlit + 63 < 320 + 64)
                    V(dl) = lane < hdist ? x : (uint8_t)0;
                }
                PGI_SYNC;
                LANES { sh->lens[lane < 32 ? 288 + lane : 320 + lane] = V(dl); }
                PGI_SYNC;
            }
            if (build) {
                // one copy of the table builder in the code for both codes of a block, fixed or dynamic: first the distance code, then
                // the literal / length code
                for (int t = 2; t >= 1; --t) {
                    PL(uint32_t, lim_t);
                    PL(int32_t, bas_t);
                    const int rc = pgi_build(t == 2 ? sh->lens + 288 : sh->lens, t == 2 ? hdist : hlit, t == 2 ? sh->sorted_d : sh->sorted_ll,
                                             sh->hist, lim_t, bas_t, t == 2 ? 32 : 288, t PGI_LANE_ARG);
                    if (rc) return rc;
#if PGI_LUT
                    pgi_build_lut(lim_t, bas_t, t == 2 ? sh->sorted_d : sh->sorted_ll, t == 2 ? 32 : 288, t == 2 ? sh->lut_d : sh->lut_ll PGI_LANE_ARG);
#endif
                    if (t == 2) {
                        LANES { V(lim_d) = V(lim_t); }
                        LANES { V(bas_d) = V(bas_t); }
                    } else {
                        LANES { V(lim_ll) = V(lim_t); }
                        LANES { V(bas_ll) = V(bas_t); }
                    }
                }
                fixed_built = btype == 1;
            }
            // ---- the symbols of the block ----
            // No check of a symbol leaves the loop: `acc` collects, in its sign bit, a length symbol beyond 28 and a distance symbol
            // beyond 29 (both also what an invalid bit pattern decodes to), `accd` a distance that reaches in front of the member; they
            // are looked at where the loop is left anyway -- at the end of the block, and in the flush branch every 2 KiB of output,
            // together with the output running past its recorded size.  Until then a damaged stream decodes garbage into the LDS
            // ring, which is harmless: the global memory is only touched by the flush and by matches from beyond the ring, and
            // those are taken only when their source lies inside the member.
            int32_t acc = 0, accd = 0;
            for (;;) {
                if (pos - fl >= PGI_FLUSH_AT) {
                    if ((acc | accd) < 0 || pos > out_len) break;
                    PGI_FLUSH(0);
                }
                PGI_NEED(32);
                int sym, L;
                PGI_DECODE_LUT(sym, L, sh->lut_ll, lim_ll, bas_ll, sh->sorted_ll, 288, 257 + 63);
                if (sym < 256) {
                    LANES { ring[PGI_RIX(pos)] = (uint8_t)sym; }           // (every lane the same byte)
                    ++pos;
                    continue;
                }
                if (sym == 256) break;
                sym -= 257;
                acc |= 28 - sym;
                const uint32_t lc = (uint32_t)READLANE(lconst, sym);               // (sym <= 63 whatever was decoded)
                const uint32_t le = lc >> 16;
                const uint32_t len = (lc & 0xFFFFu) + (uint32_t)UNI((uint32_t)buf & ((1u << le) - 1u));
                PGI_DROP(le);
                PGI_NEED(32);
                int dsym;
                PGI_DECODE_LUT(dsym, L, sh->lut_d, lim_d, bas_d, sh->sorted_d, 32, 63);
                acc |= 29 - dsym;
                const uint32_t dc = (uint32_t)READLANE(dconst, dsym);
                const uint32_t de = dc >> 16;
                const uint32_t dist = (dc & 0xFFFFu) + (uint32_t)UNI((uint32_t)buf & ((1u << de) - 1u));
                PGI_DROP(de);
                accd |= (int32_t)(pos - dist);                           // negative: the distance reaches in front of the member
                PL(uint8_t, v0);
                PL(uint8_t, v1);
                PL(uint8_t, v2);
                PL(uint8_t, v3);
                PL(uint8_t, v4);
                if (dist >= len) {
                    // source and destination do not overlap (the usual case: the match is in a line further up): all its bytes are read
                    // -- up to five reads of 64 bytes in flight at once -- before any is written
                    if (dist <= PGI_NEAR || dist > pos) {
                        const uint32_t sp = pos - dist + (uint32_t)0;
                        LANES { V(v0) = ring[PGI_RIX(sp + (uint32_t)lane)]; }
                        if (len > 64u) LANES { V(v1) = ring[PGI_RIX(sp + 64u + (uint32_t)lane)]; }
                        if (len > 128u) LANES { V(v2) = ring[PGI_RIX(sp + 128u + (uint32_t)lane)]; }
                        if (len > 192u) LANES { V(v3) = ring[PGI_RIX(sp + 192u + (uint32_t)lane)]; }
                        if (len > 256u) LANES { V(v4) = ring[PGI_RIX(sp + 256u + (uint32_t)lane)]; }
                    } else {
                        // further back than the ring reaches: those bytes are in global memory (flushed: see the note at the ring)
                        const uint8_t *sp = dst + (pos - dist);
                        LANES { V(v0) = sp[(uint32_t)lane < len ? (uint32_t)lane : 0u]; }
                        if (len > 64u) LANES { V(v1) = sp[64u + (uint32_t)lane < len ? 64u + (uint32_t)lane : 0u]; }
                        if (len > 128u) LANES { V(v2) = sp[128u + (uint32_t)lane < len ? 128u + (uint32_t)lane : 0u]; }
                        if (len > 192u) LANES { V(v3) = sp[192u + (uint32_t)lane < len ? 192u + (uint32_t)lane : 0u]; }
                        if (len > 256u) LANES { V(v4) = sp[256u + (uint32_t)lane < len ? 256u + (uint32_t)lane : 0u]; }
                    }
                    LANES { ring[PGI_RIX(pos + (uint32_t)lane)] = V(v0); }
                    if (len > 64u) LANES { ring[PGI_RIX(pos + 64u + (uint32_t)lane)] = V(v1); }
                    if (len > 128u) LANES { ring[PGI_RIX(pos + 128u + (uint32_t)lane)] = V(v2); }
                    if (len > 192u) LANES { ring[PGI_RIX(pos + 192u + (uint32_t)lane)] = V(v3); }
                    if (len > 256u) LANES { ring[PGI_RIX(pos + 256u + (uint32_t)lane)] = V(v4); }
                } else if (dist >= 64u) {
                    // the match overlaps itself with a period of 64 bytes or more: 64 bytes per step, a later step reads what an
                    // earlier one wrote (LDS operations of a wavefront execute in order)
                    for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
                        LANES { V(v0) = ring[PGI_RIX(pos - dist + i0 + (uint32_t)lane)]; }
                        LANES { ring[PGI_RIX(pos + i0 + (uint32_t)lane)] = V(v0); }
                    }
                } else {
                    // ... with a short period: a pattern of `dist` bytes, repeated.  Every lane holds the byte of its place in the
                    // pattern; a step writes as many whole periods as fit into 64 lanes (and the start of one more: the same bytes
                    // the next step writes there)
                    const uint32_t step = (64u / dist) * dist;
                    LANES { V(v0) = ring[PGI_RIX(pos - dist + (uint32_t)lane % dist)]; }
                    for (uint32_t i0 = 0; i0 < len; i0 += step) {
                        LANES { ring[PGI_RIX(pos + i0 + (uint32_t)lane)] = V(v0); }
                    }
                }
                pos += len;
            }
            if (acc < 0) return PGI_ERR_CODE;
            if (accd < 0) return PGI_ERR_DIST;
            if (pos > out_len) return PGI_ERR_OUT;
        }
        if (bfinal) break;
    }
    if ((uint64_t)widx * 32u - (uint64_t)cnt > end_bits) return PGI_ERR_IN;
    if (pos != out_len) return PGI_ERR_OUT;
    PGI_FLUSH(1);
    if (nl_n_out) *nl_n_out = nl_n;
    return 0;
}
