// A DEFLATE (RFC 1951) decoder for ONE long stream on ONE host thread -- the single-member `.geno.gz` that `gzip` writes and the
// reference reads with gzip.open (genomics.py:1917, README.md:106).  Such a file has no independent pieces, so whatever inflates it
// runs serially; zlib's inflate() does 0.35 - 0.45 GB/s of `.geno` text on the hosts measured (profiles/r06/gzip_stream_reader.txt),
// which made a plain-gzip north star a matter of minutes.  This decoder is written for that one job:
//   * the whole compressed file is mapped, so the bit reader refills with one unaligned 8-byte load and no end-of-buffer checks in the
//     inner loop (the last bytes of a member go through a zero-padded copy);
//   * literal / length codes are looked up 11 bits at a time (one table access for every code of `.geno` text), distances 8 bits at a
//     time, longer codes through second-level tables; an entry carries the base value and the number of extra bits, so a length or
//     distance costs one look-up, one mask and one add;
//   * a match is copied eight bytes at a time (`.geno` matches are a line long), a short-period overlap by a widened pattern;
//   * the output goes straight into the caller's block buffer; the decoder stops at any byte and resumes (state: bit buffer, block,
//     tables, the rest of a match), keeping the last 32 KiB of what it wrote for matches that reach behind the new buffer.
// Everything is bounds-checked against damaged input: a bad code, a distance beyond the output, a stream that runs out of bytes
// return an error.  Test: tests/test_inflate.py (every level, strategy and block type against zlib; damaged streams; AddressSanitizer
// through `make asan-test`).
#pragma once
#include <stdint.h>
#include <string.h>

namespace pgfi {

enum { OK = 0, NEED_OUTPUT = 1, STREAM_END = 2, ERR_DATA = -1, ERR_INPUT = -2 };

constexpr int LL_BITS = 11, D_BITS = 8;
constexpr int LL_SIZE = (1 << LL_BITS) + 4800, D_SIZE = (1 << D_BITS) + 4000;
// entry: bits 0-7 code length (bits to drop), 8-11 extra bits, 12-15 kind, 16-31 value (literal, length base, distance base, subtable offset)
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_DIST = 5 };

static inline uint32_t entry(uint32_t value, int kind, int extra, int len) { return (value << 16) | ((uint32_t)kind << 12) | ((uint32_t)extra << 8) | (uint32_t)len; }

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t rev_bits(uint32_t c, int n) {
    uint32_t r = 0;
    for (int k = 0; k < n; ++k) r |= ((c >> k) & 1u) << (n - 1 - k);
    return r;
}

// canonical code of n symbols with lengths lens[] -> look-up table (root_bits at a time, second-level tables behind the root).
// kind: 0 the code-length code (values = symbols, K_LIT), 1 literal / length, 2 distance.  false: over-subscribed, or incomplete
// in a way zlib refuses (only a distance code of a single one-bit code may be incomplete).
static bool build(const uint8_t *lens, int n, int kind, int root_bits, uint32_t *table, int table_size) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    count[0] = 0;
    int maxlen = 0;
    uint32_t code = 0, next[16];
    for (int L = 1; L <= 15; ++L) {
        code = (code + (uint32_t)count[L - 1]) << 1;
        next[L] = code;
        if (count[L]) maxlen = L;
        if (code + (uint32_t)count[L] > (1u << L)) return false;                       // over-subscribed
    }
    if (maxlen == 0) {                                                                 // no code at all: every look-up fails
        if (kind == 0) return false;
        for (int k = 0; k < (1 << root_bits); ++k) table[k] = entry(0, K_BAD, 0, 1);
        return true;
    }
    {
        uint32_t left = 1;                                                             // Kraft: complete?
        for (int L = 1; L <= 15; ++L) left = (left << 1) - (uint32_t)count[L];
        if (left != 0 && (kind == 0 || maxlen != 1)) return false;                        // (zlib: an incomplete set only as ONE one-bit code)
    }
    const int root = 1 << root_bits;
    for (int k = 0; k < root; ++k) table[k] = entry(0, K_BAD, 0, 1);
    // second-level tables: the longest code behind every root prefix
    uint8_t sub_bits[1 << 11];
    memset(sub_bits, 0, (size_t)root);
    if (maxlen > root_bits) {
        uint32_t nx[16];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < n; ++s) {
            const int L = lens[s];
            if (!L) continue;
            const uint32_t c = nx[L]++;
            if (L > root_bits) {
                const uint32_t r = rev_bits(c, L) & (uint32_t)(root - 1);
                if (L - root_bits > sub_bits[r]) sub_bits[r] = (uint8_t)(L - root_bits);
            }
        }
    }
    int used = root;
    for (int r = 0; r < root; ++r) {
        if (!sub_bits[r]) continue;
        const int size = 1 << sub_bits[r];
        if (used + size > table_size) return false;
        table[r] = entry((uint32_t)used, K_SUB, sub_bits[r], root_bits);
        for (int k = 0; k < size; ++k) table[used + k] = entry(0, K_BAD, 0, 1);
        used += size;
    }
    for (int s = 0; s < n; ++s) {
        const int L = lens[s];
        if (!L) continue;
        const uint32_t c = next[L]++;
        const uint32_t r = rev_bits(c, L);
        uint32_t e;
        if (kind == 0) e = entry((uint32_t)s, K_LIT, 0, L);
        else if (kind == 1) {
            if (s < 256) e = entry((uint32_t)s, K_LIT, 0, L);
            else if (s == 256) e = entry(0, K_EOB, 0, L);
            else if (s <= 285) e = entry(LEN_BASE[s - 257], K_LEN, LEN_EXTRA[s - 257], L);
            else e = entry(0, K_BAD, 0, L);
        } else {
            e = s < 30 ? entry(DIST_BASE[s], K_DIST, DIST_EXTRA[s], L) : entry(0, K_BAD, 0, L);
        }
        if (L <= root_bits) {
            for (uint32_t k = r; k < (uint32_t)root; k += 1u << L) table[k] = e;
        } else {
            const uint32_t pre = r & (uint32_t)(root - 1);
            const uint32_t base = table[pre] >> 16;
            const int sb = sub_bits[pre];
            e = (e & ~0xFFu) | (uint32_t)(L - root_bits);                             // bits to drop behind the root's
            for (uint32_t k = r >> root_bits; k < (1u << sb); k += 1u << (L - root_bits)) table[base + k] = e;
        }
    }
    return true;
}

struct State {
    // input: the mapped file
    const uint8_t *in = nullptr, *in_end = nullptr;
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    // block
    int phase = 0;              // 0: expect a block header, 1: inside a stored block, 2: inside a compressed block, 3: the stream has ended
    bool last = false;
    uint32_t stored_left = 0;
    uint32_t pend_len = 0, pend_dist = 0;       // the rest of a match that did not fit the previous output buffer
    bool fixed_built = false, tables_fixed = false;
    uint32_t ll[LL_SIZE], d[D_SIZE];
    // the last 32 KiB written to earlier output buffers (matches that reach behind the current one)
    uint8_t window[32768];
    uint32_t win_len = 0;       // valid bytes (the newest is window[win_len - 1] once full: kept linear, newest at the end)
    uint64_t total_out = 0;
    int over = 0;               // zero bytes fed to the bit buffer behind the end of the input (a valid stream never consumes them)
    // marker mode (pg_par_gunzip.h: a chunk decoded before the 32 KiB in front of it are known): bit positions, relative to `base`, of
    // block starts at which the decoder is to stop (ascending); stopped_at = the one it stopped at
    const uint8_t *base = nullptr;
    const uint64_t *stops = nullptr;
    int n_stops = 0;
    uint64_t stopped_at = 0;
};
enum { STOPPED = 3 };

static inline uint64_t load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;                   // (little-endian hosts: x86-64, aarch64 as configured here)
}

// refill from the mapped input; near its end from a zero-padded copy, so that eight bytes can always be loaded
struct Reader {
    State &s;
    explicit Reader(State &st) : s(st) {}
    inline void refill() {
        if (s.in_end - s.in >= 8) {
            s.bitbuf |= load64(s.in) << s.bitcnt;
            s.in += (63 - s.bitcnt) >> 3;
            s.bitcnt |= 56;
        } else {
            slow_refill();
        }
    }
    void slow_refill() {
        while (s.bitcnt <= 56) {
            uint64_t b = 0;                                                            // zeros behind the end: a valid stream never needs them
            if (s.in < s.in_end) b = *s.in++;
            else ++s.over;
            s.bitbuf |= b << s.bitcnt;
            s.bitcnt += 8;
        }
    }
};

template <typename T>
static inline void copy_match(T *out, uint32_t dist, uint32_t len) {
    // out[k] = out[k - dist]; may write up to 7 bytes past len (the caller leaves room)
    const T *src = out - dist;
    constexpr uint32_t PER = 8 / sizeof(T);                                            // elements per eight-byte copy
    if (dist >= PER) {
        T *end = out + len;
        do {
            memcpy(out, src, 8);
            out += PER;
            src += PER;
        } while (out < end);
    } else if (dist == 1 && sizeof(T) == 1) {
        memset(out, (int)*src, len);
    } else {
        for (uint32_t k = 0; k < len; ++k) out[k] = src[k];
    }
}

// byte k before the start of the current output buffer (k = 1: the newest), from the saved window
static inline uint8_t window_byte(const State &s, uint32_t k) { return s.window[s.win_len - k]; }

static void save_window(State &s, const uint8_t *out_begin, const uint8_t *out) {
    const uint64_t n = (uint64_t)(out - out_begin);
    if (n >= 32768) {
        memcpy(s.window, out - 32768, 32768);
        s.win_len = 32768;
    } else if (n > 0) {
        const uint32_t keep = (uint32_t)((uint64_t)s.win_len + n > 32768 ? 32768 - n : s.win_len);
        memmove(s.window, s.window + (s.win_len - keep), keep);
        memcpy(s.window + keep, out_begin, (size_t)n);
        s.win_len = keep + (uint32_t)n;
    }
}

// Inflate into out_begin[0 .. cap): returns NEED_OUTPUT (cap bytes written, more to come), STREAM_END (*n_out bytes written, the
// deflate stream has ended; s.in / s.bitcnt tell where), or an error.
// MARK (T = uint16_t): the 32 KiB in front of out_begin are not known yet -- a byte copied from there becomes the marker
// 0x8000 | its index in that window (0 = the oldest byte, 32767 = the byte right in front of out_begin), to be replaced later;
// out_begin is then the chunk's first element in every call, `at` the number of elements already written, and the decoder also
// stops (STOPPED) in front of a block that starts at one of s.stops.
template <typename T, bool MARK>
static int inflate_t(State &s, T *out_begin, uint64_t cap, uint64_t *n_out, uint64_t at = 0) {
    Reader rd(s);
    T *out = out_begin + at, *const out_end = out_begin + cap;
    T *const first = out;
    *n_out = 0;
    auto done = [&](int rc) {
        *n_out = (uint64_t)(out - first);
        s.total_out += *n_out;
        if (!MARK) save_window(s, reinterpret_cast<const uint8_t *>(out_begin), reinterpret_cast<const uint8_t *>(out));
        return rc;
    };
    auto window_at = [&](uint32_t k) -> T {                                             // element k in front of out_begin (k = 1: the newest)
        if (MARK) return (T)(0x8000u | (32768u - k));
        return (T)s.window[s.win_len - k];
    };
    // the rest of a match the last buffer could not take
    auto emit_match = [&](uint32_t len, uint32_t dist) -> bool {                        // false: the buffer is full (rest in pend_*)
        const uint64_t have = (uint64_t)(out - out_begin);
        if (dist > have) {
            // (part of) the source lies in earlier output: byte by byte through the saved window
            if ((uint64_t)dist - have > s.win_len) return true;                        // (checked by the caller; never here)
            while (len && out < out_end && (uint64_t)(out - out_begin) < dist) {
                *out = window_at(dist - (uint32_t)(out - out_begin));
                ++out;
                --len;
            }
        }
        if (len) {
            const uint64_t room = (uint64_t)(out_end - out);
            const uint32_t n = len < room ? len : (uint32_t)room;
            if ((uint64_t)(out - out_begin) >= dist) {
                if (room >= (uint64_t)n + 8) copy_match(out, dist, n);
                else
                    for (uint32_t k = 0; k < n; ++k) out[k] = out[(int64_t)k - (int64_t)dist];
                out += n;
                len -= n;
            }
        }
        s.pend_len = len;
        s.pend_dist = dist;
        return len == 0;
    };
    if (s.pend_len) {
        if (!emit_match(s.pend_len, s.pend_dist)) return done(NEED_OUTPUT);
    }
    for (;;) {
        if (s.phase == 3) return done(STREAM_END);
        if (s.phase == 0) {
            if (MARK && s.n_stops && s.over == 0) {
                const uint64_t at_bit = (uint64_t)(s.in - s.base) * 8u - (uint64_t)s.bitcnt;
                if (at_bit >= s.stops[0]) {
                    int lo = 0, hi = s.n_stops;                                        // the first stop >= at_bit
                    while (lo < hi) {
                        const int mid = (lo + hi) / 2;
                        if (s.stops[mid] < at_bit) lo = mid + 1;
                        else hi = mid;
                    }
                    if (lo < s.n_stops && s.stops[lo] == at_bit) {
                        s.stopped_at = at_bit;
                        return done(STOPPED);
                    }
                }
            }
            rd.refill();
            const uint32_t hdr = (uint32_t)s.bitbuf & 7u;
            s.bitbuf >>= 3;
            s.bitcnt -= 3;
            s.last = hdr & 1u;
            const int type = (int)(hdr >> 1);
            if (type == 3) return done(ERR_DATA);
            if (type == 0) {
                // stored: skip to the byte boundary, LEN / NLEN
                const int drop = s.bitcnt & 7;
                s.bitbuf >>= drop;
                s.bitcnt -= drop;
                rd.refill();
                const uint32_t lw = (uint32_t)s.bitbuf;
                s.bitbuf >>= 32;
                s.bitcnt -= 32;
                if (((lw & 0xFFFFu) ^ 0xFFFFu) != (lw >> 16)) return done(ERR_DATA);
                s.stored_left = lw & 0xFFFFu;
                // give the whole bytes of the bit buffer back to the input
                const int back = s.bitcnt >> 3;
                if (s.over > back) return done(ERR_INPUT);
                s.in -= back - s.over;
                s.over = 0;
                s.bitbuf = 0;
                s.bitcnt = 0;
                s.phase = 1;
            } else {
                if (type == 1) {
                    if (!s.tables_fixed) {
                        uint8_t lens[320];
                        for (int k = 0; k < 288; ++k) lens[k] = (uint8_t)(k < 144 ? 8 : k < 256 ? 9 : k < 280 ? 7 : 8);
                        for (int k = 0; k < 32; ++k) lens[288 + k] = 5;
                        if (!build(lens, 288, 1, LL_BITS, s.ll, LL_SIZE) || !build(lens + 288, 32, 2, D_BITS, s.d, D_SIZE)) return done(ERR_DATA);
                        s.tables_fixed = true;
                    }
                } else {
                    s.tables_fixed = false;
                    rd.refill();
                    const int hlit = (int)(s.bitbuf & 31u) + 257, hdist = (int)((s.bitbuf >> 5) & 31u) + 1, hclen = (int)((s.bitbuf >> 10) & 15u) + 4;
                    s.bitbuf >>= 14;
                    s.bitcnt -= 14;
                    if (hlit > 286 || hdist > 30) return done(ERR_DATA);
                    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    uint8_t cl[19] = {0};
                    for (int k = 0; k < hclen; ++k) {
                        if (s.bitcnt < 3) rd.refill();
                        cl[order[k]] = (uint8_t)(s.bitbuf & 7u);
                        s.bitbuf >>= 3;
                        s.bitcnt -= 3;
                    }
                    uint32_t pre[1 << 7];
                    if (!build(cl, 19, 0, 7, pre, 1 << 7)) return done(ERR_DATA);
                    uint8_t lens[320 + 140];
                    int k = 0;
                    while (k < hlit + hdist) {
                        rd.refill();
                        const uint32_t e = pre[s.bitbuf & 127u];
                        if (((e >> 12) & 15u) != K_LIT) return done(ERR_DATA);
                        s.bitbuf >>= (e & 255u);
                        s.bitcnt -= (int)(e & 255u);
                        const int sym = (int)(e >> 16);
                        if (sym < 16) lens[k++] = (uint8_t)sym;
                        else {
                            int rep, val = 0;
                            if (sym == 16) {
                                if (k == 0) return done(ERR_DATA);
                                val = lens[k - 1];
                                rep = 3 + (int)(s.bitbuf & 3u);
                                s.bitbuf >>= 2;
                                s.bitcnt -= 2;
                            } else if (sym == 17) {
                                rep = 3 + (int)(s.bitbuf & 7u);
                                s.bitbuf >>= 3;
                                s.bitcnt -= 3;
                            } else {
                                rep = 11 + (int)(s.bitbuf & 127u);
                                s.bitbuf >>= 7;
                                s.bitcnt -= 7;
                            }
                            if (k + rep > hlit + hdist) return done(ERR_DATA);
                            memset(lens + k, val, (size_t)rep);
                            k += rep;
                        }
                    }
                    if (lens[256] == 0) return done(ERR_DATA);
                    if (!build(lens, hlit, 1, LL_BITS, s.ll, LL_SIZE) || !build(lens + hlit, hdist, 2, D_BITS, s.d, D_SIZE)) return done(ERR_DATA);
                }
                s.phase = 2;
            }
            if (s.over > 8) return done(ERR_INPUT);
        }
        if (s.phase == 1) {
            uint64_t n = s.stored_left;
            if ((uint64_t)(s.in_end - s.in) < n) return done(ERR_INPUT);
            const uint64_t room = (uint64_t)(out_end - out);
            if (n > room) n = room;
            if (sizeof(T) == 1) memcpy(out, s.in, (size_t)n);
            else
                for (uint64_t k = 0; k < n; ++k) out[k] = (T)s.in[k];
            out += n;
            s.in += n;
            s.stored_left -= (uint32_t)n;
            if (s.stored_left) return done(NEED_OUTPUT);
            s.phase = s.last ? 3 : 0;
            continue;
        }
        // ---- a compressed block ----
        const uint32_t *const ll = s.ll, *const dt = s.d;
        for (;;) {
            // fast path: room for a whole match (+ the copy's over-write) and eight readable input bytes
            if ((uint64_t)(out_end - out) < 258 + 16 || s.in_end - s.in < 16) break;
            rd.refill();
            uint32_t e = ll[s.bitbuf & ((1u << LL_BITS) - 1u)];
            if (((e >> 12) & 15u) == K_SUB) {
                s.bitbuf >>= LL_BITS;
                s.bitcnt -= LL_BITS;
                e = ll[(e >> 16) + (uint32_t)(s.bitbuf & ((1u << ((e >> 8) & 15u)) - 1u))];
            }
            s.bitbuf >>= (e & 255u);
            s.bitcnt -= (int)(e & 255u);
            uint32_t kind = (e >> 12) & 15u;
            if (kind == K_LIT) {
                *out++ = (T)(e >> 16);
                // a second and a third literal out of the same refill (45 bits at most)
                e = ll[s.bitbuf & ((1u << LL_BITS) - 1u)];
                if (((e >> 12) & 15u) != K_LIT) continue;
                s.bitbuf >>= (e & 255u);
                s.bitcnt -= (int)(e & 255u);
                *out++ = (T)(e >> 16);
                e = ll[s.bitbuf & ((1u << LL_BITS) - 1u)];
                if (((e >> 12) & 15u) != K_LIT) continue;
                s.bitbuf >>= (e & 255u);
                s.bitcnt -= (int)(e & 255u);
                *out++ = (T)(e >> 16);
                continue;
            }
            if (kind != K_LEN) {
                if (kind == K_EOB) goto block_end;
                return done(ERR_DATA);
            }
            {
                const uint32_t xb = (e >> 8) & 15u;
                const uint32_t len = (e >> 16) + (uint32_t)(s.bitbuf & ((1u << xb) - 1u));
                s.bitbuf >>= xb;
                s.bitcnt -= (int)xb;
                uint32_t f = dt[s.bitbuf & ((1u << D_BITS) - 1u)];
                if (((f >> 12) & 15u) == K_SUB) {
                    s.bitbuf >>= D_BITS;
                    s.bitcnt -= D_BITS;
                    f = dt[(f >> 16) + (uint32_t)(s.bitbuf & ((1u << ((f >> 8) & 15u)) - 1u))];
                }
                if (((f >> 12) & 15u) != K_DIST) return done(ERR_DATA);
                s.bitbuf >>= (f & 255u);
                s.bitcnt -= (int)(f & 255u);
                const uint32_t db = (f >> 8) & 15u;
                const uint32_t dist = (f >> 16) + (uint32_t)(s.bitbuf & ((1u << db) - 1u));
                s.bitbuf >>= db;
                s.bitcnt -= (int)db;
                const uint64_t have = (uint64_t)(out - out_begin);
                if (dist <= have) {
                    copy_match(out, dist, len);
                    out += len;
                } else {
                    if ((uint64_t)dist - have > s.win_len) return done(ERR_DATA);      // further back than anything written
                    emit_match(len, dist);                                             // (room for the whole match: never leaves a rest)
                }
            }
        }
        // careful path: the end of the output buffer or of the input is near
        for (;;) {
            if (out == out_end) return done(NEED_OUTPUT);
            rd.refill();
            if (s.over > 8) return done(ERR_INPUT);
            uint32_t e = ll[s.bitbuf & ((1u << LL_BITS) - 1u)];
            if (((e >> 12) & 15u) == K_SUB) {
                s.bitbuf >>= LL_BITS;
                s.bitcnt -= LL_BITS;
                e = ll[(e >> 16) + (uint32_t)(s.bitbuf & ((1u << ((e >> 8) & 15u)) - 1u))];
            }
            s.bitbuf >>= (e & 255u);
            s.bitcnt -= (int)(e & 255u);
            const uint32_t kind = (e >> 12) & 15u;
            if (kind == K_LIT) {
                *out++ = (T)(e >> 16);
                continue;
            }
            if (kind == K_EOB) goto block_end;
            if (kind != K_LEN) return done(ERR_DATA);
            const uint32_t xb = (e >> 8) & 15u;
            const uint32_t len = (e >> 16) + (uint32_t)(s.bitbuf & ((1u << xb) - 1u));
            s.bitbuf >>= xb;
            s.bitcnt -= (int)xb;
            uint32_t f = dt[s.bitbuf & ((1u << D_BITS) - 1u)];
            if (((f >> 12) & 15u) == K_SUB) {
                s.bitbuf >>= D_BITS;
                s.bitcnt -= D_BITS;
                f = dt[(f >> 16) + (uint32_t)(s.bitbuf & ((1u << ((f >> 8) & 15u)) - 1u))];
            }
            if (((f >> 12) & 15u) != K_DIST) return done(ERR_DATA);
            s.bitbuf >>= (f & 255u);
            s.bitcnt -= (int)(f & 255u);
            const uint32_t db = (f >> 8) & 15u;
            const uint32_t dist = (f >> 16) + (uint32_t)(s.bitbuf & ((1u << db) - 1u));
            s.bitbuf >>= db;
            s.bitcnt -= (int)db;
            const uint64_t have = (uint64_t)(out - out_begin);
            if (dist > have && (uint64_t)dist - have > s.win_len) return done(ERR_DATA);
            if (!emit_match(len, dist)) return done(NEED_OUTPUT);
            // back to the fast path when there is room again
            if ((uint64_t)(out_end - out) >= 258 + 16 && s.in_end - s.in >= 16) break;
        }
        continue;
    block_end:
        if (s.over > 8) return done(ERR_INPUT);
        s.phase = s.last ? 3 : 0;
    }
}

static inline int inflate(State &s, uint8_t *out_begin, uint64_t cap, uint64_t *n_out) { return inflate_t<uint8_t, false>(s, out_begin, cap, n_out); }

// start decoding at bit `bit` of base[0 .. len) (a block header is expected there)
static inline void start_at(State &s, const uint8_t *base, uint64_t len, uint64_t bit) {
    s.base = base;
    s.in = base + (bit >> 3);
    s.in_end = base + len;
    s.bitbuf = 0; s.bitcnt = 0; s.over = 0; s.phase = 0; s.last = false; s.pend_len = 0; s.stored_left = 0; s.tables_fixed = false;
    if (bit & 7u) {
        Reader rd(s);
        rd.refill();
        s.bitbuf >>= (bit & 7u);
        s.bitcnt -= (int)(bit & 7u);
    }
}

// after STREAM_END: the first byte behind the deflate stream (the gzip trailer); false: the stream used bytes behind the input's end
static inline bool stream_tail(State &s, const uint8_t **p) {
    const int back = s.bitcnt >> 3;
    if (s.over > back) return false;
    *p = s.in - (back - s.over);
    s.over = 0;
    s.bitbuf = 0;
    s.bitcnt = 0;
    return *p <= s.in_end;
}

}  // namespace pgfi
