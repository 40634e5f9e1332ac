// A DEFLATE (RFC 1951) COMPRESSOR for BGZF members of `.geno` / CSV / VCF text -- the writer side of `-o out.geno.gz`
// (popgenWindows.py:316, `parseVCF.py ... | bgzip`, VCF_processing/README.md:33).  zlib's deflate at bgzip's default level 6 does
// 80 - 100 MB/s a thread on such text, and the VCF drop-in spent more than half of its 4.6 s waiting for it (profiles/r05,
// profiles/r06/vcf_bench_6GB.json).  Text written row by row repeats the rows above: this compressor tries the distance of the last
// match (one line back) first, then walks a chain of at most 24 earlier places with the same four-byte hash (6 once a match of 32
// bytes is in hand), compares eight bytes at a time, defers a match by one byte when the next position has a longer one (one step
// of lazy evaluation, for matches shorter than 16), and codes one dynamic Huffman block per member (length-limited codes from a
// two-queue merge over the sorted frequencies, the code lengths run-length coded as the format wants).  On `.geno` text: 2.5 x zlib's
// level-6 speed at 92 % of its ratio (22.98 : 1 against 25.08 : 1; chains of 48 / 96: 95 % / 99 % at 1.7 x / 1.3 x).  Its output is any
// inflater's input (tests: zlib, this library's decoders, k_inflate); it does not try to be zlib's bytes.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>

namespace pgfd {

struct BitOut {
    uint8_t *p, *end;
    uint64_t acc = 0;
    int n = 0;
    bool overflow = false;
    BitOut(uint8_t *out, size_t cap) : p(out), end(out + cap) {}
    inline void put(uint32_t v, int bits) {                  // bits <= 32
        acc |= (uint64_t)v << n;
        n += bits;
        while (n >= 8) {
            if (p < end) *p++ = (uint8_t)acc;
            else overflow = true;
            acc >>= 8;
            n -= 8;
        }
    }
    inline void flush() {
        if (n > 0) {
            if (p < end) *p++ = (uint8_t)acc;
            else overflow = true;
            acc = 0;
            n = 0;
        }
    }
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Tables {
    uint8_t len_sym[259];        // match length -> length symbol - 257
    uint8_t dist_sym_lo[512];    // distance - 1 < 512 -> distance symbol
    uint8_t dist_sym_hi[128];    // (distance - 1) >> 8 for larger ones
    Tables() {
        for (int s = 0; s < 29; ++s)
            for (int l = LEN_BASE[s]; l < (s == 28 ? 259 : LEN_BASE[s + 1]) && l <= 258; ++l) len_sym[l] = (uint8_t)s;
        len_sym[258] = 28;
        for (int s = 0; s < 30; ++s) {
            const int lo = DIST_BASE[s] - 1, hi = (s == 29 ? 32768 : DIST_BASE[s + 1] - 1);
            for (int d = lo; d < hi; ++d) {
                if (d < 512) dist_sym_lo[d] = (uint8_t)s;
                else dist_sym_hi[d >> 8] = (uint8_t)s;            // (from 512 on a symbol covers whole multiples of 256)
            }
        }
    }
    inline int dsym(uint32_t dist) const {
        const uint32_t d = dist - 1;
        return d < 512 ? dist_sym_lo[d] : dist_sym_hi[d >> 8];
    }
};
static const Tables TAB;

// code lengths (<= max_bits) for n symbols of frequency freq[]: a Huffman tree by the two-queue merge over the sorted leaves,
// then the lengths bounded the way miniz / zlib do it (longer codes folded into the bound, the Kraft sum repaired from the bottom)
static void code_lengths(const uint32_t *freq, int n, int max_bits, uint8_t *lens) {
    struct Leaf { uint32_t f; int s; };
    Leaf leaf[288];
    int m = 0;
    for (int s = 0; s < n; ++s) {
        lens[s] = 0;
        if (freq[s]) leaf[m++] = Leaf{freq[s], s};
    }
    if (m == 0) return;
    if (m == 1) { lens[leaf[0].s] = 1; return; }
    std::sort(leaf, leaf + m, [](const Leaf &a, const Leaf &b) { return a.f < b.f || (a.f == b.f && a.s < b.s); });
    // nodes: 0 .. m-1 leaves (sorted), m .. 2m-2 internal in the order they are made (their weights do not decrease)
    uint64_t w[576];
    int parent[576];
    for (int i = 0; i < m; ++i) w[i] = leaf[i].f;
    int a = 0, b = m, made = m;
    auto take = [&]() -> int {
        if (a < m && (b >= made || w[a] <= w[b])) return a++;
        return b++;
    };
    while (made < 2 * m - 1) {
        const int x = take(), y = take();
        w[made] = w[x] + w[y];
        parent[x] = parent[y] = made;
        ++made;
    }
    int depth[576];
    depth[2 * m - 2] = 0;
    for (int i = 2 * m - 3; i >= 0; --i) depth[i] = depth[parent[i]] + 1;
    int count[64] = {0};
    for (int i = 0; i < m; ++i) ++count[depth[i] < 63 ? depth[i] : 63];
    // bound the lengths
    for (int i = max_bits + 1; i < 64; ++i) {
        count[max_bits] += count[i];
        count[i] = 0;
    }
    uint64_t total = 0;
    for (int i = max_bits; i >= 1; --i) total += (uint64_t)count[i] << (max_bits - i);
    while (total != (1ull << max_bits)) {
        --count[max_bits];
        for (int i = max_bits - 1; i >= 1; --i)
            if (count[i]) {
                --count[i];
                count[i + 1] += 2;
                break;
            }
        --total;
    }
    // the shortest codes to the most frequent symbols (leaves are sorted by rising frequency: the longest lengths first)
    int at = 0;
    for (int l = max_bits; l >= 1; --l)
        for (int k = 0; k < count[l]; ++k) lens[leaf[at++].s] = (uint8_t)l;
}

// canonical codes, bit-reversed (deflate packs Huffman codes most significant bit first into a stream filled from the least)
static void make_codes(const uint8_t *lens, int n, uint16_t *codes) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    count[0] = 0;
    uint32_t next[16], code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) { codes[s] = 0; continue; }
        uint32_t c = next[l]++, r = 0;
        for (int k = 0; k < l; ++k) r |= ((c >> k) & 1u) << (l - 1 - k);
        codes[s] = (uint16_t)r;
    }
}

static inline uint32_t load32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

// length of the common prefix of a[0 ..) and b[0 ..), at most max_len (b > a; both readable for max_len bytes)
static inline uint32_t match_len(const uint8_t *a, const uint8_t *b, uint32_t max_len) {
    uint32_t n = 0;
    while (n + 8 <= max_len) {
        const uint64_t x = load64(a + n) ^ load64(b + n);
        if (x) return n + (uint32_t)(__builtin_ctzll(x) >> 3);
        n += 8;
    }
    while (n < max_len && a[n] == b[n]) ++n;
    return n;
}

struct Work {
    uint16_t head[1 << 15];      // four-byte hash -> position + 1 of its last occurrence in this member
    uint16_t prev[65536];        // position -> position + 1 of the one before it with the same hash (the chain)
    uint32_t tok[65536 + 8];     // literal: the byte; match: 0x80000000 | (length - 3) << 16 | (distance - 1)
};

// in[0 .. n), n <= 65535 -> one raw deflate stream (final block) at out; returns its length, 0 when cap is too small
static size_t deflate_member(const uint8_t *in, uint32_t n, uint8_t *out, size_t cap, Work &wk) {
    auto stored = [&]() -> size_t {
        if (cap < (size_t)n + 5) return 0;
        out[0] = 1;
        out[1] = (uint8_t)(n & 255); out[2] = (uint8_t)(n >> 8);
        out[3] = (uint8_t)(~n & 255); out[4] = (uint8_t)((~n >> 8) & 255);
        if (n) memcpy(out + 5, in, n);
        return (size_t)n + 5;
    };
    if (n < 16) return stored();
    // ---- matches ----
    memset(wk.head, 0, sizeof(wk.head));
    uint32_t freq_ll[288] = {0}, freq_d[32] = {0};
    uint32_t nt = 0, i = 0, last_dist = 0;
    const uint32_t last_start = n >= 8 ? n - 8 : 0;          // (matches start where eight bytes can still be loaded)
    auto hash = [](uint32_t v) -> uint32_t { return (v * 2654435761u) >> 17; };
    // the best match at position at_ (>= 4 bytes, else 0): one line back (the last match's distance) first, then down the chain of
    // earlier places with the same four-byte hash -- at most `depth` of them, fewer once a good match is in hand
    auto find = [&](uint32_t at_, uint32_t have, uint32_t *dist_out) -> uint32_t {
        const uint32_t maxl = std::min<uint32_t>(258, n - at_);
        const uint32_t v = load32(in + at_);
        uint32_t best = have, bdist = 0;
        auto consider = [&](uint32_t d) {
            if (load32(in + at_ - d) == v && in[at_ - d + best] == in[at_ + best]) {
                const uint32_t l = match_len(in + at_ - d, in + at_, maxl);
                if (l > best) { best = l; bdist = d; }
            }
        };
        if (last_dist && last_dist <= at_ && last_dist <= 32768) consider(last_dist);
        uint32_t c = wk.head[hash(v)];
        for (int depth = best >= 32 ? 6 : 24; c && depth > 0 && best < 128 && best < maxl; --depth) {
            const uint32_t d = at_ - (c - 1);
            if (d > 32768) break;                                    // (older places only further down)
            if (d != last_dist) consider(d);
            c = wk.prev[c - 1];
        }
        *dist_out = bdist;
        return bdist ? best : 0;
    };
    auto insert = [&](uint32_t at_) {
        const uint32_t h = hash(load32(in + at_));
        wk.prev[at_] = wk.head[h];
        wk.head[h] = (uint16_t)(at_ + 1);
    };
    uint32_t hashed = 0;                                     // positions < hashed are in the chains
    while (i < n) {
        uint32_t best = 0, bdist = 0;
        if (i < last_start) {
            while (hashed < i) insert(hashed++);
            best = find(i, 3, &bdist);
            insert(i);
            hashed = i + 1;
            // one step of lazy evaluation (zlib's levels 4+): a longer match one byte on wins, the byte in between goes out as a literal
            if (best && best < 16 && i + 1 < last_start) {
                uint32_t d2 = 0;
                const uint32_t l2 = find(i + 1, best, &d2);
                if (l2 > best) {
                    wk.tok[nt++] = in[i];
                    ++freq_ll[in[i]];
                    ++i;
                    insert(i);
                    hashed = i + 1;
                    best = l2;
                    bdist = d2;
                }
            }
        }
        if (best >= 4) {
            wk.tok[nt++] = 0x80000000u | ((best - 3) << 16) | (bdist - 1);
            ++freq_ll[257 + TAB.len_sym[best]];
            ++freq_d[TAB.dsym(bdist)];
            last_dist = bdist;
            i += best;
            // (the positions inside the match enter the chains when the next search needs them: `hashed`)
            if (i > last_start) hashed = i;
            else while (hashed < i) insert(hashed++);
        } else {
            wk.tok[nt++] = in[i];
            ++freq_ll[in[i]];
            ++i;
        }
    }
    freq_ll[256] = 1;
    // ---- codes ----
    uint8_t len_ll[288], len_d[32];
    uint16_t code_ll[288], code_d[32];
    code_lengths(freq_ll, 286, 15, len_ll);
    code_lengths(freq_d, 30, 15, len_d);
    {
        int used = 0;
        for (int s = 0; s < 286; ++s) used += len_ll[s] != 0;
        if (used < 2) len_ll[len_ll[0] ? 1 : 0] = 1;            // (a complete code needs two symbols)
        used = 0;
        for (int s = 0; s < 30; ++s) used += len_d[s] != 0;
        if (used == 0) len_d[0] = 1;                              // (no match at all: one unused distance code of one bit)
    }
    len_ll[286] = len_ll[287] = 0;
    make_codes(len_ll, 286, code_ll);
    make_codes(len_d, 30, code_d);
    int hlit = 286, hdist = 30;
    while (hlit > 257 && !len_ll[hlit - 1]) --hlit;
    while (hdist > 1 && !len_d[hdist - 1]) --hdist;
    // the code lengths, run-length coded (3.2.7)
    uint8_t seq[320];
    const int total = hlit + hdist;
    memcpy(seq, len_ll, (size_t)hlit);
    memcpy(seq + hlit, len_d, (size_t)hdist);
    uint8_t cl_sym[320], cl_extra[320];
    int ncl = 0;
    uint32_t freq_cl[19] = {0};
    for (int k = 0; k < total;) {
        int run = 1;
        while (k + run < total && seq[k + run] == seq[k]) ++run;
        if (seq[k] == 0 && run >= 3) {
            const int r = std::min(run, 138);
            cl_sym[ncl] = r <= 10 ? 17 : 18;
            cl_extra[ncl] = (uint8_t)(r <= 10 ? r - 3 : r - 11);
            ++freq_cl[cl_sym[ncl++]];
            k += r;
        } else if (seq[k] != 0 && run >= 4) {
            cl_sym[ncl] = seq[k];
            cl_extra[ncl] = 0;
            ++freq_cl[cl_sym[ncl++]];
            const int r = std::min(run - 1, 6);
            cl_sym[ncl] = 16;
            cl_extra[ncl] = (uint8_t)(r - 3);
            ++freq_cl[16];
            ++ncl;
            k += 1 + r;
        } else {
            cl_sym[ncl] = seq[k];
            cl_extra[ncl] = 0;
            ++freq_cl[cl_sym[ncl++]];
            ++k;
        }
    }
    uint8_t len_cl[19];
    uint16_t code_cl[19];
    code_lengths(freq_cl, 19, 7, len_cl);
    {
        int used = 0;
        for (int s = 0; s < 19; ++s) used += len_cl[s] != 0;
        if (used < 2) len_cl[len_cl[0] ? 1 : 0] = 1;            // (the code-length code must be complete)
    }
    make_codes(len_cl, 19, code_cl);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && !len_cl[order[hclen - 1]]) --hclen;
    // ---- bits ----
    BitOut bo(out, cap);
    bo.put(1u | (2u << 1), 3);                                   // BFINAL, dynamic
    bo.put((uint32_t)(hlit - 257), 5);
    bo.put((uint32_t)(hdist - 1), 5);
    bo.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; ++k) bo.put(len_cl[order[k]], 3);
    for (int k = 0; k < ncl; ++k) {
        const int s = cl_sym[k];
        bo.put(code_cl[s], len_cl[s]);
        if (s == 16) bo.put(cl_extra[k], 2);
        else if (s == 17) bo.put(cl_extra[k], 3);
        else if (s == 18) bo.put(cl_extra[k], 7);
    }
    for (uint32_t k = 0; k < nt; ++k) {
        const uint32_t t = wk.tok[k];
        if (!(t & 0x80000000u)) {
            bo.put(code_ll[t], len_ll[t]);
        } else {
            const uint32_t len = ((t >> 16) & 0x7FFFu) + 3, dist = (t & 0xFFFFu) + 1;
            const int ls = TAB.len_sym[len], ds = TAB.dsym(dist);
            bo.put(code_ll[257 + ls], len_ll[257 + ls]);
            if (LEN_EXTRA[ls]) bo.put(len - LEN_BASE[ls], LEN_EXTRA[ls]);
            bo.put(code_d[ds], len_d[ds]);
            if (DIST_EXTRA[ds]) bo.put(dist - DIST_BASE[ds], DIST_EXTRA[ds]);
        }
    }
    bo.put(code_ll[256], len_ll[256]);
    bo.flush();
    const size_t produced = (size_t)(bo.p - out);
    if (bo.overflow || produced >= (size_t)n + 5) return stored();
    return produced;
}

}  // namespace pgfd
