// DEFLATE on the device: text that lies in HBM -> BGZF members (`-o out.geno.gz` of the VCF drop-in, whose rows are made on the device:
// `parseVCF.py ... | bgzip`, VCF_processing/README.md:33; popgenWindows.py:316) -- the counterpart of k_inflate.  The host's
// compressor (pg_fast_deflate.h) was what the drop-in waited for once the device parsed the VCF (1.5 of 2.2 s on 16 CPUs).
//
//   k_deflate        a wavefront per member of 65 280 bytes of text (persistent: a wave takes member after member).
//     matches        64 positions at a time, a lane each: the four bytes at the position are hashed into one of 256 buckets of the TWO
//                    most recent places with that hash, its twelve bytes into one of 512 buckets of the EIGHT most recent (LDS, 9 KB a
//                    wave; why two tables: see DF_WA / DF_WB below) -- plus the nearest earlier lane of the window in its 4-byte bucket;
//                    every candidate is compared over sixteen bytes (two 8-byte loads from the text in HBM / L2), the best one kept.
//                    Then the window is walked from its first uncovered position: a literal run up to the next lane with a match goes
//                    out in one step (all lanes), a match of the full sixteen bytes is first extended by the WHOLE wave (lane k
//                    compares dword k of candidate and text: up to 256 bytes in one step).  The window's positions enter their buckets
//                    in order without a serial loop: a lane's rank among the lanes of its bucket (a bit mask per bucket in LDS) is its slot.
//     codes          one dynamic Huffman block per member: the frequencies are counted in LDS while the tokens are made; lane 0
//                    builds the length-limited codes and the block header as the host's compressor does (two-queue merge, miniz-style
//                    bound, run-length coded code lengths); a member whose coded size would reach its text's is stored.
//     bits           64 tokens at a time: code + extra bits of a token form one value of up to 48 bits, a wave scan of the bit counts
//                    gives its place, the lanes OR their pieces into a window in LDS, whole dwords leave for HBM.
//   k_crc32_pieces   (pg_inflate.hip) the members' CRC-32
//   k_bgzf_assemble  gzip header + BC field, the deflated bytes, CRC-32 and size of every member, one behind the other
// Its output is any inflater's input (tests: zlib, k_inflate); it does not try to be zlib's bytes.
#include "pg_ctx.h"

#include <algorithm>
#include <vector>

int pg_launch_crc32_pieces(pg_ctx *c, hipStream_t st, const uint8_t *text, const long long *total_d, uint32_t piece, int64_t max_pieces,
                           uint32_t *crc_out);

namespace {

// Two tables of recent places.  Text written row by row repeats a four-byte context so often that the last places of its hash all lie in
// the same row; a twelve-byte context recurs seldom enough for its last eight places to reach the rows above.  Measured on 326 MB of
// `.geno` rows against zlib level 6's bytes (profiles/r06/deflate_bench_table_depths.txt; correlated rows in brackets): eight places
// per 4-byte hash alone 1.115 x (1.155 x) in 13.6 ms; two per 4-byte + four per 12-byte hash 1.073 x (1.13 x) in 13.7 ms; two + eight
// 1.048 x (1.10 x) in 20.9 ms -- the default; four + eight 1.042 x in 27.8 ms.  A CPU prototype of the wave's algorithm settled the
// shape first (racy insertion: -15 %; a row-above or last-distance candidate, lazy evaluation: nothing).
#ifndef PGD_WA
#define PGD_WA 2
#endif
#ifndef PGD_WB
#define PGD_WB 8
#endif
constexpr int DF_WA = PGD_WA, DF_HBA = 8;     // places per bucket (2 or 4) / log2 buckets of the 4-byte contexts
constexpr int DF_WB = PGD_WB, DF_HBB = 9;     // ... (4 or 8) of the 12-byte contexts
constexpr int DF_NC = DF_WA + DF_WB + 1;      // candidates of a position: the buckets' places + the nearest earlier lane of its 4-byte group
#ifndef PGD_WAVES
#define PGD_WAVES 3
#endif
constexpr uint32_t DF_PIECE = 65280;          // text per member (bgzip's)
constexpr uint32_t DF_SLOT = 65536;           // bytes a member's deflate stream may take (stored: text + 5)
constexpr uint32_t DF_MAXL = 256;             // longest match (the format's 258 would need a 65th dword in the wave's compare)

// lane 0's work space for the codes, the header's bits and the window of bits on their way out: all of it after the matches, in the
// memory the buckets leave behind
struct DfWork {
    uint32_t win[112];                        // the bits of a batch of tokens on their way out
    uint32_t hdr[192];                        // the block header's bits
    uint32_t w[576];
    uint16_t parent[576], sym[288];
    uint8_t depth[576];
    uint8_t seq[320], cl_sym[320], cl_extra[320];
    uint32_t freq_cl[19];
    uint8_t len_cl[19];
    uint16_t code_cl[19];
    int count[64];
    uint32_t next[16];
    uint16_t code_ll[288], code_d[32];        // the block's codes (bit-reversed) and their lengths
    uint8_t len_ll[288], len_d[32];
};

struct DfMatch {
    uint32_t bucketA[(1 << DF_HBA) * DF_WA / 2];   // the most recent places (+ 1, 16 bits each) of every hash
    uint32_t bucketB[(1 << DF_HBB) * DF_WB / 2];
    uint32_t grp[(1 << DF_HBB) * 2];               // which lanes of the window at hand hash into a bucket (a bit per lane); table A's, then table B's
};

struct DfShared {
    union {
        DfMatch m;
        DfWork k;
    } u;
    uint32_t freq_ll[288], freq_d[32];
};

__device__ inline uint64_t ld64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ inline uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ inline uint32_t rl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ inline uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// length 3 .. 257 -> (index of its length symbol, number of extra bits, their value); distance likewise
__device__ inline void len_code(uint32_t len, uint32_t *idx, uint32_t *eb, uint32_t *ev) {
    const uint32_t l3 = len - 3;
    if (l3 < 8) { *idx = l3; *eb = 0; *ev = 0; return; }
    const uint32_t nb = 31u - (uint32_t)__clz((int)l3), e = nb - 2;
    *idx = 8 + 4 * (e - 1) + ((l3 - (1u << nb)) >> e);
    *eb = e;
    *ev = l3 & ((1u << e) - 1u);
}
__device__ inline void dist_code(uint32_t dist, uint32_t *idx, uint32_t *eb, uint32_t *ev) {
    const uint32_t d1 = dist - 1;
    if (d1 < 4) { *idx = d1; *eb = 0; *ev = 0; return; }
    const uint32_t nb = 31u - (uint32_t)__clz((int)d1), e = nb - 1;
    *idx = 2 * nb + ((d1 >> e) & 1u);
    *eb = e;
    *ev = d1 & ((1u << e) - 1u);
}
__device__ inline uint32_t len_extra_of(int idx) { return idx < 8 || idx >= 28 ? 0u : (uint32_t)((idx - 4) >> 2); }
__device__ inline uint32_t dist_extra_of(int idx) { return idx < 4 ? 0u : (uint32_t)((idx - 2) >> 1); }

// ---- lane 0: the codes (pg_fast_deflate.h's code_lengths / make_codes, with their work arrays in LDS) ----
__device__ void df_code_lengths(const uint32_t *freq, int n, int max_bits, uint8_t *lens, DfShared &sh) {
    int m = 0;
    for (int s = 0; s < n; ++s) {
        lens[s] = 0;
        if (freq[s]) {
            // insertion into the leaves sorted by (frequency, symbol)
            int k = m++;
            const uint32_t f = freq[s];
            while (k > 0 && sh.u.k.w[k - 1] > f) {
                sh.u.k.w[k] = sh.u.k.w[k - 1];
                sh.u.k.sym[k] = sh.u.k.sym[k - 1];
                --k;
            }
            sh.u.k.w[k] = f;
            sh.u.k.sym[k] = (uint16_t)s;
        }
    }
    if (m == 0) return;
    if (m == 1) { lens[sh.u.k.sym[0]] = 1; return; }
    int a = 0, b = m, made = m;
    while (made < 2 * m - 1) {
        int x, y;
        if (a < m && (b >= made || sh.u.k.w[a] <= sh.u.k.w[b])) x = a++; else x = b++;
        if (a < m && (b >= made || sh.u.k.w[a] <= sh.u.k.w[b])) y = a++; else y = b++;
        sh.u.k.w[made] = sh.u.k.w[x] + sh.u.k.w[y];
        sh.u.k.parent[x] = sh.u.k.parent[y] = (uint16_t)made;
        ++made;
    }
    for (int i = 0; i < 64; ++i) sh.u.k.count[i] = 0;
    sh.u.k.depth[2 * m - 2] = 0;
    for (int i = 2 * m - 3; i >= 0; --i) {
        const int d = sh.u.k.depth[sh.u.k.parent[i]] + 1;
        sh.u.k.depth[i] = (uint8_t)(d < 63 ? d : 63);
        if (i < m) ++sh.u.k.count[sh.u.k.depth[i]];
    }
    for (int i = max_bits + 1; i < 64; ++i) {
        sh.u.k.count[max_bits] += sh.u.k.count[i];
        sh.u.k.count[i] = 0;
    }
    unsigned long long total = 0;
    for (int i = max_bits; i >= 1; --i) total += (unsigned long long)sh.u.k.count[i] << (max_bits - i);
    while (total != (1ull << max_bits)) {
        --sh.u.k.count[max_bits];
        for (int i = max_bits - 1; i >= 1; --i)
            if (sh.u.k.count[i]) {
                --sh.u.k.count[i];
                sh.u.k.count[i + 1] += 2;
                break;
            }
        --total;
    }
    int at = 0;
    for (int l = max_bits; l >= 1; --l)
        for (int k = 0; k < sh.u.k.count[l]; ++k) lens[sh.u.k.sym[at++]] = (uint8_t)l;
}

__device__ void df_make_codes(const uint8_t *lens, int n, uint16_t *codes, DfShared &sh) {
    for (int l = 0; l < 16; ++l) sh.u.k.count[l] = 0;
    for (int s = 0; s < n; ++s) ++sh.u.k.count[lens[s]];
    sh.u.k.count[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (uint32_t)sh.u.k.count[l - 1]) << 1;
        sh.u.k.next[l] = code;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) { codes[s] = 0; continue; }
        const uint32_t c = sh.u.k.next[l]++;
        codes[s] = (uint16_t)(__brev(c) >> (32 - l));
    }
}

struct HdrBits {
    uint32_t *p;
    unsigned long long acc;
    int n, words;
    __device__ void put(uint32_t v, int bits) {
        acc |= (unsigned long long)v << n;
        n += bits;
        if (n >= 32) {
            p[words++] = (uint32_t)acc;
            acc >>= 32;
            n -= 32;
        }
    }
};

// lane 0: codes from the frequencies, the block header's bits into sh.u.k.hdr; returns the bits of the header, *total_bits the whole stream's
__device__ uint32_t df_codes_and_header(DfShared &sh, unsigned long long *total_bits) {
    sh.freq_ll[256] = 1;
    df_code_lengths(sh.freq_ll, 286, 15, sh.u.k.len_ll, sh);
    df_code_lengths(sh.freq_d, 30, 15, sh.u.k.len_d, sh);
    {
        int used = 0;
        for (int s = 0; s < 286; ++s) used += sh.u.k.len_ll[s] != 0;
        if (used < 2) sh.u.k.len_ll[sh.u.k.len_ll[0] ? 1 : 0] = 1;      // (a complete code needs two symbols)
        used = 0;
        for (int s = 0; s < 30; ++s) used += sh.u.k.len_d[s] != 0;
        if (used == 0) sh.u.k.len_d[0] = 1;                          // (no match at all: one unused distance code of one bit)
    }
    sh.u.k.len_ll[286] = sh.u.k.len_ll[287] = 0;
    df_make_codes(sh.u.k.len_ll, 286, sh.u.k.code_ll, sh);
    df_make_codes(sh.u.k.len_d, 30, sh.u.k.code_d, sh);
    int hlit = 286, hdist = 30;
    while (hlit > 257 && !sh.u.k.len_ll[hlit - 1]) --hlit;
    while (hdist > 1 && !sh.u.k.len_d[hdist - 1]) --hdist;
    const int total = hlit + hdist;
    for (int k = 0; k < hlit; ++k) sh.u.k.seq[k] = sh.u.k.len_ll[k];
    for (int k = 0; k < hdist; ++k) sh.u.k.seq[hlit + k] = sh.u.k.len_d[k];
    int ncl = 0;
    for (int s = 0; s < 19; ++s) sh.u.k.freq_cl[s] = 0;
    for (int k = 0; k < total;) {                               // the code lengths, run-length coded (RFC 1951, 3.2.7)
        int run = 1;
        while (k + run < total && sh.u.k.seq[k + run] == sh.u.k.seq[k]) ++run;
        if (sh.u.k.seq[k] == 0 && run >= 3) {
            const int r = run < 138 ? run : 138;
            sh.u.k.cl_sym[ncl] = (uint8_t)(r <= 10 ? 17 : 18);
            sh.u.k.cl_extra[ncl] = (uint8_t)(r <= 10 ? r - 3 : r - 11);
            ++sh.u.k.freq_cl[sh.u.k.cl_sym[ncl++]];
            k += r;
        } else if (sh.u.k.seq[k] != 0 && run >= 4) {
            sh.u.k.cl_sym[ncl] = sh.u.k.seq[k];
            sh.u.k.cl_extra[ncl] = 0;
            ++sh.u.k.freq_cl[sh.u.k.cl_sym[ncl++]];
            const int r = run - 1 < 6 ? run - 1 : 6;
            sh.u.k.cl_sym[ncl] = 16;
            sh.u.k.cl_extra[ncl] = (uint8_t)(r - 3);
            ++sh.u.k.freq_cl[16];
            ++ncl;
            k += 1 + r;
        } else {
            sh.u.k.cl_sym[ncl] = sh.u.k.seq[k];
            sh.u.k.cl_extra[ncl] = 0;
            ++sh.u.k.freq_cl[sh.u.k.cl_sym[ncl++]];
            ++k;
        }
    }
    df_code_lengths(sh.u.k.freq_cl, 19, 7, sh.u.k.len_cl, sh);
    {
        int used = 0;
        for (int s = 0; s < 19; ++s) used += sh.u.k.len_cl[s] != 0;
        if (used < 2) sh.u.k.len_cl[sh.u.k.len_cl[0] ? 1 : 0] = 1;
    }
    df_make_codes(sh.u.k.len_cl, 19, sh.u.k.code_cl, sh);
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && !sh.u.k.len_cl[order[hclen - 1]]) --hclen;
    HdrBits hb{sh.u.k.hdr, 0ull, 0, 0};
    hb.put(1u | (2u << 1), 3);                                   // BFINAL, dynamic
    hb.put((uint32_t)(hlit - 257), 5);
    hb.put((uint32_t)(hdist - 1), 5);
    hb.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; ++k) hb.put(sh.u.k.len_cl[order[k]], 3);
    for (int k = 0; k < ncl; ++k) {
        const int s = sh.u.k.cl_sym[k];
        hb.put(sh.u.k.code_cl[s], sh.u.k.len_cl[s]);
        if (s == 16) hb.put(sh.u.k.cl_extra[k], 2);
        else if (s == 17) hb.put(sh.u.k.cl_extra[k], 3);
        else if (s == 18) hb.put(sh.u.k.cl_extra[k], 7);
    }
    const uint32_t hdr_bits = (uint32_t)hb.words * 32u + (uint32_t)hb.n;
    sh.u.k.hdr[hb.words] = (uint32_t)hb.acc;                         // the unfinished dword
    unsigned long long bits = hdr_bits;
    for (int s = 0; s < 286; ++s) bits += (unsigned long long)sh.freq_ll[s] * (sh.u.k.len_ll[s] + (s >= 257 ? len_extra_of(s - 257) : 0u));
    for (int s = 0; s < 30; ++s) bits += (unsigned long long)sh.freq_d[s] * (sh.u.k.len_d[s] + dist_extra_of(s));
    *total_bits = bits;
    return hdr_bits;
}

__device__ inline int wave_incl_scan_u(int x, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// in[0 .. n) -> one raw deflate stream (final block) at out (DF_SLOT bytes); returns its length (every lane)
__device__ uint32_t df_member(const uint8_t *__restrict__ in, uint32_t n, uint8_t *__restrict__ out, uint32_t *__restrict__ tok, DfShared &sh, int lane) {
    auto stored = [&]() -> uint32_t {
        if (lane == 0) {
            out[0] = 1;
            out[1] = (uint8_t)(n & 255); out[2] = (uint8_t)(n >> 8);
            out[3] = (uint8_t)(~n & 255); out[4] = (uint8_t)((~n >> 8) & 255);
        }
        for (uint32_t k = (uint32_t)lane; k < n; k += 64) out[5 + k] = in[k];
        return n + 5;
    };
    if (n < 16) return stored();
    for (int k = lane; k < (int)(sizeof(DfMatch) / 4); k += 64) reinterpret_cast<uint32_t *>(&sh.u.m)[k] = 0;
    for (int k = lane; k < 288; k += 64) sh.freq_ll[k] = 0;
    if (lane < 32) sh.freq_d[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- matches ----
    uint32_t nt = 0, pos = 0;
    for (uint32_t w0 = 0; w0 < n; w0 += 64) {
        const uint32_t p = w0 + (uint32_t)lane;
        const bool inb = p + 4 <= n, inbB = p + 12 <= n;
        const uint32_t maxl = inb ? (n - p < DF_MAXL ? n - p : DF_MAXL) : 0u;
        uint64_t a0 = 0, a1 = 0;
        if (p < n) { a0 = ld64(in + p); a1 = ld64(in + p + 8); }          // (the text's buffer is readable 32 bytes past its end)
        const uint32_t v = (uint32_t)a0;
        const uint32_t hA = (v * 2654435761u) >> (32 - DF_HBA);
        const uint32_t hB = (uint32_t)(((a0 * 0x9E3779B97F4A7C15ull) ^ ((a1 & 0xffffffffull) * 0xC2B2AE3D27D4EB4Full)) >> (64 - DF_HBB));
        uint32_t bkA[DF_WA / 2], bkB[DF_WB / 2];                   // (two places a word)
#pragma unroll
        for (int k = 0; k < DF_WA / 2; ++k) bkA[k] = inb ? sh.u.m.bucketA[hA * (DF_WA / 2) + k] : 0u;
#pragma unroll
        for (int k = 0; k < DF_WB / 2; ++k) bkB[k] = inbB ? sh.u.m.bucketB[hB * (DF_WB / 2) + k] : 0u;
        // the lanes of the window that share this lane's bucket, as a bit per lane: every lane sets its bit in the bucket's mask (LDS),
        // reads the mask back and clears it again -- a lane's rank in its group, the group's size and the nearest earlier lane of it
        // (a candidate like the bucket's: its bytes are compared) are three bit counts (the first version asked all 64 lanes for their
        // hash, one v_readlane pair after the other: a third of the kernel's instructions).  Table A, then table B in the same words.
        const unsigned long long lt = (1ull << lane) - 1ull;
        unsigned long long gmA = 0, gmB = 0;
        if (inb) atomicOr(&sh.u.m.grp[2 * hA + ((uint32_t)lane >> 5)], 1u << ((uint32_t)lane & 31u));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (inb) gmA = (unsigned long long)sh.u.m.grp[2 * hA] | ((unsigned long long)sh.u.m.grp[2 * hA + 1] << 32);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (inb) sh.u.m.grp[2 * hA + ((uint32_t)lane >> 5)] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (inbB) atomicOr(&sh.u.m.grp[2 * hB + ((uint32_t)lane >> 5)], 1u << ((uint32_t)lane & 31u));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (inbB) gmB = (unsigned long long)sh.u.m.grp[2 * hB] | ((unsigned long long)sh.u.m.grp[2 * hB + 1] << 32);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (inbB) sh.u.m.grp[2 * hB + ((uint32_t)lane >> 5)] = 0;
        const int rankA = __popcll(gmA & lt), gsizeA = __popcll(gmA), rankB = __popcll(gmB & lt), gsizeB = __popcll(gmB);
        const int close = (gmA & lt) ? 63 - __clzll((long long)(gmA & lt)) : -1;
        const uint32_t wend = w0 + 64 < n ? w0 + 64 : n;
        const bool need = pos < wend;                             // (a window inside a long match only enters its buckets)
        uint32_t bestL = 0, bestD = 0, capmask = 0;
        uint32_t q1s[DF_NC];
#pragma unroll
        for (int k = 0; k < DF_WA; ++k) q1s[k] = k & 1 ? bkA[k / 2] >> 16 : bkA[k / 2] & 0xffffu;
#pragma unroll
        for (int k = 0; k < DF_WB; ++k) q1s[DF_WA + k] = k & 1 ? bkB[k / 2] >> 16 : bkB[k / 2] & 0xffffu;
        q1s[DF_NC - 1] = close >= 0 ? w0 + (uint32_t)close + 1u : 0u;
        // two rounds of loads, the candidates' side by side (one after the other they cost a member 8.7 ms: eighteen trips to
        // L2 / HBM per window in a row; profiles/r06/vcf_gz_to_gz_kernel_stats_first.csv)
        bool ok[DF_NC];
        uint32_t qq[DF_NC];
        uint64_t m0[DF_NC], m1[DF_NC];
        const uint32_t safe = p < n ? p : 0u;                        // (a lane without a candidate reads its own place: no branch around a load)
#pragma unroll
        for (int c = 0; c < DF_NC; ++c) {
            ok[c] = need && inb && q1s[c] != 0 && p - (q1s[c] - 1) <= 32768u;
            qq[c] = ok[c] ? q1s[c] - 1 : safe;
        }
        if (need) {
#pragma unroll
            for (int c = 0; c < DF_NC; ++c) m0[c] = ld64(in + qq[c]);
#pragma unroll
            for (int c = 0; c < DF_NC; ++c) m0[c] ^= a0;
#pragma unroll
            for (int c = 0; c < DF_NC; ++c) m1[c] = ld64(in + qq[c] + 8);
#pragma unroll
            for (int c = 0; c < DF_NC; ++c) m1[c] ^= a1;
        } else {
#pragma unroll
            for (int c = 0; c < DF_NC; ++c) m0[c] = m1[c] = ~0ull;
        }
#pragma unroll
        for (int c = 0; c < DF_NC; ++c) {
            if (ok[c]) {
                const uint32_t d = p - (q1s[c] - 1);
                uint32_t L;
                if (m0[c]) L = (uint32_t)(__ffsll((long long)m0[c]) - 1) >> 3;
                else L = m1[c] ? 8u + ((uint32_t)(__ffsll((long long)m1[c]) - 1) >> 3) : 16u;
                if (L > maxl) L = maxl;
                if (L >= 4 && (L > bestL || (L == bestL && d < bestD))) { bestL = L; bestD = d; }
                if (L == 16 && maxl > 16) capmask |= 1u << c;
            }
        }
        // every lane's match as the token it would be, and the symbols it would count (the walk below only picks)
        uint32_t tokw = 0, lsym = 0, dsym = 0;
        if (bestL >= 4) {
            uint32_t eb, ev;
            tokw = 0x80000000u | ((bestL - 3) << 16) | (bestD - 1);
            len_code(bestL, &lsym, &eb, &ev);
            dist_code(bestD, &dsym, &eb, &ev);
        }
        // ---- the window's tokens, from its first uncovered position ----
        while (pos < wend) {
            const int l = (int)(pos - w0);
            const uint32_t L = rl(bestL, l);
            if (L < 4) {
                const unsigned long long mm = __builtin_amdgcn_ballot_w64(bestL >= 4) >> l;
                uint32_t run = mm ? (uint32_t)(__ffsll((long long)mm) - 1) : 64u;
                if (run > wend - pos) run = wend - pos;
                if ((uint32_t)lane >= (uint32_t)l && (uint32_t)lane < (uint32_t)l + run) {
                    const uint32_t byte = (uint32_t)(a0 & 0xffu);
                    tok[nt + (uint32_t)(lane - l)] = byte;
                    atomicAdd(&sh.freq_ll[byte], 1u);
                }
                nt += run;
                pos += run;
            } else {
                uint32_t mL = L;
                const uint32_t caps = rl(capmask, l);
                if (!caps) {
                    if (lane == 0) {
                        tok[nt] = rl(tokw, l);
                        atomicAdd(&sh.freq_ll[257 + rl(lsym, l)], 1u);
                        atomicAdd(&sh.freq_d[rl(dsym, l)], 1u);
                    }
                } else {
                    // the candidates that matched all sixteen bytes: the whole wave compares 256 bytes of each
                    uint32_t mD = rl(bestD, l);
                    const uint32_t lim = rl(maxl, l);
                    const bool cmp = 4u * (uint32_t)lane < lim;    // (nothing is read behind the member's last dword)
                    uint32_t mine = 0;
                    if (cmp) mine = ld32(in + pos + 4u * (uint32_t)lane);
                    mL = 0;
                    for (uint32_t cm = caps; cm; cm &= cm - 1) {
                        const int c = __ffs((int)cm) - 1;
                        uint32_t q1 = 0;
#pragma unroll
                        for (int k = 0; k < DF_NC; ++k)
                            if (k == c) q1 = rl(q1s[k], l);
                        const uint32_t q = q1 - 1, d = pos - q;
                        uint32_t x = 1;
                        if (cmp) x = ld32(in + q + 4u * (uint32_t)lane) ^ mine;
                        const unsigned long long ne = __builtin_amdgcn_ballot_w64(x != 0);
                        uint32_t len = 256;
                        if (ne) {
                            const int k0 = __ffsll((long long)ne) - 1;
                            len = 4u * (uint32_t)k0 + ((uint32_t)(__ffs((int)rl(x, k0)) - 1) >> 3);
                        }
                        if (len > lim) len = lim;
                        if (len > mL || (len == mL && d < mD)) { mL = len; mD = d; }
                    }
                    if (lane == 0) {
                        tok[nt] = 0x80000000u | ((mL - 3) << 16) | (mD - 1);
                        uint32_t idx, eb, ev;
                        len_code(mL, &idx, &eb, &ev);
                        atomicAdd(&sh.freq_ll[257 + idx], 1u);
                        dist_code(mD, &idx, &eb, &ev);
                        atomicAdd(&sh.freq_d[idx], 1u);
                    }
                }
                ++nt;
                pos += mL;
            }
        }
        // ---- the window's positions into their buckets, most recent first: a lane's rank among the lanes of its bucket is its slot ----
        if (inb) {
            uint16_t *b16 = reinterpret_cast<uint16_t *>(&sh.u.m.bucketA[hA * (DF_WA / 2)]);
            const int slot = gsizeA - 1 - rankA;
            if (slot < DF_WA) b16[slot] = (uint16_t)(p + 1);
            if (rankA == gsizeA - 1 && gsizeA < DF_WA) {             // the group's last lane moves the old places back
#pragma unroll
                for (int k = 0; k < DF_WA; ++k)
                    if (k + gsizeA < DF_WA) b16[k + gsizeA] = (uint16_t)q1s[k];
            }
        }
        if (inbB) {
            uint16_t *b16 = reinterpret_cast<uint16_t *>(&sh.u.m.bucketB[hB * (DF_WB / 2)]);
            const int slot = gsizeB - 1 - rankB;
            if (slot < DF_WB) b16[slot] = (uint16_t)(p + 1);
            if (rankB == gsizeB - 1 && gsizeB < DF_WB) {
#pragma unroll
                for (int k = 0; k < DF_WB; ++k)
                    if (k + gsizeB < DF_WB) b16[k + gsizeB] = (uint16_t)q1s[DF_WA + k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0) tok[nt] = 256;                                // end of block, a token like the others
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- codes ----
    unsigned long long total_bits = 0;
    uint32_t hdr_bits = 0;
    if (lane == 0) hdr_bits = df_codes_and_header(sh, &total_bits);
    hdr_bits = rfl(hdr_bits);
    const uint32_t tb_lo = rfl((uint32_t)total_bits), tb_hi = rfl((uint32_t)(total_bits >> 32));
    total_bits = ((unsigned long long)tb_hi << 32) | tb_lo;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if ((total_bits + 7) / 8 >= (unsigned long long)n + 5) return stored();
    // ---- bits ----
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    uint32_t wbase = hdr_bits >> 5;                              // dwords of the stream that have left
    for (uint32_t k = (uint32_t)lane; k < wbase; k += 64) out32[k] = sh.u.k.hdr[k];
    for (int k = lane; k < 112; k += 64) sh.u.k.win[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) sh.u.k.win[0] = sh.u.k.hdr[wbase];
    uint32_t obit = hdr_bits;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (uint32_t k0 = 0; k0 <= nt; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane;
        unsigned long long val = 0;
        int nb = 0;
        if (k <= nt) {
            const uint32_t t = tok[k];
            if (!(t & 0x80000000u)) {
                val = sh.u.k.code_ll[t];
                nb = sh.u.k.len_ll[t];
            } else {
                const uint32_t len = ((t >> 16) & 0x7fffu) + 3, dist = (t & 0xffffu) + 1;
                uint32_t li, leb, lev, di, deb, dev;
                len_code(len, &li, &leb, &lev);
                dist_code(dist, &di, &deb, &dev);
                val = sh.u.k.code_ll[257 + li];
                nb = sh.u.k.len_ll[257 + li];
                val |= (unsigned long long)lev << nb;
                nb += (int)leb;
                val |= (unsigned long long)sh.u.k.code_d[di] << nb;
                nb += sh.u.k.len_d[di];
                val |= (unsigned long long)dev << nb;
                nb += (int)deb;
            }
        }
        const int incl = wave_incl_scan_u(nb, lane);
        const uint32_t total = (uint32_t)__shfl(incl, 63, 64);
        if (nb) {
            const uint32_t rel = obit + (uint32_t)(incl - nb) - wbase * 32u;
            const uint32_t wi = rel >> 5, sft = rel & 31u;
            atomicOr(&sh.u.k.win[wi], (uint32_t)(val << sft));
            if (sft + (uint32_t)nb > 32) atomicOr(&sh.u.k.win[wi + 1], (uint32_t)(sft ? val >> (32 - sft) : val >> 32));
            if (sft + (uint32_t)nb > 64) atomicOr(&sh.u.k.win[wi + 2], (uint32_t)(val >> (64 - sft)));
        }
        obit += total;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t full = (obit >> 5) - wbase;               // whole dwords of the window
        uint32_t carry = 0, mine0 = 0, mine1 = 0;
        if ((uint32_t)lane < full) mine0 = sh.u.k.win[lane];
        if ((uint32_t)lane + 64 < full) mine1 = sh.u.k.win[lane + 64];
        carry = sh.u.k.win[full];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if ((uint32_t)lane < full) out32[wbase + (uint32_t)lane] = mine0;
        if ((uint32_t)lane + 64 < full) out32[wbase + (uint32_t)lane + 64] = mine1;
        for (int j = lane; j < 112; j += 64) sh.u.k.win[j] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) sh.u.k.win[0] = carry;
        wbase += full;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0 && (obit & 31u)) out32[wbase] = sh.u.k.win[0];
    return (obit + 7) >> 3;
}

// status_p (may be null): a nonzero word there cancels the work (the VCF kernels raised it: the block goes to the host).
// tok_all: 65 536 + 64 dwords per block of the grid; out_all: DF_SLOT bytes per member; out_len[m]: its deflate stream's bytes
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PGD_WAVES, PGD_WAVES))) void k_deflate(const uint8_t *__restrict__ text, const long long *__restrict__ total_p, const long long *__restrict__ status_p,
                                                uint32_t *__restrict__ tok_all, uint8_t *__restrict__ out_all, uint32_t *__restrict__ out_len) {
    __shared__ DfShared sh;
    const int lane = (int)threadIdx.x;
    if (status_p && status_p[0] != 0) return;
    const long long total = *total_p;
    const long long n_members = (total + DF_PIECE - 1) / DF_PIECE;
    uint32_t *tok = tok_all + (size_t)blockIdx.x * (65536 + 64);
    for (long long m = blockIdx.x; m < n_members; m += gridDim.x) {
        const long long left = total - m * DF_PIECE;
        const uint32_t n = (uint32_t)(left < DF_PIECE ? left : DF_PIECE);
        const uint32_t got = df_member(text + m * DF_PIECE, n, out_all + (size_t)m * DF_SLOT, tok, sh, lane);
        if (lane == 0) out_len[m] = got;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// the members one behind the other at dst: gzip header with the BC field, the deflate stream, CRC-32, size; *comp_total = their bytes
__global__ __launch_bounds__(64) void k_bgzf_assemble(const long long *__restrict__ total_p, const long long *__restrict__ status_p,
                                                      const uint8_t *__restrict__ out_all, const uint32_t *__restrict__ out_len,
                                                      const uint32_t *__restrict__ crc, uint8_t *__restrict__ dst, long long *__restrict__ comp_total) {
    const int lane = (int)threadIdx.x;
    if (status_p && status_p[0] != 0) {
        if (blockIdx.x == 0 && lane == 0) *comp_total = 0;
        return;
    }
    const long long total = *total_p;
    const long long n_members = (total + DF_PIECE - 1) / DF_PIECE;
    if (n_members == 0 && blockIdx.x == 0 && lane == 0) *comp_total = 0;
    for (long long m = blockIdx.x; m < n_members; m += gridDim.x) {
        long long at = 0;
        for (long long k = lane; k < m; k += 64) at += (long long)out_len[k] + 26;
        for (int d = 32; d >= 1; d >>= 1) at += __shfl_xor(at, d, 64);
        const uint32_t n = out_len[m], size = n + 26;
        const long long left = total - m * DF_PIECE;
        const uint32_t isize = (uint32_t)(left < DF_PIECE ? left : DF_PIECE);
        uint8_t *o = dst + at;
        if (lane < 18) {
            const uint8_t hd[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)((size - 1) & 255), (uint8_t)((size - 1) >> 8)};
            o[lane] = hd[lane];
        }
        const uint8_t *src = out_all + (size_t)m * DF_SLOT;
        for (uint32_t k = (uint32_t)lane; k < n; k += 64) o[18 + k] = src[k];
        if (lane < 4) o[18 + n + lane] = (uint8_t)(crc[m] >> (8 * lane));
        else if (lane < 8) o[18 + n + lane] = (uint8_t)(isize >> (8 * (lane - 4)));
        if (m == n_members - 1 && lane == 0) *comp_total = at + size;
    }
}

}  // namespace

// bytes the members of `text_bytes` of text may take at most, and how many they are
static inline int64_t df_members(int64_t text_bytes) { return (text_bytes + DF_PIECE - 1) / DF_PIECE; }

// Queues the three kernels on `st`: the text at text_d (its length read on the device from *total_d, at most max_text bytes; readable
// 32 bytes past its end) -> BGZF members at D.comp, their bytes in *comp_total_d.  status_d: see k_deflate.
int pg_deflate_queue(pg_ctx *c, hipStream_t st, pg_ctx::Deflate &D, const uint8_t *text_d, const long long *total_d, int64_t max_text,
                     const long long *status_d, long long *comp_total_d) {
    const int64_t max_members = std::max<int64_t>(1, df_members(max_text));
    int rc;
    hipDeviceProp_t prop;
    static int cus = 0;
    if (!cus) {
        HIPCHK(hipGetDeviceProperties(&prop, c->device));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int64_t waves = std::min<int64_t>(max_members, (int64_t)cus * 4 * PGD_WAVES);       // 14.6 KB of LDS and 168 registers a wave: ten per compute unit
    if ((rc = D.tok.ensure_roomy((size_t)waves * (65536 + 64))) != PG_OK) return rc;
    if ((rc = D.slots.ensure_roomy((size_t)max_members * DF_SLOT + 64)) != PG_OK) return rc;
    if ((rc = D.out_len.ensure_roomy((size_t)max_members)) != PG_OK) return rc;
    if ((rc = D.crc.ensure_roomy((size_t)max_members)) != PG_OK) return rc;
    if ((rc = D.comp.ensure_roomy((size_t)max_members * (DF_SLOT) + 64)) != PG_OK) return rc;
    hipLaunchKernelGGL(k_deflate, dim3((unsigned)waves), dim3(64), 0, st, text_d, total_d, status_d, D.tok.p, D.slots.p, D.out_len.p);
    HIPCHK(hipGetLastError());
    if ((rc = pg_launch_crc32_pieces(c, st, text_d, total_d, DF_PIECE, max_members, D.crc.p)) != PG_OK) return rc;
    hipLaunchKernelGGL(k_bgzf_assemble, dim3((unsigned)std::min<int64_t>(max_members, 4096)), dim3(64), 0, st, total_d, status_d, D.slots.p,
                       D.out_len.p, D.crc.p, D.comp.p, comp_total_d);
    HIPCHK(hipGetLastError());
    return PG_OK;
}

// text[0 .. len) (host) -> BGZF members at out (host; no end-of-file member), compressed on the device.  For tests and tools: the
// text crosses PCIe twice; the drop-ins compress what already lies in HBM (pg_vcf_dev_rows_bgzf).  kernel_ms_out: the three kernels.
extern "C" int pg_bgzf_compress_device(pg_ctx *c, const uint8_t *text, int64_t len, uint8_t *out, int64_t out_cap, int64_t *out_len_out,
                                       double *kernel_ms_out) {
    if (!c || len < 0 || (len && !text) || !out_len_out) return pg_fail(PG_ERR_ARG, "pg_bgzf_compress_device: bad argument");
    *out_len_out = 0;
    if (len == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream_up;
    pg_ctx::Deflate &D = c->deflate;
    int rc;
    if ((rc = D.text.ensure_roomy((size_t)len + 64)) != PG_OK) return rc;
    if ((rc = D.totals.ensure(4)) != PG_OK || (rc = D.h_totals.ensure(4)) != PG_OK) return rc;
    D.h_totals.p[0] = len;
    D.h_totals.p[1] = 0;
    HIPCHK(hipMemcpyAsync(D.totals.p, D.h_totals.p, 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(D.text.p, text, (size_t)len, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(D.text.p + len, 0, 64, st));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return pg_fail(PG_ERR_HIP, "hipEventCreate");
    }
    rc = [&]() -> int {                                       // (every early return of HIPCHK passes the events' release below)
        HIPCHK(hipEventRecord(e0, st));
        int r = pg_deflate_queue(c, st, D, D.text.p, reinterpret_cast<const long long *>(D.totals.p), len, nullptr,
                                 reinterpret_cast<long long *>(D.totals.p + 1));
        if (r != PG_OK) return r;
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipMemcpyAsync(D.h_totals.p + 1, D.totals.p + 1, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (kernel_ms_out) *kernel_ms_out = ms;
        const int64_t got = D.h_totals.p[1];
        *out_len_out = got;
        if (out && got <= out_cap) {
            HIPCHK(hipMemcpy(out, D.comp.p, (size_t)got, hipMemcpyDeviceToHost));
        } else if (out)
            return pg_fail(PG_ERR_ARG, "pg_bgzf_compress_device: the members take %lld bytes, the output holds %lld", (long long)got, (long long)out_cap);
        return PG_OK;
    }();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
