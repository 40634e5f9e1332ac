// K0 on the device: `.geno` text -> resident rows ("GPU-side text tokenizer").
// Replaces, like the host tokenizer (pg_encode.cpp), GenoFileReader.nextSite / parseGenoLine (genomics.py:1940-1945, 1884-1904) +
// splitSeq / forceHomo / seqArrayToNumArray (genomics.py:390-396, 407-408, 74-77) -- for the regular case: every data line is
//     scaffold <ws> position <ws> cell <sep> cell <sep> ... cell '\n'
// with ONE separator character between cells and cells of one width (what parseVCF.py writes).  Anything else -- comment or
// blank lines inside the block, runs of blanks between cells, '\r', cells of mixed width (mixed ploidy) -- makes the kernels
// raise a status bit, and the caller takes the host tokenizer for that block: the fast path never guesses.
//
//   k_nl_count / k_nl_scan / k_nl_write   positions of the line feeds (tile counts, one-block scan, compacted write)
//   k_tok_parse                           one wave per line: lane 0 reads scaffold + position, the lanes take the cells
//                                         c, c+64, ... -> one-hot codes written into the row at the slots of the layout
#include "pg_ctx.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include <unistd.h>

namespace {

constexpr int NL_SUB = 4096;             // bytes per pass of a 256-thread block over its tile (16 bytes per thread)
constexpr int NL_PASSES = 4;
constexpr int NL_TILE = NL_SUB * NL_PASSES;

__device__ __forceinline__ int count_nl16(const uint8_t *text, int64_t at, int64_t len, uint32_t *mask_out) {
    uint32_t mask = 0u;                                   // bit k: text[at + k] == '\n'
    if (at + 16 <= len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + at);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (((w[q] >> (8 * b)) & 0xFFu) == 10u) mask |= 1u << (4 * q + b);
    } else {
        for (int k = 0; k < 16 && at + k < len; ++k)
            if (text[at + k] == 10) mask |= 1u << k;
    }
    *mask_out = mask;
    return __popc(mask);
}

__global__ __launch_bounds__(256) void k_nl_count(const uint8_t *__restrict__ text, int64_t len, int32_t *__restrict__ tile_count) {
    __shared__ int sh[256];
    int mine = 0;
#pragma unroll
    for (int ps = 0; ps < NL_PASSES; ++ps) {
        const int64_t at = (int64_t)blockIdx.x * NL_TILE + ps * NL_SUB + threadIdx.x * 16;
        uint32_t m;
        if (at < len) mine += count_nl16(text, at, len, &m);
    }
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_count[blockIdx.x] = sh[0];
}

// exclusive prefix of the tile counts (one block; a 1 GiB block of text has 65 536 tiles): every thread adds up a contiguous share
// of the tiles, the 256 shares are scanned in LDS, every thread writes the prefixes of its share (the first version scanned 256
// tiles per trip with sixteen barriers each: 0.30 ms per block of text, a tenth of what the tokenizer's kernels took)
__global__ __launch_bounds__(256) void k_nl_scan(const int32_t *__restrict__ tile_count, int64_t n_tiles, int64_t *__restrict__ tile_base,
                                                 int64_t *__restrict__ total) {
    __shared__ long long sh[256];
    const int64_t per = (n_tiles + 255) / 256;
    const int64_t t0 = (int64_t)threadIdx.x * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    long long mine = 0;
    for (int64_t t = t0; t < t1; ++t) mine += tile_count[t];
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const long long x = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += x;
        __syncthreads();
    }
    long long run = sh[threadIdx.x] - mine;
    for (int64_t t = t0; t < t1; ++t) {
        tile_base[t] = run;
        run += tile_count[t];
    }
    if (threadIdx.x == 255) *total = sh[255];
}

__global__ __launch_bounds__(256) void k_nl_write(const uint8_t *__restrict__ text, int64_t len, const int64_t *__restrict__ tile_base,
                                                  int64_t *__restrict__ nl_pos) {
    __shared__ int sh[256];
    int64_t base = tile_base[blockIdx.x];
    for (int ps = 0; ps < NL_PASSES; ++ps) {
        const int64_t at = (int64_t)blockIdx.x * NL_TILE + ps * NL_SUB + threadIdx.x * 16;
        uint32_t mask = 0u;
        const int mine = at < len ? count_nl16(text, at, len, &mask) : 0;
        sh[threadIdx.x] = mine;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int x = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += x;
            __syncthreads();
        }
        int64_t k = base + sh[threadIdx.x] - mine;
        while (mask) {
            const int b = __builtin_ctz(mask);
            mask &= mask - 1u;
            nl_pos[k++] = at + b;
        }
        base += sh[255];
        __syncthreads();
    }
}

__device__ __forceinline__ bool blank(uint8_t ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f'; }
__device__ __forceinline__ int8_t base_code(uint8_t ch) { return ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 4 : ch == 'T' ? 8 : 0; }

// status bits (host: any bit set -> this block goes through the host tokenizer instead)
enum { TOK_IRREGULAR = 1, TOK_BAD_POS = 2, TOK_COMMENT = 4 };

// dip[ch]: the two one-hot codes of IUPAC diploid character ch, low nibble | high nibble << 4 (genomics.py:14-15)
struct DipTable { uint8_t v[256]; };

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane)); }

// One wave per line.  The first 64 bytes of the line (scaffold, position, start of the cells) are read one byte per lane and
// taken apart with ballots: the ends of the two tokens are bit scans, the position is a short scalar loop over readlane.  A line
// whose prefix does not fit the 64 bytes (or starts with blanks) goes through the byte-by-byte walk on lane 0.
__global__ __launch_bounds__(256) void k_tok_parse(const uint8_t *__restrict__ text, const int64_t *__restrict__ nl_pos, int64_t n_lines,
                                                   int fmt, int n_cols, int cells_w, int max_ploidy,
                                                   const int32_t *__restrict__ col_slot, const int32_t *__restrict__ col_ploidy,
                                                   const int32_t *__restrict__ col_off, const int32_t *__restrict__ col_w,
                                                   int8_t *__restrict__ rows, int S, int64_t *__restrict__ pos_out,
                                                   int64_t *__restrict__ run_row, int64_t *__restrict__ run_off, int32_t *__restrict__ run_len,
                                                   int32_t *__restrict__ n_runs, int64_t run_cap, int32_t *__restrict__ status, DipTable dip) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_lines) return;
    const int64_t ls = row ? nl_pos[row - 1] + 1 : 0, le = nl_pos[row];            // [ls, le): the line without its '\n'
    int64_t cells_at = -1;
    int bad = 0;
    const int nv = (int)(le - ls < 64 ? le - ls : 64);
    const int ch = lane < nv ? (int)text[ls + lane] : 10;
    const uint64_t valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    const uint64_t bl = __ballot(blank((uint8_t)ch)) & valid, nb = ~bl & valid;
    bool fast = false;
    if (nv > 0 && (nb & 1ull) && bl) {
        const int p1 = __builtin_ctzll(bl);                                       // end of the scaffold token
        const uint64_t r1 = nb & ~((1ull << p1) - 1ull);
        if (r1) {
            int d0 = __builtin_ctzll(r1);                                         // start of the position
            const uint64_t b2 = bl & ~((1ull << d0) - 1ull);
            if (b2) {
                const int p2 = __builtin_ctzll(b2);                               // its end
                const uint64_t r2 = nb & ~((1ull << p2) - 1ull);
                // the previous line's scaffold token, for the run flag
                bool differs = row == 0, prev_ok = true;
                if (row > 0) {
                    const int64_t pls = row > 1 ? nl_pos[row - 2] + 1 : 0, ple = nl_pos[row - 1];
                    const int pnv = (int)(ple - pls < 64 ? ple - pls : 64);
                    const int pch = lane < pnv ? (int)text[pls + lane] : 10;
                    const int p0 = rl(pch, 0);
                    prev_ok = pnv > 0 && !blank((uint8_t)p0);
                    const uint64_t neq = __ballot(ch != pch) & ((1ull << p1) - 1ull);
                    const bool ends = p1 >= pnv ? (p1 == pnv && ple - pls == p1) : blank((uint8_t)rl(pch, p1));
                    differs = neq != 0ull || !ends;
                }
                if (r2 && prev_ok) {
                    fast = true;
                    if (rl(ch, 0) == '#') bad |= TOK_COMMENT;
                    const int c0 = rl(ch, d0);
                    const bool neg = c0 == '-';
                    if (c0 == '+' || c0 == '-') ++d0;
                    const uint64_t dg = __ballot(ch >= '0' && ch <= '9');
                    const uint64_t range = ((1ull << p2) - 1ull) & ~((1ull << d0) - 1ull);
                    long long v = 0;
                    if (p2 <= d0 || p2 - d0 > 18 || (dg & range) != range) bad |= TOK_BAD_POS;      // (up to 18 digits: int64, as the host tokenizer)
                    else
                        for (int k = d0; k < p2; ++k) v = v * 10 + (rl(ch, k) - '0');
                    cells_at = ls + __builtin_ctzll(r2);
                    if (le - cells_at != (int64_t)cells_w) bad |= TOK_IRREGULAR;
                    if (lane == 0) {
                        pos_out[row] = neg ? -v : v;
                        if (differs) {
                            const int k = atomicAdd(n_runs, 1);
                            if (k < run_cap) { run_row[k] = row; run_off[k] = ls; run_len[k] = p1; }
                        }
                    }
                }
            }
        }
    }
    if (!fast && lane == 0) {
        int64_t p = ls;
        if (p >= le || text[p] == '#') bad |= TOK_COMMENT;
        while (p < le && blank(text[p])) ++p;
        const int64_t s0 = p;
        while (p < le && !blank(text[p])) ++p;
        if (p == s0) bad |= TOK_COMMENT;                                         // blank line
        // a new scaffold run starts where the token differs from the previous line's
        bool differs = row == 0;
        if (row > 0) {
            int64_t q = row > 1 ? nl_pos[row - 2] + 1 : 0;
            const int64_t qe = nl_pos[row - 1];
            while (q < qe && blank(text[q])) ++q;
            int64_t a = s0;
            while (a < p && q < qe && text[a] == text[q]) { ++a; ++q; }
            differs = !(a == p && (q == qe || blank(text[q])));
        }
        if (differs) {
            const int k = atomicAdd(n_runs, 1);
            if (k < run_cap) { run_row[k] = row; run_off[k] = s0; run_len[k] = (int32_t)(p - s0); }
        }
        while (p < le && blank(text[p])) ++p;
        bool neg = false;
        if (p < le && (text[p] == '+' || text[p] == '-')) { neg = text[p] == '-'; ++p; }
        long long v = 0;
        const int64_t d0 = p;
        while (p < le && text[p] >= '0' && text[p] <= '9' && p - d0 < 18) { v = v * 10 + (text[p] - '0'); ++p; }
        if (p == d0 || (p < le && !blank(text[p]))) bad |= TOK_BAD_POS;      // (a longer or zero-padded number: the host tokenizer's case)
        pos_out[row] = neg ? -v : v;
        while (p < le && blank(text[p])) ++p;
        cells_at = p;
        // the regular layout: n_cols cells of their columns' widths, one separator between them, the last cell ends the line
        if (le - p != (int64_t)cells_w) bad |= TOK_IRREGULAR;
    }
    cells_at = ((int64_t)__builtin_amdgcn_readfirstlane((int)(cells_at >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cells_at);
    bad = __builtin_amdgcn_readfirstlane(bad);
    if (!bad) {
        int8_t *out = rows + row * (int64_t)S;
        for (int c = lane; c < n_cols; c += 64) {
            // (every column has its own width: the haploid samples of a mixed-ploidy file have shorter cells; col_off / col_w are
            // what the block's first line shows, and a line that deviates is irregular)
            const int cellw = col_w[c];
            const uint8_t *cell = text + cells_at + col_off[c];
            if (c + 1 < n_cols && !blank(cell[cellw])) bad |= TOK_IRREGULAR;
            for (int k = 0; k < cellw; ++k)
                if (blank(cell[k])) bad |= TOK_IRREGULAR;                       // a shorter cell
            const int pl = col_ploidy[c];
            if (pl <= 0) continue;
            const int32_t *slots = col_slot + (size_t)c * max_ploidy;
            if (fmt == PG_FMT_DIPLO) {
                const uint8_t d = dip.v[cell[0]];
                out[slots[0]] = (int8_t)(d & 15);
                out[slots[1]] = (int8_t)(d >> 4);
            } else {
                const int step = fmt == PG_FMT_PHASED ? 2 : 1;
                for (int k = 0; k < pl; ++k) out[slots[k]] = base_code(cell[step * k]);
            }
        }
    }
    if (bad) atomicOr(status, bad);
}


// k_tok_parse, second form: two kernels.
//   k_tok_heads   a THREAD per line: the line's first 64 bytes go into LDS (four unaligned 16-byte loads), the thread walks them --
//                 scaffold token, position, start of the cells -- the way the host tokenizer does, compares its scaffold token with
//                 the line before (the neighbour's bytes are in LDS too) and leaves position, start of the cells (-1: a line the fast
//                 path does not take) and, where a run starts, the run.  Vector instructions on 64 lines at once: the first
//                 form spent 860 SCALAR instructions per line on this (ballots, mask arithmetic, a 64-bit multiply per digit,
//                 execution masks: it was bound by the scalar unit, 2.1 - 2.6 ms per GiB of text).
//   k_tok_cells   a wavefront per line, 16 lines one after the other: column tables (slots, ploidies, cell offsets and widths) in
//                 LDS, a cell and its separator as ONE (unaligned) dword load, the row put together in LDS (a byte per slot) and
//                 stored whole with coalesced dwords (zeros where no column lands: the caller need not clear the rows).
// Same outputs, same status bits as k_tok_parse, which stays for layouts whose tables do not fit LDS and as PG_TOK_PARSE=1 (A/B, tests).
constexpr int TOK_LPW = 16;
constexpr int HEAD_BYTES = 64, HEAD_PITCH = 68;

__global__ __launch_bounds__(256) void k_tok_heads(const uint8_t *__restrict__ text, const int64_t *__restrict__ nl_pos, int64_t n_lines, int cells_w,
                                                   int64_t *__restrict__ pos_out, int64_t *__restrict__ cells_at_out,
                                                   int64_t *__restrict__ run_row, int64_t *__restrict__ run_off, int32_t *__restrict__ run_len,
                                                   int32_t *__restrict__ n_runs, int64_t run_cap, int32_t *__restrict__ status) {
    __shared__ uint32_t hb[257][HEAD_PITCH / 4];           // [0]: the line in front of the block's first line
    const int t = (int)threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * 256 + t;
    int64_t ls = 0, le = 0;
    if (row < n_lines) {
        ls = row ? nl_pos[row - 1] + 1 : 0;
        le = nl_pos[row];
        // (the text buffer has 96 bytes of room behind its last byte: a head may be read past the end of a short last line)
        uint4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_memcpy(&q[k], text + ls + 16 * k, 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hb[t + 1][4 * k] = q[k].x;
            hb[t + 1][4 * k + 1] = q[k].y;
            hb[t + 1][4 * k + 2] = q[k].z;
            hb[t + 1][4 * k + 3] = q[k].w;
        }
    }
    int64_t pls = 0, ple = 0;
    if (row > 0 && row < n_lines) {
        pls = row > 1 ? nl_pos[row - 2] + 1 : 0;
        ple = ls - 1;
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                uint32_t w;
                __builtin_memcpy(&w, text + pls + 4 * k, 4);
                hb[0][k] = w;
            }
        }
    }
    __syncthreads();
    if (row >= n_lines) return;
    const uint8_t *mine = reinterpret_cast<const uint8_t *>(hb[t + 1]), *prev = reinterpret_cast<const uint8_t *>(hb[t]);
    const int64_t n = le - ls, pn = ple - pls;
    auto B = [&](int64_t k) -> uint8_t { return k < HEAD_BYTES ? mine[k] : text[ls + k]; };       // k < n
    auto PB = [&](int64_t k) -> uint8_t { return k < HEAD_BYTES ? prev[k] : text[pls + k]; };    // k < pn
    int bad = 0;
    int64_t p = 0;
    if (n <= 0 || B(0) == '#') bad |= TOK_COMMENT;
    while (p < n && blank(B(p))) ++p;
    const int64_t s0 = p;
    while (p < n && !blank(B(p))) ++p;
    if (p == s0) bad |= TOK_COMMENT;                                             // blank line
    // a new scaffold run starts where the token differs from the previous line's
    bool differs = row == 0;
    if (row > 0) {
        int64_t q = 0;
        while (q < pn && blank(PB(q))) ++q;
        int64_t a = s0;
        while (a < p && q < pn && B(a) == PB(q)) { ++a; ++q; }
        differs = !(a == p && (q == pn || blank(PB(q))));
    }
    if (differs) {
        const int k = atomicAdd(n_runs, 1);
        if (k < run_cap) { run_row[k] = row; run_off[k] = ls + s0; run_len[k] = (int32_t)(p - s0); }
    }
    while (p < n && blank(B(p))) ++p;
    bool neg = false;
    if (p < n && (B(p) == '+' || B(p) == '-')) { neg = B(p) == '-'; ++p; }
    long long v = 0;
    const int64_t d0 = p;
    while (p < n && B(p) >= '0' && B(p) <= '9' && p - d0 < 18) { v = v * 10 + (B(p) - '0'); ++p; }
    if (p == d0 || (p < n && !blank(B(p)))) bad |= TOK_BAD_POS;                  // (a longer or zero-padded number: the host tokenizer's case)
    pos_out[row] = neg ? -v : v;
    while (p < n && blank(B(p))) ++p;
    // the regular layout: n_cols cells of their columns' widths, one separator between them, the last cell ends the line
    if (n - p != (int64_t)cells_w) bad |= TOK_IRREGULAR;
    cells_at_out[row] = bad ? -1 : ls + p;
    if (bad) atomicOr(status, bad);
}

__global__ __launch_bounds__(256) void k_tok_cells(const uint8_t *__restrict__ text, const int64_t *__restrict__ cells_at_in, int64_t n_lines,
                                                   int fmt, int n_cols, int max_ploidy, const int32_t *__restrict__ dcols,
                                                   int8_t *__restrict__ rows, int S, int32_t *__restrict__ status, DipTable dip) {
    extern __shared__ int32_t tab[];                      // col_slot [n_cols][max_ploidy] | col_ploidy | col_off | col_w | 4 rows of S bytes
    const int n_tab = n_cols * (max_ploidy + 3);
    for (int k = (int)threadIdx.x; k < n_tab; k += 256) tab[k] = dcols[k];
    __syncthreads();
    const int32_t *col_slot = tab, *col_ploidy = tab + (size_t)n_cols * max_ploidy, *col_off = col_ploidy + n_cols, *col_w = col_off + n_cols;
    const int lane = threadIdx.x & 63;
    int8_t *const lrow = reinterpret_cast<int8_t *>(tab + n_tab) + (size_t)(threadIdx.x >> 6) * S;      // this wavefront's row (S % 16 == 0)
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * TOK_LPW;
    if (row0 >= n_lines) return;
    const int n_here = (int)(n_lines - row0 < TOK_LPW ? n_lines - row0 : TOK_LPW);
    long long cav = -1;                                   // lane i: where the cells of line row0 + i begin (-1: not a regular line)
    if (lane < n_here) cav = cells_at_in[row0 + lane];
    const int ca_lo = (int)(uint32_t)cav, ca_hi = (int)(cav >> 32);
    int bad = 0;
    for (int r = 0; r < n_here; ++r) {
        const int64_t row = row0 + r;
        const int64_t cells_at = ((int64_t)__builtin_amdgcn_readlane(ca_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane(ca_lo, r);
        for (int k = lane; k < S / 4; k += 64) reinterpret_cast<uint32_t *>(lrow)[k] = 0u;
        if (cells_at >= 0) {
            int8_t *out = lrow;
            const uint8_t *cells = text + cells_at;
            for (int c = lane; c < n_cols; c += 64) {
                const int cellw = col_w[c];
                const uint8_t *cell = cells + col_off[c];
                const int pl = col_ploidy[c];
                const int32_t *slots = col_slot + (size_t)c * max_ploidy;
                if (cellw <= 3) {
                    // the cell and its separator in one load (the text buffer has room behind its last byte; any alignment)
                    uint32_t w4;
                    __builtin_memcpy(&w4, cell, 4);
                    const uint8_t b0 = (uint8_t)w4, b1 = (uint8_t)(w4 >> 8), b2 = (uint8_t)(w4 >> 16), b3 = (uint8_t)(w4 >> 24);
                    const uint8_t sep = cellw == 1 ? b1 : cellw == 2 ? b2 : b3;
                    int cb = (int)(c + 1 < n_cols) & (int)!blank(sep);
                    cb |= (int)blank(b0) | ((int)(cellw > 1) & (int)blank(b1)) | ((int)(cellw > 2) & (int)blank(b2));      // a shorter cell
                    if (cb) bad |= TOK_IRREGULAR;
                    if (pl <= 0) continue;
                    if (fmt == PG_FMT_DIPLO) {
                        const uint8_t d = dip.v[b0];
                        out[slots[0]] = (int8_t)(d & 15);
                        out[slots[1]] = (int8_t)(d >> 4);
                    } else if (fmt == PG_FMT_PHASED) {
                        out[slots[0]] = base_code(b0);
                        if (pl > 1) out[slots[1]] = base_code(b2);
                    } else {
                        out[slots[0]] = base_code(b0);
                        if (pl > 1) out[slots[1]] = base_code(b1);
                        if (pl > 2) out[slots[2]] = base_code(b2);
                    }
                } else {
                    if (c + 1 < n_cols && !blank(cell[cellw])) bad |= TOK_IRREGULAR;
                    for (int k = 0; k < cellw; ++k)
                        if (blank(cell[k])) bad |= TOK_IRREGULAR;
                    if (pl <= 0) continue;
                    if (fmt == PG_FMT_DIPLO) {
                        const uint8_t d = dip.v[cell[0]];
                        out[slots[0]] = (int8_t)(d & 15);
                        out[slots[1]] = (int8_t)(d >> 4);
                    } else {
                        const int step = fmt == PG_FMT_PHASED ? 2 : 1;
                        for (int k = 0; k < pl; ++k) out[slots[k]] = base_code(cell[step * k]);
                    }
                }
            }
        }
        // (LDS operations of a wavefront execute in order: the reads below see the bytes scattered above)
        uint32_t *grow = reinterpret_cast<uint32_t *>(rows + row * (int64_t)S);
        for (int k = lane; k < S / 4; k += 64) grow[k] = reinterpret_cast<const uint32_t *>(lrow)[k];
    }
    if (bad) atomicOr(status, bad);
}

// k_tok_cells for the usual layouts (every cell of at most three characters: phased / pairs of ploidy <= 2 / 3, haplo, diplo): no branch
// on a lane's column anywhere -- a lane past the last column takes the last column again (the same bytes to the same places), a
// column that is not wanted or an allele a sample does not have writes into a dump byte behind the row, blanks and base codes are
// bit arithmetic, the diplo table sits in LDS -- so the compiler keeps the line loop's state in scalar registers and the loop is a
// few dozen vector instructions per 64 cells.
__device__ __forceinline__ uint32_t blank32(uint32_t ch) {          // ' ' \t \r \v \f  (9, 11, 12, 13, 32)
    const uint32_t t = ch - 9u;
    return (t < 24u ? (0x80001Du >> t) : 0u) & 1u;
}
__device__ __forceinline__ uint32_t code32(uint32_t ch) {           // A 1, C 2, G 4, T 8, anything else 0
    const uint32_t i = (ch >> 1) & 3u;                               // A 0, C 1, T 2, G 3
    return ch == ((0x47544341u >> (8u * i)) & 255u) ? (0x04080201u >> (8u * i)) & 255u : 0u;
}
// NIT > 0: the layout has 64 * (NIT - 1) + 1 ... 64 * NIT columns -- a lane's NIT column-table entries stay in registers for all of its
// wave's lines and a line's NIT loads leave together (the table read from LDS in front of every load made a line four trips one after
// the other: 0.70 ms per GiB of text at a quarter of what its bytes cost); NIT = 0: any number of columns, the table read per step
template <int FMT, int NIT>
__global__ __launch_bounds__(256) void k_tok_cells3(const uint8_t *__restrict__ text, const int64_t *__restrict__ cells_at_in, int64_t n_lines,
                                                    int n_cols, int max_ploidy, const int32_t *__restrict__ dcols, int8_t *__restrict__ rows, int S,
                                                    int32_t *__restrict__ status, DipTable dip) {
    extern __shared__ int32_t lds[];                      // packed column table [n_cols] x 4 | diplo table 64 words | 4 x (row S + dump 64)
    int4 *ctab = reinterpret_cast<int4 *>(lds);
    uint32_t *dtab = reinterpret_cast<uint32_t *>(lds + 4 * n_cols);
    {
        const int32_t *col_slot = dcols, *col_ploidy = dcols + (size_t)n_cols * max_ploidy, *col_off = col_ploidy + n_cols, *col_w = col_off + n_cols;
        for (int c = (int)threadIdx.x; c < n_cols; c += 256) {
            const int pl = col_ploidy[c];
            const int s0 = pl > 0 ? col_slot[(size_t)c * max_ploidy] : 0, s1 = pl > 1 ? col_slot[(size_t)c * max_ploidy + 1] : 0,
                      s2 = pl > 2 ? col_slot[(size_t)c * max_ploidy + 2] : 0;
            ctab[c] = make_int4(col_off[c], col_w[c] | ((pl < 0 ? 0 : pl) << 8), s0 | (s1 << 16), s2);
        }
        if (threadIdx.x < 64) {
            const int k = 4 * (int)threadIdx.x;
            dtab[threadIdx.x] = dip.v[k] | (dip.v[k + 1] << 8) | (dip.v[k + 2] << 16) | ((uint32_t)dip.v[k + 3] << 24);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int RS = S + 64;                                // a wavefront's row + 64 dump bytes
    uint8_t *const lrow = reinterpret_cast<uint8_t *>(lds + 4 * n_cols + 64) + (size_t)(threadIdx.x >> 6) * RS;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * TOK_LPW;
    if (row0 >= n_lines) return;
    const int n_here = (int)(n_lines - row0 < TOK_LPW ? n_lines - row0 : TOK_LPW);
    long long cav = -1;
    if (lane < n_here) cav = cells_at_in[row0 + lane];
    const int ca_lo = (int)(uint32_t)cav, ca_hi = (int)(cav >> 32);
    const int n_it = (n_cols + 63) >> 6, n_dw = S >> 2, n_dwit = (n_dw + 63) >> 6;
    const int dump = S + lane;
    uint32_t badv = 0u;
    int4 ev[NIT > 0 ? NIT : 1];
    if (NIT > 0) {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int c0 = lane + 64 * j;
            ev[j] = ctab[c0 < n_cols ? c0 : n_cols - 1];
        }
    }
    // (NIT > 0) the NEXT line's bytes are asked for before this line's are worked on
    uint32_t wn[NIT > 0 ? NIT : 1];
    auto ask = [&](int r) {
        const int64_t at = ((int64_t)__builtin_amdgcn_readlane(ca_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane(ca_lo, r);
        if (at >= 0) {
#pragma unroll
            for (int j = 0; j < (NIT > 0 ? NIT : 1); ++j) __builtin_memcpy(&wn[j], text + at + ev[j].x, 4);
        }
    };
    if (NIT > 0) ask(0);
    for (int r = 0; r < n_here; ++r) {
        const int64_t row = row0 + r;
        const int64_t cells_at = ((int64_t)__builtin_amdgcn_readlane(ca_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane(ca_lo, r);
        uint32_t wv[NIT > 0 ? NIT : 1];
        if (NIT > 0) {
#pragma unroll
            for (int j = 0; j < NIT; ++j) wv[j] = wn[j];
            if (r + 1 < n_here) ask(r + 1);
        }
        for (int j = 0; j < n_dwit; ++j) {
            const int k = lane + 64 * j;
            reinterpret_cast<uint32_t *>(lrow)[k < n_dw ? k : n_dw - 1] = 0u;
        }
        if (cells_at >= 0) {                              // (uniform)
            const uint8_t *cells = text + cells_at;
            // one step: the cell of column c (table entry e, the four bytes w4 at its place) into the row
            auto step = [&](int c, const int4 e, uint32_t w4) {
                const uint32_t cellw = (uint32_t)e.y & 255u, pl = (uint32_t)e.y >> 8;
                const uint32_t b0 = w4 & 255u, b1 = (w4 >> 8) & 255u, b2 = (w4 >> 16) & 255u, b3 = w4 >> 24;
                const uint32_t sep = cellw == 1u ? b1 : cellw == 2u ? b2 : b3;
                badv |= ((uint32_t)(c + 1 < n_cols) & (blank32(sep) ^ 1u)) | blank32(b0) | ((uint32_t)(cellw > 1u) & blank32(b1)) |
                        ((uint32_t)(cellw > 2u) & blank32(b2));
                uint32_t x0, x1, x2 = 0u;
                if (FMT == PG_FMT_DIPLO) {
                    const uint32_t d = (dtab[b0 >> 2] >> (8u * (b0 & 3u))) & 255u;
                    x0 = d & 15u;
                    x1 = d >> 4;
                } else if (FMT == PG_FMT_PHASED) {
                    x0 = code32(b0);
                    x1 = code32(b2);
                } else {
                    x0 = code32(b0);
                    x1 = code32(b1);
                    x2 = code32(b2);
                }
                const int s0 = e.z & 0xFFFF, s1 = (int)((uint32_t)e.z >> 16);
                lrow[pl > 0u ? s0 : dump] = (uint8_t)x0;
                lrow[pl > 1u ? s1 : dump] = (uint8_t)x1;
                if (FMT == PG_FMT_PAIRS) lrow[pl > 2u ? e.w : dump] = (uint8_t)x2;
            };
            if (NIT > 0) {                                // (launched with NIT == n_it)
#pragma unroll
                for (int j = 0; j < NIT; ++j) {
                    const int c0 = lane + 64 * j;
                    step(c0 < n_cols ? c0 : n_cols - 1, ev[j], wv[j]);
                }
            } else {
                for (int j = 0; j < n_it; ++j) {
                    const int c0 = lane + 64 * j, c = c0 < n_cols ? c0 : n_cols - 1;
                    const int4 e = ctab[c];
                    uint32_t w4;
                    __builtin_memcpy(&w4, cells + e.x, 4);
                    step(c, e, w4);
                }
            }
        }
        // (LDS operations of a wavefront execute in order: the reads below see the bytes scattered above)
        uint32_t *grow = reinterpret_cast<uint32_t *>(rows + row * (int64_t)S);
        for (int j = 0; j < n_dwit; ++j) {
            const int k0 = lane + 64 * j, k = k0 < n_dw ? k0 : n_dw - 1;
            grow[k] = reinterpret_cast<const uint32_t *>(lrow)[k];
        }
    }
    if (__ballot(badv != 0u) && lane == 0) atomicOr(status, TOK_IRREGULAR);
}

}  // namespace

// ---- host side: three steps per block, two blocks in flight ------------------------------------------------------------------
//   submit   the block's text goes to the device (self-paced staging threads, their own copy streams) into text slot 0 or 1; the
//            line feeds are counted behind the copies (k_nl_count / k_nl_scan, the total on its way back)
//   parse    (needs the number of lines: waits for that one number) positions of the line feeds, rows cleared, k_tok_parse --
//            all queued on the context's copy stream, nothing waited for
//   collect  waits for the parse, hands back positions and scaffold runs
// The ingestion thread of the drivers runs   parse(k) -> submit(k+1) -> collect(k):   the kernels of block k work while the text
// of block k+1 crosses PCIe.  pg_tokenize_text / pg_tokenize_file are the three steps in a row on slot 0.

// where the block's text is: memory (bytes, a gunzipped buffer, a memory-mapped file) or a file descriptor + offset (plain text on
// disk: the staging threads pread() straight from the page cache into their page-locked buffers -- no page faults of a mapping,
// no second pass over the text)
struct TokSource {
    const char *text;
    int fd;
    int64_t off;
    bool read(int64_t at, void *dst, size_t n) const {
        if (text) {
            memcpy(dst, text + at, n);
            return true;
        }
        size_t got = 0;
        while (got < n) {
            const ssize_t r = pread(fd, static_cast<char *>(dst) + got, n - got, (off_t)(off + at + (int64_t)got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
        return true;
    }
};

// ---- bytes to the device: self-paced staging threads ----
// Every thread takes the next 4 MiB chunk, brings it into one of its two page-locked buffers (pread / memcpy) and queues its copy
// to the device on a copy stream (four streams serve the eight threads: a stream costs 4 ms to create and four reach the link's
// rate), while its other buffer is still in flight: no thread waits for another, the queue of copies stays deep enough to keep PCIe
// busy, and the threads are started once per call (the first version started up to 16 threads per 32 MiB piece and copied one
// piece at a time: 41 GB/s of text on a 57 GB/s link; this one 55).  Returns when every byte has landed.
static int stage_bytes(pg_ctx *c, const TokSource &src, int64_t len, uint8_t *dst) {
    if (len <= 0) return PG_OK;
    int rc;
    constexpr size_t CH = 4u << 20;
    int nt = pg_host_threads();
    nt = nt < 1 ? 1 : (nt > PG_TOK_WORKERS ? PG_TOK_WORKERS : nt);
    const int64_t n_chunks = (len + (int64_t)CH - 1) / (int64_t)CH;
    const int use = (int)std::min<int64_t>(nt, n_chunks);
    const int n_streams = std::min(use, PG_TOK_STREAMS);
    if ((size_t)use * 2 * CH > c->tok_pin.cap) {
        if ((rc = c->tok_pin.ensure((size_t)PG_TOK_WORKERS * 2 * CH)) != PG_OK) return rc;
    }
    for (int t = 0; t < use; ++t) {
        if (t < n_streams && !c->tok_st[t]) HIPCHK(hipStreamCreateWithFlags(&c->tok_st[t], hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b)
            if (!c->tok_wev[t][b]) HIPCHK(hipEventCreateWithFlags(&c->tok_wev[t][b], hipEventDisableTiming));
    }
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};                              // 1 = read error, 2 = HIP error
    auto work = [&](int t) {
        if (hipSetDevice(c->device) != hipSuccess) { failed.store(2); return; }
        hipStream_t ws = c->tok_st[t % n_streams];
        bool used[2] = {false, false};
        int b = 0;
        for (;;) {
            const int64_t ci = next.fetch_add(1);
            if (ci >= n_chunks || failed.load()) break;
            const int64_t a = ci * (int64_t)CH;
            const size_t n = (size_t)std::min<int64_t>((int64_t)CH, len - a);
            uint8_t *pin = c->tok_pin.p + ((size_t)t * 2 + b) * CH;
            if (used[b] && hipEventSynchronize(c->tok_wev[t][b]) != hipSuccess) { failed.store(2); break; }
            if (!src.read(a, pin, n)) { failed.store(1); break; }
            if (hipMemcpyAsync(dst + a, pin, n, hipMemcpyHostToDevice, ws) != hipSuccess ||
                hipEventRecord(c->tok_wev[t][b], ws) != hipSuccess) { failed.store(2); break; }
            used[b] = true;
            b ^= 1;
        }
        for (int k = 0; k < 2; ++k)                          // (the stream is shared: wait for the own copies, not for the stream)
            if (used[k] && hipEventSynchronize(c->tok_wev[t][k]) != hipSuccess) failed.store(2);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < use; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (failed.load() == 1) return pg_fail(PG_ERR_ARG, "cannot read %lld bytes at offset %lld of the input", (long long)len, (long long)src.off);
    if (failed.load()) return pg_fail(PG_ERR_HIP, "a staging copy failed: %s", hipGetErrorString(hipGetLastError()));
    return PG_OK;
}

// the checks of a submit that do not need the text: 1 = go on, 0 = the fast path does not take this layout (*ok_out stays 0), < 0: error
static int tok_check(pg_ctx *c, int slot, int fmt, int n_cols, int max_ploidy, const int32_t *col_slot, const int32_t *col_ploidy, int *ok_out) {
    if (!c || !col_slot || !col_ploidy || !ok_out) return -pg_fail(PG_ERR_ARG, "pg_tokenize_submit: null argument");
    if (slot < 0 || slot > 1) return -pg_fail(PG_ERR_ARG, "pg_tokenize_submit: slot %d", slot);
    if (c->n_hap <= 0) return -pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (fmt < PG_FMT_PHASED || fmt > PG_FMT_DIPLO || n_cols < 1 || max_ploidy < 1) return -pg_fail(PG_ERR_ARG, "pg_tokenize_text: bad format description");
    *ok_out = 0;
    bool any = false;
    for (int k = 0; k < n_cols; ++k) {
        if (col_ploidy[k] <= 0) continue;
        any = true;
        if (col_ploidy[k] > max_ploidy) return -pg_fail(PG_ERR_ARG, "col_ploidy[%d]=%d exceeds max_ploidy", k, col_ploidy[k]);
        if (fmt == PG_FMT_DIPLO && col_ploidy[k] != 2) return 0;
        if (fmt == PG_FMT_HAPLO && col_ploidy[k] != 1) return 0;
        for (int a = 0; a < col_ploidy[k]; ++a) {
            const int s = col_slot[(size_t)k * max_ploidy + a];
            if (s < 0 || s >= c->n_hap) return -pg_fail(PG_ERR_ARG, "col_slot[%d][%d]=%d out of range", k, a, s);
        }
    }
    return any ? 1 : 0;
}

// The cell widths of the block, read off its first line [text, e) (scaffold, position, then n_cols cells with one blank between
// them): a wanted column's cell must be as wide as its ploidy says (phased: 2 p - 1 characters, pairs: p, haplo / diplo: 1); a file
// of mixed ploidy has narrower cells for its haploid samples.  Every other line is held against these widths on the device.
static bool tok_layout(pg_ctx::TokSlot &T, const char *text, const char *e, int fmt, int n_cols, int max_ploidy, const int32_t *col_slot,
                       const int32_t *col_ploidy) {
    T.cols.assign((size_t)n_cols * (max_ploidy + 3), 0);         // col_slot | col_ploidy | cell offsets | cell widths
    memcpy(T.cols.data(), col_slot, (size_t)n_cols * max_ploidy * 4);
    memcpy(T.cols.data() + (size_t)n_cols * max_ploidy, col_ploidy, (size_t)n_cols * 4);
    int32_t *col_off = T.cols.data() + (size_t)n_cols * (max_ploidy + 1), *col_w = col_off + n_cols;
    auto blank_h = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f'; };
    const char *p = text;
    for (int tok = 0; tok < 2; ++tok) {                                    // scaffold, position
        while (p < e && blank_h(*p)) ++p;
        if (p == e) return false;
        while (p < e && !blank_h(*p)) ++p;
    }
    while (p < e && blank_h(*p)) ++p;
    const char *cells0 = p;
    for (int k = 0; k < n_cols; ++k) {
        const char *b = p;
        while (p < e && !blank_h(*p)) ++p;
        const int w = (int)(p - b);
        if (w < 1) return false;
        if (col_ploidy[k] > 0) {
            const int want = fmt == PG_FMT_PHASED ? 2 * col_ploidy[k] - 1 : (fmt == PG_FMT_PAIRS ? col_ploidy[k] : 1);
            if (w != want) return false;
        }
        col_off[k] = (int32_t)(b - cells0);
        col_w[k] = w;
        if (k + 1 < n_cols) {
            if (p == e || !blank_h(*p)) return false;
            ++p;                                                            // exactly one separator
        }
    }
    if (p != e) return false;
    T.cells_w = (int)(p - cells0);
    T.fmt = fmt;
    T.n_cols = n_cols;
    T.max_ploidy = max_ploidy;
    return true;
}

// line feeds of the slot's text (T.tp, len bytes): counted on the copy stream, the total on its way to the host
static int tok_count(pg_ctx *c, pg_ctx::TokSlot &T, int64_t len) {
    int rc;
    hipStream_t st = c->stream_up;
    const int64_t n_tiles = (len + NL_TILE - 1) / NL_TILE;
    T.n_tiles = n_tiles;
    if ((rc = T.i32.ensure_roomy((size_t)n_tiles + 4)) != PG_OK) return rc;
    if ((rc = T.i64.ensure_roomy((size_t)n_tiles + 2)) != PG_OK) return rc;
    if ((rc = T.h_total.ensure(8)) != PG_OK) return rc;           // [0] lines, page-locked landing of small results: [1] status | runs, [2] inflate status
    int32_t *d_status = T.i32.p + n_tiles;                          // [0] status bits, [1] number of runs
    int64_t *d_total = T.i64.p + n_tiles;
    HIPCHK(hipMemsetAsync(d_status, 0, 8, st));
    hipLaunchKernelGGL(k_nl_count, dim3((unsigned)n_tiles), dim3(256), 0, st, T.tp, len, T.i32.p);
    hipLaunchKernelGGL(k_nl_scan, dim3(1), dim3(256), 0, st, T.i32.p, n_tiles, T.i64.p, d_total);
    HIPCHK(hipMemcpyAsync(T.h_total.p, d_total, 8, hipMemcpyDeviceToHost, st));
    if (!T.counted) HIPCHK(hipEventCreateWithFlags(&T.counted, hipEventDisableTiming));
    HIPCHK(hipEventRecord(T.counted, st));
    T.state = 2;                                                    // text on the device, lines being counted
    return PG_OK;
}

static int tok_submit(pg_ctx *c, int slot, const TokSource &src, int64_t len, int fmt, int n_cols, int max_ploidy,
                      const int32_t *col_slot, const int32_t *col_ploidy, int *ok_out) {
    const int chk = tok_check(c, slot, fmt, n_cols, max_ploidy, col_slot, col_ploidy, ok_out);
    if (chk < 0) return -chk;
    pg_ctx::TokSlot &T = c->tok[slot];
    T.state = 0;
    T.len = len;
    T.n_lines = 0;
    T.deflated = false;
    if (len == 0) { T.state = 1; *ok_out = 1; return PG_OK; }
    if (chk == 0) return PG_OK;
    char last = 0;
    if (!src.read(len - 1, &last, 1)) return pg_fail(PG_ERR_ARG, "pg_tokenize_file: cannot read %lld bytes at offset %lld", (long long)len, (long long)src.off);
    if (last != '\n') return PG_OK;
    // the block's first line (in memory: where it is; from a file: read ahead, 64 KiB and more until its line feed shows)
    std::vector<char> head;
    const char *text = src.text;
    if (!text) {
        size_t n = (size_t)std::min<int64_t>(len, 1 << 16);
        for (;;) {
            head.resize(n);
            if (!src.read(0, head.data(), n)) return pg_fail(PG_ERR_ARG, "pg_tokenize_file: read failed");
            if (memchr(head.data(), '\n', n) || (int64_t)n == len) break;
            n = (size_t)std::min<int64_t>(len, (int64_t)n * 4);
        }
        text = head.data();
    }
    const char *e = static_cast<const char *>(memchr(text, '\n', head.empty() ? (size_t)len : head.size()));
    if (!tok_layout(T, text, e, fmt, n_cols, max_ploidy, col_slot, col_ploidy)) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t_stage0 = now();
    if ((rc = T.text.ensure_roomy((size_t)len + 96)) != PG_OK) return rc;
    T.tp = T.text.p;
    if ((rc = stage_bytes(c, src, len, T.text.p)) != PG_OK) return rc;
    c->tok_stage_s += std::chrono::duration<double>(now() - t_stage0).count();
    c->tok_bytes += len;
    if ((rc = tok_count(c, T, len)) != PG_OK) return rc;
    *ok_out = 1;
    return PG_OK;
}

int pg_inflate_queue(pg_ctx *c, hipStream_t st, pg_ctx::Inflate &I, const uint32_t *comp_d, uint32_t n_dw, const uint32_t *in_off,
                     const uint32_t *in_len, const uint32_t *out_len, const uint32_t *crc, int64_t n_members, uint8_t *text_d,
                     uint32_t nl_cap, uint64_t text_limit, int64_t *d_total, int32_t *d_over, hipStream_t crc_st);
void pg_launch_nl_gather(hipStream_t st, pg_ctx::Inflate &I, uint32_t nl_cap, int64_t n_members, int64_t text_base, int64_t *nl_pos);
int pg_inflate_error(const int32_t *status);
void pg_launch_gather_bytes(hipStream_t st, const uint8_t *text, const int64_t *off, const int32_t *len, const int64_t *dst, int n,
                            uint8_t *out);

// the device half of a deflated block's submit: the members cross PCIe as they are, k_inflate writes their text behind `head` into
// the slot's text buffer and lists its line feeds (first_line_len sizes the members' lists); shared by the `.geno` tokenizer and
// the VCF kernels (pg_tok_bgzf_submit)
static int tok_bgzf_core(pg_ctx *c, int slot, const uint8_t *comp, int fd, int64_t file_offset, int64_t comp_len, const uint32_t *in_off,
                         const uint32_t *in_len, const uint32_t *out_len, const uint32_t *crc, int64_t n_members, const char *head,
                         int64_t head_len, int64_t text_len, int64_t total, int64_t first_line_len, int *ok_out) {
    pg_ctx::TokSlot &T = c->tok[slot];
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream_up;
    int rc;
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t_stage0 = now();
    static const bool trace = getenv("PG_TOK_TRACE") != nullptr;
    double tr[6] = {0, 0, 0, 0, 0, 0};
    auto lap = [&](int k) { if (trace) tr[k] = std::chrono::duration<double>(now() - t_stage0).count() * 1e3; };
    if ((rc = T.text.ensure_roomy((size_t)total + 96)) != PG_OK) return rc;
    T.tp = T.text.p;                                                 // (hipMalloc aligns to 256 bytes; the members' text starts at any byte)
    // (the slot's buffers are free: the block that used them last has been collected.  Bytes behind comp_len in the last dword are
    // never consumed by a valid stream, and a damaged one is stopped by the bounds of its member)
    const size_t n_dw = ((size_t)comp_len + 3) / 4;
    if ((rc = T.inf.comp.ensure_roomy(n_dw + 1)) != PG_OK) return rc;
    if ((rc = T.h_total.ensure(8)) != PG_OK) return rc;
    lap(0);
    bool pinned = false;
    if (comp) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, comp) == hipSuccess) pinned = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();                                // (pageable memory: not an error)
    }
    if (pinned) {
        // page-locked bytes (the reader thread filled a buffer of the engine's pool): ONE asynchronous DMA on a copy stream of its own,
        // beside the kernels of the previous block; the caller keeps the buffer alive until it has collected this block
        if (!c->tok_st[0]) HIPCHK(hipStreamCreateWithFlags(&c->tok_st[0], hipStreamNonBlocking));
        if (!T.staged) HIPCHK(hipEventCreateWithFlags(&T.staged, hipEventDisableTiming));
        HIPCHK(hipMemcpyAsync(T.inf.comp.p, comp, (size_t)comp_len, hipMemcpyHostToDevice, c->tok_st[0]));
        HIPCHK(hipEventRecord(T.staged, c->tok_st[0]));
        HIPCHK(hipStreamWaitEvent(st, T.staged, 0));
    } else {
        // (from a file: the staging threads pread() the members from the page cache straight into their page-locked buffers)
        const TokSource src{comp ? reinterpret_cast<const char *>(comp) : nullptr, comp ? -1 : fd, comp ? 0 : file_offset};
        if ((rc = stage_bytes(c, src, comp_len, reinterpret_cast<uint8_t *>(T.inf.comp.p))) != PG_OK) return rc;
    }
    lap(1);
    c->tok_stage_s += std::chrono::duration<double>(now() - t_stage0).count();
    c->tok_bytes += comp_len;
    if (head_len) {
        if ((rc = T.h_head.ensure((size_t)head_len)) != PG_OK) return rc;
        memcpy(T.h_head.p, head, (size_t)head_len);
        HIPCHK(hipMemcpyAsync(T.tp, T.h_head.p, (size_t)head_len, hipMemcpyHostToDevice, st));
    }
    lap(2);
    // The block's line feeds: listed by k_inflate member by member as the text passes through its registers (no pass over the text:
    // k_nl_count, k_nl_scan and k_nl_write took 0.7 ms per GiB, a line of the chain's 5.5), unless the carried head holds one or
    // PG_BGZF_NL=0 asks for the passes.  A member's list holds four times the lines a member of such lines has; a member with more
    // (the block's lines are much shorter than its first) sends the block through the passes after all (pg_tokenize_parse).
    static const bool nl_in_inflate = !(getenv("PG_BGZF_NL") && atoi(getenv("PG_BGZF_NL")) == 0);
    uint32_t nl_cap = 0;
    if (nl_in_inflate && n_members > 0 && !(head_len && memchr(head, '\n', (size_t)head_len))) {
        const int64_t per_member = 65536 / std::max<int64_t>(first_line_len + 1, 8) + 1;
        nl_cap = (uint32_t)std::min<int64_t>(16384, std::max<int64_t>(64, 4 * per_member));
        if (const char *cap = getenv("PG_BGZF_NL_CAP")) nl_cap = (uint32_t)std::max(1, atoi(cap));         // (tests: lists that are too short)
    }
    T.inf.nl_cap = nl_cap;
    T.n_members = n_members;
    T.head_len = head_len;
    if (nl_cap) {
        T.n_tiles = 0;
        if ((rc = T.i32.ensure_roomy(8)) != PG_OK) return rc;
        if ((rc = T.i64.ensure_roomy(4)) != PG_OK) return rc;
        HIPCHK(hipMemsetAsync(T.i32.p, 0, 16, st));                 // [0] status bits, [1] number of runs, [2] a member's list was too short
    }
    // PG_BGZF_CRC_STREAM=1: k_crc32 on a stream of its own, beside the tokenizer's kernels.  Measured on the whole north star
    // (profiles/r06/t2_whole_crc_stream_ab.txt): 0.81 - 0.84 s against 0.64 - 0.80 s with the check on the chain -- kernels of two streams
    // share the compute units and every one of them, the statistics kernels of the main thread included, gets slower; not the default.
    static const bool crc_aside = getenv("PG_BGZF_CRC_STREAM") && atoi(getenv("PG_BGZF_CRC_STREAM")) == 1;
    if (crc_aside && !c->tok_crc) HIPCHK(hipStreamCreateWithFlags(&c->tok_crc, hipStreamNonBlocking));
    // (the slot's text must not be written while the check of the block that used it last still reads it)
    if (T.inf.crc_pending && T.inf.ev_crc) HIPCHK(hipStreamWaitEvent(st, T.inf.ev_crc, 0));
    if ((rc = pg_inflate_queue(c, st, T.inf, T.inf.comp.p, (uint32_t)n_dw, in_off, in_len, out_len, crc, n_members, T.tp + head_len,
                               nl_cap, (uint64_t)(text_len - head_len), nl_cap ? T.i64.p : nullptr, nl_cap ? T.i32.p + 2 : nullptr,
                               crc_aside ? c->tok_crc : nullptr)) != PG_OK) return rc;
    lap(3);
    HIPCHK(hipMemcpyAsync(T.h_total.p + 2, T.inf.status.p, 8, hipMemcpyDeviceToHost, st));     // [error bits, first bad member]: read by parse
    if (nl_cap) {
        HIPCHK(hipMemcpyAsync(T.h_total.p, T.i64.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(T.h_total.p + 3, T.i32.p + 2, 4, hipMemcpyDeviceToHost, st));
        if (!T.counted) HIPCHK(hipEventCreateWithFlags(&T.counted, hipEventDisableTiming));
        HIPCHK(hipEventRecord(T.counted, st));
        T.state = 2;
    } else if ((rc = tok_count(c, T, text_len)) != PG_OK) return rc;
    lap(4);
    if (trace) fprintf(stderr, "PG_TOK_TRACE submit_bgzf slot %d pinned %d: alloc %.2f copy %.2f head %.2f inflate_queue %.2f count_queue %.2f ms\n", slot, (int)pinned, tr[0], tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[4] - tr[3]);
    *ok_out = 1;
    return PG_OK;
}


// The submit step for a block of bgzip-compressed text: comp[0 .. comp_len) -- or, with comp == NULL, comp_len bytes at file_offset of
// fd -- holds n_members whole BGZF members (table: pg_bgzf_walk),
// which cross PCIe as they are and are inflated on the device (k_inflate, a wavefront per member; CRC-32 checked) into the slot's
// text buffer behind `head` (head_len bytes of text the caller already has: the unfinished line the previous block ended with).
// The block's text is  head + the members' text  cut to text_len bytes (the caller keeps what follows the last line feed for the
// next block); first_line: the block's first line without its line feed (the cell widths are read off it).
static int tok_submit_bgzf(pg_ctx *c, int slot, const uint8_t *comp, int fd, int64_t file_offset, int64_t comp_len, const uint32_t *in_off, const uint32_t *in_len,
                           const uint32_t *out_len, const uint32_t *crc, int64_t n_members, const char *head, int64_t head_len,
                           int64_t text_len, const char *first_line, int64_t first_line_len, int fmt, int n_cols, int max_ploidy,
                           const int32_t *col_slot, const int32_t *col_ploidy, int *ok_out) {
    const int chk = tok_check(c, slot, fmt, n_cols, max_ploidy, col_slot, col_ploidy, ok_out);
    if (chk < 0) return -chk;
    if (comp_len < 0 || comp_len >= (1ll << 32) || n_members < 0 || head_len < 0 || text_len < 0 || first_line_len < 0 ||
        (n_members > 0 && ((!comp && fd < 0) || file_offset < 0 || !in_off || !in_len || !out_len)) || (head_len > 0 && !head) ||
        (first_line_len > 0 && !first_line))
        return pg_fail(PG_ERR_ARG, "pg_tokenize_submit_bgzf: bad argument");
    int64_t total = head_len;
    for (int64_t k = 0; k < n_members; ++k) {
        if ((int64_t)in_off[k] + in_len[k] > comp_len) return pg_fail(PG_ERR_ARG, "pg_tokenize_submit_bgzf: member %lld lies outside the compressed bytes", (long long)k);
        total += out_len[k];
    }
    if (text_len > total) return pg_fail(PG_ERR_ARG, "pg_tokenize_submit_bgzf: text_len %lld exceeds the %lld bytes of the block", (long long)text_len, (long long)total);
    pg_ctx::TokSlot &T = c->tok[slot];
    T.state = 0;
    T.len = text_len;
    T.n_lines = 0;
    T.deflated = true;
    if (text_len == 0) { T.state = 1; *ok_out = 1; return PG_OK; }
    if (chk == 0) return PG_OK;
    if (!tok_layout(T, first_line, first_line + first_line_len, fmt, n_cols, max_ploidy, col_slot, col_ploidy)) return PG_OK;
    return tok_bgzf_core(c, slot, comp, fd, file_offset, comp_len, in_off, in_len, out_len, crc, n_members, head, head_len, text_len, total,
                         first_line_len, ok_out);
}

static int tok_parse(pg_ctx *c, int slot, int64_t row_offset, int64_t row_capacity, int64_t run_capacity, int64_t *n_rows_out, int *ok_out) {
    if (!c || !n_rows_out || !ok_out) return pg_fail(PG_ERR_ARG, "pg_tokenize_parse: null argument");
    if (slot < 0 || slot > 1) return pg_fail(PG_ERR_ARG, "pg_tokenize_parse: slot %d", slot);
    pg_ctx::TokSlot &T = c->tok[slot];
    *n_rows_out = 0;
    *ok_out = 0;
    if (T.state == 1) { T.state = 4; T.n_lines = 0; *ok_out = 1; return PG_OK; }          // an empty block
    if (T.state != 2) return pg_fail(PG_ERR_STATE, "pg_tokenize_parse: nothing submitted to slot %d", slot);
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(c->device));
    // PG_TOK_PARSE_STREAM=1: the block's parse kernels on a stream of their own, beside the NEXT block's k_inflate on the copy stream
    // (submit(k + 1) queues it right behind this call) -- the block's text and line count are complete (the wait below), its rows are
    // read only after collect
    static const bool parse_aside = getenv("PG_TOK_PARSE_STREAM") && atoi(getenv("PG_TOK_PARSE_STREAM")) == 1;
    if (parse_aside && !c->tok_parse) HIPCHK(hipStreamCreateWithFlags(&c->tok_parse, hipStreamNonBlocking));
    hipStream_t st = parse_aside ? c->tok_parse : c->stream_up;
    HIPCHK(hipEventSynchronize(T.counted));
    if (T.deflated) {
        int32_t ist[2];
        memcpy(ist, T.h_total.p + 2, 8);
        if (ist[0]) { T.state = 0; return pg_inflate_error(ist); }
    }
    bool nl_lists = T.deflated && T.inf.nl_cap > 0;
    if (nl_lists) {
        int32_t over = 0;
        memcpy(&over, T.h_total.p + 3, 4);
        if (over) {
            // a member held more line feeds than its list: the passes over the text after all
            nl_lists = false;
            T.inf.nl_cap = 0;
            int rc0 = tok_count(c, T, T.len);
            if (rc0 != PG_OK) return rc0;
            HIPCHK(hipEventSynchronize(T.counted));
            ++c->tok_nl_fallbacks;
        }
    }
    const int64_t n_lines = T.h_total.p[0];
    T.n_lines = n_lines;
    *n_rows_out = n_lines;
    T.state = 0;
    if (n_lines == 0) { T.state = 4; *ok_out = 1; return PG_OK; }
    if (n_lines > row_capacity || row_offset < 0 || row_offset + n_lines > c->cap_sites) return PG_OK;   // (caller sizes the rows from a bound)
    if (run_capacity < 1) return pg_fail(PG_ERR_ARG, "pg_tokenize_parse: run_capacity < 1");
    int rc;
    const int n_cols = T.n_cols, max_ploidy = T.max_ploidy;
    const int64_t n_tiles = T.n_tiles;
    int32_t *d_status = T.i32.p + n_tiles;
    if ((rc = T.nl.ensure_roomy((size_t)n_lines)) != PG_OK) return rc;
    if (nl_lists) pg_launch_nl_gather(st, T.inf, T.inf.nl_cap, T.n_members, T.head_len, T.nl.p);
    else hipLaunchKernelGGL(k_nl_write, dim3((unsigned)n_tiles), dim3(256), 0, st, T.tp, T.len, T.i64.p, T.nl.p);
    if ((rc = T.dcols.ensure(T.cols.size())) != PG_OK) return rc;
    if ((rc = T.h_cols.ensure(T.cols.size())) != PG_OK) return rc;
    memcpy(T.h_cols.p, T.cols.data(), T.cols.size() * 4);
    HIPCHK(hipMemcpyAsync(T.dcols.p, T.h_cols.p, T.cols.size() * 4, hipMemcpyHostToDevice, st));
    const int64_t run_cap = std::min<int64_t>(run_capacity, n_lines);
    T.run_cap = run_cap;
    if ((rc = T.pos.ensure_roomy((size_t)run_cap + 8)) != PG_OK) return rc;                          // run_len [run_cap] (int32)
    if ((rc = T.pos64.ensure_roomy((size_t)n_lines + 8)) != PG_OK) return rc;                       // positions (int64)
    if ((rc = T.off.ensure((size_t)run_cap * 2)) != PG_OK) return rc;                         // run_row, run_off
    if ((rc = T.h_pos.ensure_roomy((size_t)n_lines)) != PG_OK) return rc;
    // the form with several lines per wavefront, the column tables and the row being put together in LDS (k_tok_parse2) whenever they
    // fit; PG_TOK_PARSE=1: the first form (one line per wavefront, tables in global memory, rows cleared first), which is also the
    // route of wider layouts
    const size_t tab_bytes = (size_t)n_cols * (max_ploidy + 3) * 4, lds_bytes = tab_bytes + 4 * (size_t)c->S;
    static const bool first_form = getenv("PG_TOK_PARSE") && atoi(getenv("PG_TOK_PARSE")) == 1;
    const bool second_form = lds_bytes <= 60 * 1024 && !first_form;
    if (!second_form) HIPCHK(hipMemsetAsync(c->gt.p + row_offset * c->S, 0, (size_t)n_lines * c->S, st));
    DipTable dip;
    memset(dip.v, 0, sizeof(dip.v));
    {
        const char *d = "ACGKMNSRTWY";
        const char *pr[] = {"AA", "CC", "GG", "GT", "AC", "NN", "CG", "AG", "TT", "AT", "CT"};
        auto code = [](char ch) { return ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 4 : ch == 'T' ? 8 : 0; };
        for (int k = 0; d[k]; ++k) dip.v[(int)d[k]] = (uint8_t)(code(pr[k][0]) | (code(pr[k][1]) << 4));
    }
    bool pos_on_small = false;
    if (second_form) {
        if ((rc = T.cells_at.ensure_roomy((size_t)n_lines + 8)) != PG_OK) return rc;
        hipLaunchKernelGGL(k_tok_heads, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, st, T.tp, T.nl.p, n_lines, T.cells_w,
                           T.pos64.p, T.cells_at.p, T.off.p, T.off.p + run_cap, T.pos.p, d_status + 1, run_cap, d_status);
        // the positions (8 bytes a line: 10 MB per GiB of text, 0.45 ms of a blit kernel) leave on the small stream as soon as k_tok_heads
        // has them, beside the cell kernel -- on the copy stream they stood between this block's cells and the next block's k_inflate
        static const bool pos_aside = !(getenv("PG_TOK_POS_STREAM") && atoi(getenv("PG_TOK_POS_STREAM")) == 0);
        if (pos_aside) {
            if (!c->tok_small) HIPCHK(hipStreamCreateWithFlags(&c->tok_small, hipStreamNonBlocking));
            if (!T.heads_done) HIPCHK(hipEventCreateWithFlags(&T.heads_done, hipEventDisableTiming));
            if (!T.pos_copied) HIPCHK(hipEventCreateWithFlags(&T.pos_copied, hipEventDisableTiming));
            HIPCHK(hipEventRecord(T.heads_done, st));
            HIPCHK(hipStreamWaitEvent(c->tok_small, T.heads_done, 0));
            HIPCHK(hipMemcpyAsync(T.h_pos.p, T.pos64.p, (size_t)n_lines * 8, hipMemcpyDeviceToHost, c->tok_small));
            HIPCHK(hipEventRecord(T.pos_copied, c->tok_small));
            pos_on_small = true;
        }
        const int64_t per_block = 4 * TOK_LPW;
        const dim3 grid((unsigned)((n_lines + per_block - 1) / per_block));
        int widest = 0;
        for (int k = 0; k < n_cols; ++k) widest = std::max(widest, (int)T.cols[(size_t)n_cols * (max_ploidy + 2) + k]);
        const size_t lds3 = (size_t)n_cols * 16 + 256 + 4 * ((size_t)c->S + 64);
        static const bool general_cells = getenv("PG_TOK_PARSE") && atoi(getenv("PG_TOK_PARSE")) == 2;      // (A/B, tests)
        if (widest <= 3 && lds3 <= 60 * 1024 && !general_cells) {
#define PG_CELLS3N(F, N) hipLaunchKernelGGL((k_tok_cells3<F, N>), grid, dim3(256), lds3, st, T.tp, T.cells_at.p, n_lines, n_cols, max_ploidy, T.dcols.p, \
                                            c->gt.p + row_offset * c->S, c->S, d_status, dip)
            // (a lane's column-table entries in registers for layouts of up to 256 columns; PG_TOK_CELLS_REGS=0: the table read per step)
            static const bool in_regs = !(getenv("PG_TOK_CELLS_REGS") && atoi(getenv("PG_TOK_CELLS_REGS")) == 0);
            const int nit = (n_cols + 63) / 64;
#define PG_CELLS3(F)                                         \
    do {                                                     \
        if (!in_regs || nit > 4) PG_CELLS3N(F, 0);           \
        else if (nit == 1) PG_CELLS3N(F, 1);                 \
        else if (nit == 2) PG_CELLS3N(F, 2);                 \
        else if (nit == 3) PG_CELLS3N(F, 3);                 \
        else PG_CELLS3N(F, 4);                               \
    } while (0)
            if (T.fmt == PG_FMT_DIPLO) PG_CELLS3(PG_FMT_DIPLO);
            else if (T.fmt == PG_FMT_PHASED) PG_CELLS3(PG_FMT_PHASED);
            else PG_CELLS3(PG_FMT_PAIRS);                 // pairs and haplo: an allele per character
#undef PG_CELLS3
#undef PG_CELLS3N
        } else {
            hipLaunchKernelGGL(k_tok_cells, grid, dim3(256), lds_bytes, st, T.tp, T.cells_at.p,
                               n_lines, T.fmt, n_cols, max_ploidy, T.dcols.p, c->gt.p + row_offset * c->S, c->S, d_status, dip);
        }
    } else {
        hipLaunchKernelGGL(k_tok_parse, dim3((unsigned)((n_lines + 3) / 4)), dim3(256), 0, st, T.tp, T.nl.p, n_lines, T.fmt, n_cols,
                           T.cells_w, max_ploidy, T.dcols.p, T.dcols.p + (size_t)n_cols * max_ploidy,
                           T.dcols.p + (size_t)n_cols * (max_ploidy + 1), T.dcols.p + (size_t)n_cols * (max_ploidy + 2),
                           c->gt.p + row_offset * c->S, c->S,
                           T.pos64.p, T.off.p, T.off.p + run_cap, T.pos.p, d_status + 1, run_cap, d_status, dip);
    }
    HIPCHK(hipGetLastError());
    if (T.deflated && T.inf.crc_pending) {
        HIPCHK(hipStreamWaitEvent(st, T.inf.ev_crc, 0));             // (long done: it started behind the inflate)
        HIPCHK(hipMemcpyAsync(T.h_total.p + 4, T.inf.status.p + 2, 8, hipMemcpyDeviceToHost, st));   // read by collect
    }
    HIPCHK(hipMemcpyAsync(T.h_total.p + 1, d_status, 8, hipMemcpyDeviceToHost, st));            // status | runs, as two int32
    if (!pos_on_small) HIPCHK(hipMemcpyAsync(T.h_pos.p, T.pos64.p, (size_t)n_lines * 8, hipMemcpyDeviceToHost, st));
    T.pos_pending = pos_on_small;
    if (!T.parsed) HIPCHK(hipEventCreateWithFlags(&T.parsed, hipEventDisableTiming));
    HIPCHK(hipEventRecord(T.parsed, st));                            // (collect waits for THIS, not for what the next block has queued behind it)
    T.state = 3;
    *ok_out = 1;
    c->tok_kernel_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PG_OK;
}

static int tok_collect(pg_ctx *c, int slot, int64_t *pos_out, int64_t pos_capacity, int64_t *run_row_out, int64_t *run_off_out,
                       int32_t *run_len_out, int64_t run_capacity, int64_t *n_rows_out, int64_t *n_runs_out, int *ok_out) {
    if (!c || !n_rows_out || !n_runs_out || !ok_out) return pg_fail(PG_ERR_ARG, "pg_tokenize_collect: null argument");
    if (slot < 0 || slot > 1) return pg_fail(PG_ERR_ARG, "pg_tokenize_collect: slot %d", slot);
    pg_ctx::TokSlot &T = c->tok[slot];
    *n_rows_out = T.n_lines;
    *n_runs_out = 0;
    *ok_out = 0;
    if (T.state == 4) { T.state = 0; *ok_out = 1; return PG_OK; }                            // an empty block
    if (T.state != 3) return PG_OK;                                                          // refused earlier: nothing to collect
    T.state = 0;
    const auto t0 = std::chrono::steady_clock::now();
    struct Clock {
        pg_ctx *c;
        std::chrono::steady_clock::time_point t0;
        ~Clock() { c->tok_kernel_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    } clock{c, t0};
    HIPCHK(hipSetDevice(c->device));
    // the block's own work only: the next block's copy, inflate and line count are already queued behind it on the same stream and
    // go on while the host turns this block into windows
    if (T.parsed) HIPCHK(hipEventSynchronize(T.parsed));
    else HIPCHK(hipStreamSynchronize(c->stream_up));
    if (T.pos_pending) {
        HIPCHK(hipEventSynchronize(T.pos_copied));
        T.pos_pending = false;
    }
    if (!c->tok_small) HIPCHK(hipStreamCreateWithFlags(&c->tok_small, hipStreamNonBlocking));
    hipStream_t st = c->tok_small;
    const int64_t n_lines = T.n_lines;
    if (!pos_out || !run_row_out || !run_off_out || !run_len_out || pos_capacity < n_lines)
        return pg_fail(PG_ERR_ARG, "pg_tokenize_collect: outputs too small for %lld rows", (long long)n_lines);
    if (T.deflated && T.inf.crc_pending) {
        T.inf.crc_pending = false;
        int32_t cst[2];
        memcpy(cst, T.h_total.p + 4, 8);
        if (cst[0]) return pg_inflate_error(cst);                    // a member's text is not what its trailer says
    }
    int32_t status[2];
    memcpy(status, T.h_total.p + 1, 8);
    if (status[0] != 0) return PG_OK;
    memcpy(pos_out, T.h_pos.p, (size_t)n_lines * 8);
    const int64_t nr = status[1], run_cap = T.run_cap;
    *n_runs_out = nr;
    if (nr > run_cap || nr > run_capacity) return PG_OK;                                     // more runs than the caller has room for
    std::vector<int64_t> rr((size_t)nr), ro((size_t)nr);
    std::vector<int32_t> rlen((size_t)nr);
    if (nr) {
        HIPCHK(hipMemcpyAsync(rr.data(), T.off.p, (size_t)nr * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(ro.data(), T.off.p + run_cap, (size_t)nr * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(rlen.data(), T.pos.p, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    std::vector<int64_t> order((size_t)nr);
    for (int64_t k = 0; k < nr; ++k) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return rr[(size_t)a] < rr[(size_t)b]; });   // the appends arrive in any order
    for (int64_t k = 0; k < nr; ++k) {
        const size_t j = (size_t)order[(size_t)k];
        run_row_out[k] = rr[j];
        run_off_out[k] = ro[j];
        run_len_out[k] = rlen[j];
    }
    *ok_out = 1;
    return PG_OK;
}

// ---- the same first steps for the VCF kernels (pg_vcf_dev.hip): a block's text into a slot, its line feeds listed -----------------
// pg_tok_text_submit / pg_tok_bgzf_submit: the text (or the deflated members) to the device, line feeds counted behind it;
// pg_tok_lines: waits for the count, leaves the line-feed positions in the slot's `nl` (queued on the copy stream) and returns their
// number.  The slot's buffers are the tokenizer's: a process runs one of the two.
int pg_tok_text_submit(pg_ctx *c, int slot, const char *text, int fd, int64_t file_offset, int64_t len) {
    pg_ctx::TokSlot &T = c->tok[slot];
    T.state = 0;
    T.len = len;
    T.n_lines = 0;
    T.deflated = false;
    if (len == 0) { T.state = 1; return PG_OK; }
    HIPCHK(hipSetDevice(c->device));
    int rc;
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = T.text.ensure_roomy((size_t)len + 96)) != PG_OK) return rc;
    T.tp = T.text.p;
    const TokSource src{text, text ? -1 : fd, text ? 0 : file_offset};
    if ((rc = stage_bytes(c, src, len, T.text.p)) != PG_OK) return rc;
    c->tok_stage_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->tok_bytes += len;
    return tok_count(c, T, len);
}

int pg_tok_bgzf_submit(pg_ctx *c, int slot, const uint8_t *comp, int64_t comp_len, const uint32_t *in_off, const uint32_t *in_len,
                       const uint32_t *out_len, const uint32_t *crc, int64_t n_members, const char *head, int64_t head_len,
                       int64_t text_len, int64_t line_len_hint) {
    if (comp_len < 0 || comp_len >= (1ll << 32) || n_members < 0 || head_len < 0 || text_len < 0 ||
        (n_members > 0 && (!comp || !in_off || !in_len || !out_len)) || (head_len > 0 && !head))
        return pg_fail(PG_ERR_ARG, "pg_vcf_dev_submit_bgzf: bad argument");
    int64_t total = head_len;
    for (int64_t k = 0; k < n_members; ++k) {
        if ((int64_t)in_off[k] + in_len[k] > comp_len) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_submit_bgzf: member %lld lies outside the compressed bytes", (long long)k);
        total += out_len[k];
    }
    if (text_len > total) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_submit_bgzf: text_len %lld exceeds the %lld bytes of the block", (long long)text_len, (long long)total);
    pg_ctx::TokSlot &T = c->tok[slot];
    T.state = 0;
    T.len = text_len;
    T.n_lines = 0;
    T.deflated = true;
    if (text_len == 0) { T.state = 1; return PG_OK; }
    int ok = 0;
    return tok_bgzf_core(c, slot, comp, -1, 0, comp_len, in_off, in_len, out_len, crc, n_members, head, head_len, text_len, total,
                         line_len_hint, &ok);
}

int pg_tok_lines(pg_ctx *c, int slot, int64_t *n_lines_out) {
    pg_ctx::TokSlot &T = c->tok[slot];
    *n_lines_out = 0;
    if (T.state == 1) return PG_OK;
    if (T.state != 2) return pg_fail(PG_ERR_STATE, "nothing submitted to slot %d", slot);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream_up;
    HIPCHK(hipEventSynchronize(T.counted));
    if (T.deflated) {
        int32_t ist[2];
        memcpy(ist, T.h_total.p + 2, 8);
        if (ist[0]) { T.state = 0; return pg_inflate_error(ist); }
    }
    bool nl_lists = T.deflated && T.inf.nl_cap > 0;
    if (nl_lists) {
        int32_t over = 0;
        memcpy(&over, T.h_total.p + 3, 4);
        if (over) {                                                  // a member held more line feeds than its list: the passes over the text
            nl_lists = false;
            T.inf.nl_cap = 0;
            int rc0 = tok_count(c, T, T.len);
            if (rc0 != PG_OK) return rc0;
            HIPCHK(hipEventSynchronize(T.counted));
            ++c->tok_nl_fallbacks;
        }
    }
    const int64_t n_lines = T.h_total.p[0];
    T.n_lines = n_lines;
    *n_lines_out = n_lines;
    T.state = 0;
    if (n_lines == 0) return PG_OK;
    int rc;
    if ((rc = T.nl.ensure_roomy((size_t)n_lines)) != PG_OK) return rc;
    if (nl_lists) pg_launch_nl_gather(st, T.inf, T.inf.nl_cap, T.n_members, T.head_len, T.nl.p);
    else hipLaunchKernelGGL(k_nl_write, dim3((unsigned)T.n_tiles), dim3(256), 0, st, T.tp, T.len, T.i64.p, T.nl.p);
    HIPCHK(hipGetLastError());
    if (T.deflated && T.inf.crc_pending) {                           // (PG_BGZF_CRC_FOLD=0 / PG_BGZF_CRC_STREAM=1: the check by a kernel of its own)
        HIPCHK(hipStreamWaitEvent(st, T.inf.ev_crc, 0));
        HIPCHK(hipMemcpyAsync(T.h_total.p + 4, T.inf.status.p + 2, 8, hipMemcpyDeviceToHost, st));
    }
    return PG_OK;
}

// the members' checksums of a deflated block whose kernels have finished (0: all as their trailers say)
int pg_tok_crc_result(pg_ctx *c, int slot) {
    pg_ctx::TokSlot &T = c->tok[slot];
    if (T.deflated && T.inf.crc_pending) {
        T.inf.crc_pending = false;
        int32_t cst[2];
        memcpy(cst, T.h_total.p + 4, 8);
        if (cst[0]) return pg_inflate_error(cst);
    }
    return PG_OK;
}

static int tokenize_block(pg_ctx *c, const TokSource &src, int64_t len, int fmt, int n_cols, int max_ploidy,
                          const int32_t *col_slot, const int32_t *col_ploidy, int64_t row_offset, int64_t *pos_out,
                          int64_t row_capacity, int64_t *run_row_out, int64_t *run_off_out, int32_t *run_len_out,
                          int64_t run_capacity, int64_t *n_rows_out, int64_t *n_runs_out, int *ok_out) {
    if (!n_rows_out || !n_runs_out || !ok_out) return pg_fail(PG_ERR_ARG, "pg_tokenize_text: null argument");
    *n_rows_out = 0;
    *n_runs_out = 0;
    *ok_out = 0;
    int ok = 0;
    int rc = tok_submit(c, 0, src, len, fmt, n_cols, max_ploidy, col_slot, col_ploidy, &ok);
    if (rc != PG_OK || !ok) return rc;
    rc = tok_parse(c, 0, row_offset, row_capacity, run_capacity < 1 ? 1 : run_capacity, n_rows_out, &ok);
    if (rc != PG_OK || !ok) return rc;
    if (*n_rows_out > 0 && (!pos_out || !run_row_out || !run_off_out || !run_len_out || run_capacity < 1))
        return pg_fail(PG_ERR_ARG, "pg_tokenize_text: null output");
    return tok_collect(c, 0, pos_out, row_capacity, run_row_out, run_off_out, run_len_out, run_capacity, n_rows_out, n_runs_out, ok_out);
}

extern "C" int pg_tokenize_text(pg_ctx *c, const char *text, int64_t len, int fmt, int n_cols, int max_ploidy,
                                const int32_t *col_slot, const int32_t *col_ploidy, int64_t row_offset, int64_t *pos_out,
                                int64_t row_capacity, int64_t *run_row_out, int64_t *run_off_out, int32_t *run_len_out,
                                int64_t run_capacity, int64_t *n_rows_out, int64_t *n_runs_out, int *ok_out) {
    if (!text && len) return pg_fail(PG_ERR_ARG, "pg_tokenize_text: null argument");
    const TokSource src{text ? text : "", -1, 0};
    return tokenize_block(c, src, len, fmt, n_cols, max_ploidy, col_slot, col_ploidy, row_offset, pos_out, row_capacity, run_row_out,
                          run_off_out, run_len_out, run_capacity, n_rows_out, n_runs_out, ok_out);
}

extern "C" int pg_tokenize_file(pg_ctx *c, int fd, int64_t file_offset, int64_t len, int fmt, int n_cols, int max_ploidy,
                                const int32_t *col_slot, const int32_t *col_ploidy, int64_t row_offset, int64_t *pos_out,
                                int64_t row_capacity, int64_t *run_row_out, int64_t *run_off_out, int32_t *run_len_out,
                                int64_t run_capacity, int64_t *n_rows_out, int64_t *n_runs_out, int *ok_out) {
    if (fd < 0 || file_offset < 0 || len < 0) return pg_fail(PG_ERR_ARG, "pg_tokenize_file: bad file range");
    const TokSource src{nullptr, fd, file_offset};
    return tokenize_block(c, src, len, fmt, n_cols, max_ploidy, col_slot, col_ploidy, row_offset, pos_out, row_capacity, run_row_out,
                          run_off_out, run_len_out, run_capacity, n_rows_out, n_runs_out, ok_out);
}

extern "C" int pg_tokenize_submit(pg_ctx *c, int slot, const char *text, int fd, int64_t file_offset, int64_t len, int fmt, int n_cols,
                                  int max_ploidy, const int32_t *col_slot, const int32_t *col_ploidy, int *ok_out) {
    if ((!text && fd < 0 && len) || file_offset < 0 || len < 0) return pg_fail(PG_ERR_ARG, "pg_tokenize_submit: no text");
    const TokSource src{text ? text : (fd < 0 ? "" : nullptr), fd, file_offset};
    return tok_submit(c, slot, src, len, fmt, n_cols, max_ploidy, col_slot, col_ploidy, ok_out);
}

extern "C" int pg_tokenize_submit_bgzf(pg_ctx *c, int slot, const uint8_t *comp, int fd, int64_t file_offset, int64_t comp_len, const uint32_t *in_off,
                                       const uint32_t *in_len, const uint32_t *out_len, const uint32_t *crc, int64_t n_members,
                                       const char *head, int64_t head_len, int64_t text_len, const char *first_line,
                                       int64_t first_line_len, int fmt, int n_cols, int max_ploidy, const int32_t *col_slot,
                                       const int32_t *col_ploidy, int *ok_out) {
    return tok_submit_bgzf(c, slot, comp, fd, file_offset, comp_len, in_off, in_len, out_len, crc, n_members, head, head_len, text_len, first_line,
                           first_line_len, fmt, n_cols, max_ploidy, col_slot, col_ploidy, ok_out);
}

// The scaffold names of the runs pg_tokenize_collect reported for `slot` (their offsets and lengths in the block's text), read back
// from the text on the device: out receives the names one after the other.  For blocks whose text the host never had (BGZF).
extern "C" int pg_tokenize_run_names(pg_ctx *c, int slot, const int64_t *run_off, const int32_t *run_len, int64_t n_runs, char *out,
                                     int64_t out_capacity) {
    if (!c || slot < 0 || slot > 1 || n_runs < 0 || (n_runs > 0 && (!run_off || !run_len || !out)))
        return pg_fail(PG_ERR_ARG, "pg_tokenize_run_names: bad argument");
    if (n_runs == 0) return PG_OK;
    pg_ctx::TokSlot &T = c->tok[slot];
    if (!T.tp) return pg_fail(PG_ERR_STATE, "pg_tokenize_run_names: no text in slot %d", slot);
    std::vector<int64_t> idx((size_t)n_runs * 2);                   // source offsets | destination offsets
    int64_t total = 0;
    for (int64_t k = 0; k < n_runs; ++k) {
        if (run_off[k] < 0 || run_len[k] < 0 || run_off[k] + run_len[k] > T.len) return pg_fail(PG_ERR_ARG, "pg_tokenize_run_names: run %lld lies outside the block", (long long)k);
        idx[(size_t)k] = run_off[k];
        idx[(size_t)(n_runs + k)] = total;
        total += run_len[k];
    }
    if (total > out_capacity) return pg_fail(PG_ERR_ARG, "pg_tokenize_run_names: %lld bytes of names, room for %lld", (long long)total, (long long)out_capacity);
    if (total == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    // (the block has been collected: its text is complete; the small stream, so that this does not wait for the next block's inflate)
    if (!c->tok_small) HIPCHK(hipStreamCreateWithFlags(&c->tok_small, hipStreamNonBlocking));
    hipStream_t st = c->tok_small;
    int rc;
    const size_t len_words = ((size_t)n_runs * 4 + 7) / 8;           // the int32 lengths ride behind the offsets
    if ((rc = T.names_idx.ensure((size_t)n_runs * 2 + len_words)) != PG_OK) return rc;
    if ((rc = T.names.ensure((size_t)total)) != PG_OK) return rc;
    HIPCHK(hipMemcpyAsync(T.names_idx.p, idx.data(), (size_t)n_runs * 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(T.names_idx.p + n_runs * 2, run_len, (size_t)n_runs * 4, hipMemcpyHostToDevice, st));
    pg_launch_gather_bytes(st, T.tp, T.names_idx.p, reinterpret_cast<const int32_t *>(T.names_idx.p + n_runs * 2), T.names_idx.p + n_runs,
                           (int)n_runs, T.names.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, T.names.p, (size_t)total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PG_OK;
}

extern "C" int pg_tokenize_parse(pg_ctx *c, int slot, int64_t row_offset, int64_t row_capacity, int64_t run_capacity,
                                 int64_t *n_rows_out, int *ok_out) {
    return tok_parse(c, slot, row_offset, row_capacity, run_capacity, n_rows_out, ok_out);
}

extern "C" int pg_tokenize_collect(pg_ctx *c, int slot, int64_t *pos_out, int64_t pos_capacity, int64_t *run_row_out,
                                   int64_t *run_off_out, int32_t *run_len_out, int64_t run_capacity, int64_t *n_rows_out,
                                   int64_t *n_runs_out, int *ok_out) {
    return tok_collect(c, slot, pos_out, pos_capacity, run_row_out, run_off_out, run_len_out, run_capacity, n_rows_out, n_runs_out, ok_out);
}

extern "C" int pg_tokenize_stats(pg_ctx *c, double *stage_seconds_out, double *kernel_seconds_out, int64_t *bytes_out) {
    if (!c || !stage_seconds_out || !kernel_seconds_out || !bytes_out) return pg_fail(PG_ERR_ARG, "pg_tokenize_stats: null argument");
    *stage_seconds_out = c->tok_stage_s;
    *kernel_seconds_out = c->tok_kernel_s;
    *bytes_out = c->tok_bytes;
    return PG_OK;
}


// ---- packed cells (`.pgeno`, codec none) straight from the file: the same staging threads, then k_unpack --------------------------
// pg_stage_file: `len` bytes at file_offset of fd -> byte dst_offset of the slot's device staging buffer (capacity bytes are made
// sure of when dst_offset == 0: stage a block's segments in increasing order).  pg_unpack_staged: n_rows rows of n_cols packed
// cells at byte src_offset of that buffer -> resident rows row_offset .. (k_unpack, queued on the copy stream).  pg_stage_sync:
// wait for everything queued on the copy stream.
extern "C" int pg_stage_file(pg_ctx *c, int slot, int fd, int64_t file_offset, int64_t len, int64_t dst_offset, int64_t capacity) {
    if (!c || slot < 0 || slot > 1 || fd < 0 || file_offset < 0 || len < 0 || dst_offset < 0 || dst_offset + len > capacity)
        return pg_fail(PG_ERR_ARG, "pg_stage_file: bad argument");
    HIPCHK(hipSetDevice(c->device));
    pg_ctx::TokSlot &T = c->tok[slot];
    T.state = 0;
    int rc;
    if ((size_t)capacity + 32 > T.text.cap) {
        if (dst_offset != 0) return pg_fail(PG_ERR_STATE, "pg_stage_file: the staging buffer can only grow at dst_offset 0");
        HIPCHK(hipStreamSynchronize(c->stream_up));              // (an earlier expansion may still read the old buffer)
        if ((rc = T.text.ensure_roomy((size_t)capacity + 32)) != PG_OK) return rc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const TokSource src{nullptr, fd, file_offset};
    rc = stage_bytes(c, src, len, T.text.p + dst_offset);
    c->tok_stage_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->tok_bytes += len;
    return rc;
}

extern "C" int pg_unpack_staged(pg_ctx *c, int slot, int64_t src_offset, int64_t n_rows, int n_cols, const int32_t *slot_src,
                                int64_t row_offset) {
    if (!c || slot < 0 || slot > 1 || !slot_src || n_cols < 1 || n_rows < 0 || src_offset < 0) return pg_fail(PG_ERR_ARG, "pg_unpack_staged: bad argument");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (row_offset < 0 || row_offset + n_rows > c->cap_sites) return pg_fail(PG_ERR_ARG, "rows [%lld,%lld) exceed reserved %lld", (long long)row_offset, (long long)(row_offset + n_rows), (long long)c->cap_sites);
    pg_ctx::TokSlot &T = c->tok[slot];
    if ((size_t)(src_offset + n_rows * n_cols) > T.text.cap) return pg_fail(PG_ERR_ARG, "pg_unpack_staged: cells beyond the staged bytes");
    for (int h = 0; h < c->n_hap; ++h)
        if (slot_src[h] < -1 || slot_src[h] >= 2 * n_cols) return pg_fail(PG_ERR_ARG, "slot_src[%d] = %d out of range", h, slot_src[h]);
    if (n_rows == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = T.h_cols.ensure((size_t)c->n_hap)) != PG_OK) return rc;
    if ((rc = T.dcols.ensure((size_t)c->n_hap)) != PG_OK) return rc;
    HIPCHK(hipStreamSynchronize(c->stream_up));                  // (the table of an earlier call may still be on its way)
    memcpy(T.h_cols.p, slot_src, (size_t)c->n_hap * 4);
    HIPCHK(hipMemcpyAsync(T.dcols.p, T.h_cols.p, (size_t)c->n_hap * 4, hipMemcpyHostToDevice, c->stream_up));
    pg_launch_unpack(c->stream_up, T.text.p + src_offset, n_cols, n_rows, T.dcols.p, c->n_hap, c->gt.p + row_offset * c->S, c->S);
    HIPCHK(hipGetLastError());
    return PG_OK;
}

extern "C" int pg_stage_sync(pg_ctx *c) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream_up));
    return PG_OK;
}
