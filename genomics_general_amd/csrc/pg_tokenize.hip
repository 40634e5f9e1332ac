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
#include <cstring>
#include <thread>
#include <vector>

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

// exclusive prefix of the tile counts (one block; a 1 GiB block of text has 65 536 tiles)
__global__ __launch_bounds__(256) void k_nl_scan(const int32_t *__restrict__ tile_count, int64_t n_tiles, int64_t *__restrict__ tile_base,
                                                 int64_t *__restrict__ total) {
    __shared__ long long sh[256];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < n_tiles; t0 += 256) {
        const int64_t t = t0 + threadIdx.x;
        const long long v = t < n_tiles ? tile_count[t] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const long long x = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += x;
            __syncthreads();
        }
        if (t < n_tiles) tile_base[t] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
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
                                                   int8_t *__restrict__ rows, int S, int32_t *__restrict__ pos_out,
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
                    if (p2 <= d0 || (dg & range) != range) bad |= TOK_BAD_POS;
                    else
                        for (int k = d0; k < p2 && v <= 0x7FFFFFFFll; ++k) v = v * 10 + (rl(ch, k) - '0');
                    if (v > 0x7FFFFFFFll) bad |= TOK_BAD_POS;
                    cells_at = ls + __builtin_ctzll(r2);
                    if (le - cells_at != (int64_t)cells_w) bad |= TOK_IRREGULAR;
                    if (lane == 0) {
                        pos_out[row] = (int32_t)(neg ? -v : v);
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
        while (p < le && text[p] >= '0' && text[p] <= '9' && v <= 0x7FFFFFFFll) { v = v * 10 + (text[p] - '0'); ++p; }
        if (p == d0 || v > 0x7FFFFFFFll || (p < le && !blank(text[p]))) bad |= TOK_BAD_POS;
        pos_out[row] = (int32_t)(neg ? -v : v);
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

}  // namespace

// Tokenise `len` bytes of complete `.geno` data lines (no header) into resident rows row_offset .. on the context's copy stream.
// The text travels to the device in pieces through two page-locked staging buffers (host threads copy piece k+1 while piece k is
// in flight).  Returns through *ok_out whether the regular-layout fast path applied; if not, nothing may be assumed about the
// rows and the caller tokenises the block on the host.
extern "C" int pg_tokenize_text(pg_ctx *c, const char *text, int64_t len, int fmt, int n_cols, int max_ploidy,
                                const int32_t *col_slot, const int32_t *col_ploidy, int64_t row_offset, int32_t *pos_out,
                                int64_t row_capacity, int64_t *run_row_out, int64_t *run_off_out, int32_t *run_len_out,
                                int64_t run_capacity, int64_t *n_rows_out, int64_t *n_runs_out, int *ok_out) {
    if (!c || (!text && len) || !col_slot || !col_ploidy || !n_rows_out || !n_runs_out || !ok_out)
        return pg_fail(PG_ERR_ARG, "pg_tokenize_text: null argument");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (fmt < PG_FMT_PHASED || fmt > PG_FMT_DIPLO || n_cols < 1 || max_ploidy < 1) return pg_fail(PG_ERR_ARG, "pg_tokenize_text: bad format description");
    *n_rows_out = 0;
    *n_runs_out = 0;
    *ok_out = 0;
    if (len == 0) { *ok_out = 1; return PG_OK; }
    bool any = false;
    for (int k = 0; k < n_cols; ++k) {
        if (col_ploidy[k] <= 0) continue;
        any = true;
        if (col_ploidy[k] > max_ploidy) return pg_fail(PG_ERR_ARG, "col_ploidy[%d]=%d exceeds max_ploidy", k, col_ploidy[k]);
        if (fmt == PG_FMT_DIPLO && col_ploidy[k] != 2) return PG_OK;
        if (fmt == PG_FMT_HAPLO && col_ploidy[k] != 1) return PG_OK;
        for (int a = 0; a < col_ploidy[k]; ++a) {
            const int s = col_slot[(size_t)k * max_ploidy + a];
            if (s < 0 || s >= c->n_hap) return pg_fail(PG_ERR_ARG, "col_slot[%d][%d]=%d out of range", k, a, s);
        }
    }
    if (!any || text[len - 1] != '\n') return PG_OK;
    // The cell widths of the block, read off its first line (scaffold, position, then n_cols cells with one blank between them):
    // a wanted column's cell must be as wide as its ploidy says (phased: 2 p - 1 characters, pairs: p, haplo / diplo: 1); a file of
    // mixed ploidy has narrower cells for its haploid samples.  Every other line is held against these widths on the device.
    std::vector<int32_t> col_geo((size_t)2 * n_cols);
    int cells_w = 0;
    {
        auto blank_h = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f'; };
        const char *p = text, *e = static_cast<const char *>(memchr(text, '\n', (size_t)len));
        for (int tok = 0; tok < 2; ++tok) {                                    // scaffold, position
            while (p < e && blank_h(*p)) ++p;
            if (p == e) return PG_OK;
            while (p < e && !blank_h(*p)) ++p;
        }
        while (p < e && blank_h(*p)) ++p;
        const char *cells0 = p;
        for (int k = 0; k < n_cols; ++k) {
            const char *b = p;
            while (p < e && !blank_h(*p)) ++p;
            const int w = (int)(p - b);
            if (w < 1) return PG_OK;
            if (col_ploidy[k] > 0) {
                const int want = fmt == PG_FMT_PHASED ? 2 * col_ploidy[k] - 1 : (fmt == PG_FMT_PAIRS ? col_ploidy[k] : 1);
                if (w != want) return PG_OK;
            }
            col_geo[(size_t)k] = (int32_t)(b - cells0);
            col_geo[(size_t)n_cols + k] = w;
            if (k + 1 < n_cols) {
                if (p == e || !blank_h(*p)) return PG_OK;
                ++p;                                                            // exactly one separator
            }
        }
        if (p != e) return PG_OK;
        cells_w = (int)(p - cells0);
    }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream_up;
    int rc;
    // ---- text to the device, double-buffered through page-locked staging ----
    if ((rc = c->tok_text.ensure((size_t)len + 32)) != PG_OK) return rc;
    const size_t piece = 32u << 20;
    for (int k = 0; k < 2; ++k)
        if ((rc = c->tok_pin[k].ensure(std::min<size_t>(piece, (size_t)len))) != PG_OK) return rc;
    if (!c->tok_ev[0]) { HIPCHK(hipEventCreateWithFlags(&c->tok_ev[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->tok_ev[1], hipEventDisableTiming)); }
    int nt = pg_host_threads();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    int64_t done = 0;
    for (int k = 0; done < len; ++k) {
        const size_t n = (size_t)std::min<int64_t>((int64_t)piece, len - done);
        uint8_t *pin = c->tok_pin[k & 1].p;
        if (k >= 2) HIPCHK(hipEventSynchronize(c->tok_ev[k & 1]));              // the copy out of this staging buffer is done
        {
            std::vector<std::thread> th;
            const int use = (int)std::min<size_t>((size_t)nt, n / (1 << 20) + 1);
            auto part = [&](int t) { const size_t a = n * t / use, b = n * (t + 1) / use; memcpy(pin + a, text + done + a, b - a); };
            for (int t = 1; t < use; ++t) th.emplace_back(part, t);
            part(0);
            for (auto &x : th) x.join();
        }
        HIPCHK(hipMemcpyAsync(c->tok_text.p + done, pin, n, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(c->tok_ev[k & 1], st));
        done += (int64_t)n;
    }
    // ---- line feeds ----
    const int64_t n_tiles = (len + NL_TILE - 1) / NL_TILE;
    if ((rc = c->tok_i32.ensure((size_t)n_tiles + 4)) != PG_OK) return rc;
    if ((rc = c->tok_i64.ensure((size_t)n_tiles + 2)) != PG_OK) return rc;
    int32_t *d_status = c->tok_i32.p + n_tiles;                                  // [0] status bits, [1] number of runs
    int64_t *d_total = c->tok_i64.p + n_tiles;
    HIPCHK(hipMemsetAsync(d_status, 0, 8, st));
    hipLaunchKernelGGL(k_nl_count, dim3((unsigned)n_tiles), dim3(256), 0, st, c->tok_text.p, len, c->tok_i32.p);
    hipLaunchKernelGGL(k_nl_scan, dim3(1), dim3(256), 0, st, c->tok_i32.p, n_tiles, c->tok_i64.p, d_total);
    int64_t n_lines = 0;
    HIPCHK(hipMemcpyAsync(&n_lines, d_total, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *n_rows_out = n_lines;
    if (n_lines == 0) { *ok_out = 1; return PG_OK; }
    if (n_lines > row_capacity || row_offset < 0 || row_offset + n_lines > c->cap_sites) return PG_OK;   // (caller sizes from pg_count_lines)
    if (!pos_out || !run_row_out || !run_off_out || !run_len_out || run_capacity < 1) return pg_fail(PG_ERR_ARG, "pg_tokenize_text: null output");
    if ((rc = c->tok_nl.ensure((size_t)n_lines)) != PG_OK) return rc;
    hipLaunchKernelGGL(k_nl_write, dim3((unsigned)n_tiles), dim3(256), 0, st, c->tok_text.p, len, c->tok_i64.p, c->tok_nl.p);
    // ---- parse ----
    if ((rc = c->tok_cols.ensure((size_t)n_cols * (max_ploidy + 3))) != PG_OK) return rc;
    HIPCHK(hipMemcpyAsync(c->tok_cols.p, col_slot, (size_t)n_cols * max_ploidy * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->tok_cols.p + (size_t)n_cols * max_ploidy, col_ploidy, (size_t)n_cols * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->tok_cols.p + (size_t)n_cols * (max_ploidy + 1), col_geo.data(), (size_t)n_cols * 8, hipMemcpyHostToDevice, st));
    const int64_t run_cap = std::min<int64_t>(run_capacity, n_lines);
    if ((rc = c->tok_pos.ensure((size_t)n_lines + (size_t)run_cap + 8)) != PG_OK) return rc;   // pos [n_lines] + run_len [run_cap] (int32 each)
    if ((rc = c->tok_off.ensure((size_t)run_cap * 2)) != PG_OK) return rc;                      // run_row, run_off
    HIPCHK(hipMemsetAsync(c->gt.p + row_offset * c->S, 0, (size_t)n_lines * c->S, st));
    DipTable dip;
    memset(dip.v, 0, sizeof(dip.v));
    {
        const char *d = "ACGKMNSRTWY";
        const char *pr[] = {"AA", "CC", "GG", "GT", "AC", "NN", "CG", "AG", "TT", "AT", "CT"};
        auto code = [](char ch) { return ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 4 : ch == 'T' ? 8 : 0; };
        for (int k = 0; d[k]; ++k) dip.v[(int)d[k]] = (uint8_t)(code(pr[k][0]) | (code(pr[k][1]) << 4));
    }
    hipLaunchKernelGGL(k_tok_parse, dim3((unsigned)((n_lines + 3) / 4)), dim3(256), 0, st, c->tok_text.p, c->tok_nl.p, n_lines, fmt, n_cols,
                       cells_w, max_ploidy, c->tok_cols.p, c->tok_cols.p + (size_t)n_cols * max_ploidy,
                       c->tok_cols.p + (size_t)n_cols * (max_ploidy + 1), c->tok_cols.p + (size_t)n_cols * (max_ploidy + 2),
                       c->gt.p + row_offset * c->S, c->S,
                       c->tok_pos.p, c->tok_off.p, c->tok_off.p + run_cap, c->tok_pos.p + n_lines, d_status + 1, run_cap, d_status, dip);
    HIPCHK(hipGetLastError());
    int32_t status[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(status, d_status, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pos_out, c->tok_pos.p, (size_t)n_lines * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (status[0] != 0) return PG_OK;
    const int64_t nr = status[1];
    *n_runs_out = nr;
    if (nr > run_cap) return PG_OK;                                             // more runs than the caller has room for
    std::vector<int64_t> rr((size_t)nr), ro((size_t)nr);
    std::vector<int32_t> rlen((size_t)nr);
    if (nr) {
        HIPCHK(hipMemcpyAsync(rr.data(), c->tok_off.p, (size_t)nr * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(ro.data(), c->tok_off.p + run_cap, (size_t)nr * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(rlen.data(), c->tok_pos.p + n_lines, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
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
