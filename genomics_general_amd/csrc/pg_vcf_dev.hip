// VCF lines -> `.geno` rows on the device: the parseVCF.py drop-in's parser for the regular spelling of a VCF line (SURVEY.md 8f row
// 4; VCF_processing/parseVCF.py:49-191 VcfSite / getGenotype, 367-370 the site filters, 380-383 the output line).  The per-line and
// per-cell logic is csrc/pg_vcf_core.h (compiled a second time into tests/vcf_emul.cpp, which the CPU suite holds against the host
// parser); this file is the division of the work over the chip and the host side of the four entry points.
//
//   k_vcf_heads      a thread per line: the nine fixed columns (token ends, site filters, POS, the allele table, site type, where GT
//                    and the filtered fields stand in FORMAT) -> a PgvLine record; the size of the row when every allele is one base
//   k_vcf_cells<0>   a wavefront per COMPLEX kept line (some allele longer or shorter than a base): the row's size
//   k_vcf_scan       one block: the rows' places in the output (exclusive sums of the sizes), their total, their number
//   k_vcf_cells<1>   a wavefront per kept line: (1) the tabs behind FORMAT, 16 bytes per lane and step, ranked by a wave scan, their
//                    positions into LDS; (2) a lane per selected sample: its column walked once (pgv_cell: the GT piece, the filtered
//                    pieces as decimals, the allele look-up), its characters stored at the row's place + the cell's offset (a table
//                    for rows of single bases, a wave scan of the cells' sizes for complex rows)
// The text never crosses PCIe as text when the input is bgzipped: k_inflate (pg_inflate.hip) writes it into the tokenizer's text slot
// and lists its line feeds; only the rows come back.  A line the device does not take (PGV_HOST) hands the BLOCK to the host parser
// (pg_vcf_dev_collect reports the line; pg_vcf_dev_text brings the block's text back).
#include "pg_ctx.h"
#include "pg_vcf_cfg.h"

#include <algorithm>
#include <chrono>
#include <cstring>

#include <unistd.h>

int pg_tok_text_submit(pg_ctx *c, int slot, const char *text, int fd, int64_t file_offset, int64_t len);
int pg_tok_bgzf_submit(pg_ctx *c, int slot, const uint8_t *comp, int64_t comp_len, const uint32_t *in_off, const uint32_t *in_len,
                       const uint32_t *out_len, const uint32_t *crc, int64_t n_members, const char *head, int64_t head_len,
                       int64_t text_len, int64_t line_len_hint);
int pg_tok_lines(pg_ctx *c, int slot, int64_t *n_lines_out);
int pg_tok_crc_result(pg_ctx *c, int slot);
int pg_deflate_queue(pg_ctx *c, hipStream_t st, pg_ctx::Deflate &D, const uint8_t *text_d, const long long *total_d, int64_t max_text,
                     const long long *status_d, long long *comp_total_d);

namespace {

#define PGV_ST_HOST 1ll
#define PGV_ST_OVERFLOW 2ll

__device__ inline void raise_host(long long *status, long long line) {
    atomicOr(reinterpret_cast<unsigned long long *>(status), (unsigned long long)PGV_ST_HOST);
    atomicMin(status + 1, line);
}

__global__ __launch_bounds__(256) void k_vcf_heads(const uint8_t *__restrict__ text, const int64_t *__restrict__ nl, int64_t n_lines,
                                                   PgvConfig cfg, const uint8_t *__restrict__ contigs, PgvLine *__restrict__ lines,
                                                   uint32_t *__restrict__ rlen, long long *__restrict__ status, const PgvKey *__restrict__ prev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lines) return;
    const int64_t ls = i ? nl[i - 1] + 1 : 0, le = nl[i];
    PgvLine *L = lines + i;
    uint32_t bytes = 0;
    if ((uint64_t)(le - ls) > 0xfffffff0ull) {
        L->flags = 0;
        raise_host(status, i);
    } else {
        if (pgv_head(text + ls, (uint32_t)(le - ls), cfg, contigs, L) != PGV_OK) {
            L->flags = 0;
            raise_host(status, i);
        }
        if ((L->flags & PGV_LINE_KEPT) && (cfg.flags & PGV_EXCLUDE_DUPLICATES)) {
            // --excludeDuplicates: against the data line before this one -- the lines above it in the block, then the key the blocks
            // before left (k_vcf_lastkey)
            const int dup = pgv_is_duplicate(text + ls, *L, *prev, [&](uint32_t j, const uint8_t **bt, uint32_t *bn) {
                if ((int64_t)j >= i) return false;
                const int64_t b = i - 1 - (int64_t)j;
                const int64_t bs = b ? nl[b - 1] + 1 : 0;
                *bt = text + bs;
                *bn = (uint32_t)(nl[b] - bs);                    // (a line of 4 GB raises the flag on its own thread)
                return true;
            });
            if (dup) L->flags = 0;
            if (dup == 2) raise_host(status, i);
        }
        if ((L->flags & PGV_LINE_KEPT) && !(L->flags & PGV_LINE_COMPLEX)) bytes = L->fixed_len + (uint32_t)cfg.plain_cells;
    }
    rlen[i] = bytes;
}

// the CHROM and POS tokens of the block's last data line -> *key (what the next block's first lines are held against); a block
// without a data line leaves the key as it is
// (a last data line the key cannot hold -- an irregular spelling, tokens of more than PGV_KEY_MAX characters -- sends ITS block to the
// host parser too, so that the host holds the key the device lacks)
__global__ void k_vcf_lastkey(const uint8_t *__restrict__ text, const int64_t *__restrict__ nl, int64_t n_lines, PgvKey *__restrict__ key,
                              long long *__restrict__ status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int64_t i = n_lines - 1; i >= 0; --i) {
        const int64_t ls = i ? nl[i - 1] + 1 : 0;
        const uint64_t n = (uint64_t)(nl[i] - ls);
        uint32_t cl = 0, po = 0, pl = 0;
        const int r = n > 0xfffffff0ull ? 2 : pgv_line_key(text + ls, (uint32_t)n, &cl, &po, &pl);
        if (r == 1) continue;
        if (r == 2 || cl > PGV_KEY_MAX || pl > PGV_KEY_MAX) {
            key->chrom_len = PGV_KEY_UNKNOWN;
            raise_host(status, i);
            return;
        }
        key->chrom_len = cl;
        key->pos_len = pl;
        for (uint32_t k = 0; k < cl; ++k) key->chrom[k] = text[ls + k];
        for (uint32_t k = 0; k < pl; ++k) key->pos[k] = text[ls + po + k];
        return;
    }
}

__device__ inline int wave_incl_scan(int x, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// 0x80 in every byte of x that equals the byte b repeated in `pat` / that is below 0x21 -- exactly (no borrow between bytes)
__device__ inline uint32_t bytes_eq(uint32_t x, uint32_t pat) {
    const uint32_t y = x ^ pat;
    return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
}
__device__ inline uint32_t bytes_blank(uint32_t x) { return ~(((x & 0x7f7f7f7fu) + 0x5f5f5f5fu) | x) & 0x80808080u; }
__device__ inline uint32_t nibble_of(uint32_t m) {             // bits 7, 15, 23, 31 -> bits 0 .. 3
    const uint32_t b = m >> 7;
    return (b | b >> 7 | b >> 14 | b >> 21) & 0xfu;
}

template <int RENDER>
__global__ __launch_bounds__(256) void k_vcf_cells(const uint8_t *__restrict__ text, const int64_t *__restrict__ nl, int64_t n_lines, PgvConfig cfg,
                                                   const int32_t *__restrict__ sel_col, const uint8_t *__restrict__ ploidy_of,
                                                   const uint8_t *__restrict__ fsel_of, const uint32_t *__restrict__ cell_off,
                                                   const PgvLine *__restrict__ lines, uint32_t *__restrict__ rlen, const int64_t *__restrict__ roff,
                                                   uint8_t *__restrict__ out, long long *__restrict__ status, int waves_per_block) {
    extern __shared__ uint32_t sh_tabs[];
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    if (wave >= waves_per_block) return;
    const int64_t i = (int64_t)blockIdx.x * waves_per_block + wave;
    if (i >= n_lines) return;
    const PgvLine &L = lines[i];
    const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.flags);
    if (!(flags & PGV_LINE_KEPT)) return;
    const bool cx = (flags & PGV_LINE_COMPLEX) != 0;
    if (!RENDER && !cx) return;
    if (RENDER && status[0] != 0) return;                        // the block goes to the host: nothing of `out` is read
    const int n_cols = cfg.n_vcf_samples;
    uint32_t *tabs = sh_tabs + (size_t)wave * (size_t)n_cols;
    const int64_t ls = i ? nl[i - 1] + 1 : 0;
    const uint8_t *t = text + ls;
    const uint32_t cells_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.cells_off);
    const uint32_t line_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.line_len);
    bool host = false;
    // ---- (1) the tabs that end the sample columns ----
    int n_tabs = 0;
    {
        const uintptr_t abs0 = reinterpret_cast<uintptr_t>(t + cells_off);
        const int64_t rel0 = (int64_t)cells_off - (int64_t)(abs0 & 15);             // line offset of the first aligned 16 bytes
        for (int64_t g0 = rel0; g0 < (int64_t)line_len && n_tabs < n_cols; g0 += 1024) {
            const int64_t rel = g0 + lane * 16;
            uint32_t tm = 0, bm = 0;
            if (rel < (int64_t)line_len) {
                const uint4 q = *reinterpret_cast<const uint4 *>(t + rel);
                tm = nibble_of(bytes_eq(q.x, 0x09090909u)) | nibble_of(bytes_eq(q.y, 0x09090909u)) << 4 |
                     nibble_of(bytes_eq(q.z, 0x09090909u)) << 8 | nibble_of(bytes_eq(q.w, 0x09090909u)) << 12;
                bm = nibble_of(bytes_blank(q.x)) | nibble_of(bytes_blank(q.y)) << 4 | nibble_of(bytes_blank(q.z)) << 8 | nibble_of(bytes_blank(q.w)) << 12;
                uint32_t valid = 0xffffu;
                if (rel < (int64_t)cells_off) valid &= 0xffffu << (uint32_t)((int64_t)cells_off - rel);
                if (rel + 16 > (int64_t)line_len) valid &= 0xffffu >> (uint32_t)(rel + 16 - (int64_t)line_len);
                tm &= valid;
                bm &= valid & ~tm;
            }
            const int cnt = __popc(tm);
            const int incl = wave_incl_scan(cnt, lane);
            const int total = __shfl(incl, 63, 64);
            int r = n_tabs + incl - cnt;
            if (bm) {                                            // another blank among the columns the header names: str.split()'s business
                uint32_t b = bm;
                while (b) {
                    const int j = __ffs((int)b) - 1;
                    if (r + __popc(tm & ((1u << j) - 1u)) < n_cols) host = true;
                    b &= b - 1;
                }
            }
            while (tm) {
                const int j = __ffs((int)tm) - 1;
                if (r < n_cols) tabs[r] = (uint32_t)(rel + j);
                ++r;
                tm &= tm - 1;
            }
            n_tabs += total;
        }
        if (n_tabs > n_cols) n_tabs = n_cols;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (n_tabs < n_cols - 1) host = true;                    // fewer columns than the #CHROM line names
        else
            for (int c0 = 0; c0 < n_cols; c0 += 64) {           // none of them empty
                const int c = c0 + lane;
                if (c < n_cols) {
                    const uint32_t a = c ? tabs[c - 1] + 1 : cells_off;
                    const uint32_t b = c < n_tabs ? tabs[c] : line_len;
                    if (b <= a) host = true;
                }
            }
    }
    if (__builtin_amdgcn_ballot_w64(host) != 0) {
        if (lane == 0) raise_host(status, (long long)i);
        return;
    }
    // ---- (2) the selected samples' cells ----
    const uint32_t fixed_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.fixed_len);
    uint8_t *row = RENDER ? out + roff[i] : nullptr;
    if (RENDER) {
        // CHROM sep POS sep [REF sep]
        const uint32_t chrom_len = L.chrom_len, pos_off = L.pos_off, pos_len = L.pos_len;
        for (uint32_t k = (uint32_t)lane; k < chrom_len; k += 64) row[k] = t[k];
        for (uint32_t k = (uint32_t)lane; k < pos_len; k += 64) row[chrom_len + 1 + k] = t[pos_off + k];
        if (lane == 0) {
            row[chrom_len] = (uint8_t)cfg.sep;
            row[chrom_len + 1 + pos_len] = (uint8_t)cfg.sep;
        }
        if (cfg.add_ref) {
            const uint32_t ref_off = L.al_off[0], ref_len = L.al_len[0], at = chrom_len + 2 + pos_len;
            for (uint32_t k = (uint32_t)lane; k < ref_len; k += 64) row[at + k] = t[ref_off + k];
            if (lane == 0) row[at + ref_len] = (uint8_t)cfg.sep;
        }
    }
    const int n_sel = cfg.n_sel;
    uint32_t run = fixed_len;                                    // complex rows: where the next chunk of cells starts
    for (int s0 = 0; s0 < n_sel; s0 += 64) {
        const int s = s0 + lane;
        const bool active = s < n_sel;
        PgvCell cell;
        cell.c0 = cell.c1 = (uint8_t)cfg.missing;
        cell.phase = '/';
        cell.a0 = cell.a1 = -1;
        int pl = 1;
        uint32_t bytes = 0;
        if (active) {
            const int col = sel_col[s];
            pl = ploidy_of[s];
            const uint32_t a = col ? tabs[col - 1] + 1 : cells_off;
            const uint32_t b = col < n_tabs ? tabs[col] : line_len;
            if (pgv_cell(t, a, b, L, cfg, pl, fsel_of[s], &cell) != PGV_OK) host = true;
            else if (cx) bytes = pgv_cell_bytes(L, cell, pl);
        }
        if (cx) {
            const int incl = wave_incl_scan((int)bytes, lane);
            const uint32_t at = run + (uint32_t)incl - bytes;
            run += (uint32_t)__shfl(incl, 63, 64);
            if (RENDER && active && !host) pgv_cell_put(t, L, cfg, cell, pl, true, s + 1 == n_sel, row + at);
        } else if (RENDER && active && !host)
            pgv_cell_put(t, L, cfg, cell, pl, false, s + 1 == n_sel, row + fixed_len + cell_off[s]);
    }
    if (__builtin_amdgcn_ballot_w64(host) != 0) {
        if (lane == 0) raise_host(status, (long long)i);
        return;
    }
    if (!RENDER && lane == 0) rlen[i] = run;
}

// the rows' places: exclusive sums of rlen over the lines (one block, every thread a stretch of lines), their total and number;
// a total beyond the output buffer raises PGV_ST_OVERFLOW
__global__ __launch_bounds__(1024) void k_vcf_scan(const uint32_t *__restrict__ rlen, int64_t n_lines, int64_t *__restrict__ roff,
                                                    long long *__restrict__ status, int64_t out_cap) {
    __shared__ long long sh[1024], shn[1024];
    const int64_t per = (n_lines + 1023) / 1024;
    const int64_t a = std::min<int64_t>(n_lines, (int64_t)threadIdx.x * per), b = std::min<int64_t>(n_lines, a + per);
    long long mine = 0, rows = 0;
    for (int64_t k = a; k < b; ++k) {
        mine += rlen[k];
        rows += rlen[k] != 0;
    }
    sh[threadIdx.x] = mine;
    shn[threadIdx.x] = rows;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long x = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0, y = (int)threadIdx.x >= d ? shn[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += x;
        shn[threadIdx.x] += y;
        __syncthreads();
    }
    long long run = sh[threadIdx.x] - mine;
    for (int64_t k = a; k < b; ++k) {
        roff[k] = run;
        run += rlen[k];
    }
    if (threadIdx.x == 1023) {
        status[2] = sh[1023];
        status[3] = shn[1023];
        if (sh[1023] > out_cap) atomicOr(reinterpret_cast<unsigned long long *>(status), (unsigned long long)PGV_ST_OVERFLOW);
    }
}

int check_slot(pg_ctx *c, int slot, const char *who) {
    if (!c || slot < 0 || slot > 1) return pg_fail(PG_ERR_ARG, "%s: bad context or slot", who);
    if (!c->vcf.configured) return pg_fail(PG_ERR_STATE, "%s: pg_vcf_dev_config must be called first", who);
    return PG_OK;
}

}  // namespace

// The option set of the run (the arguments of pg_encode_vcf, + the output's separator and --addRefTrack).  *taken_out = 0: the device
// path does not take it (more than eight genotype filters, ...: why_out names the reason); the caller then stays
// on pg_encode_vcf.
extern "C" int pg_vcf_dev_config(pg_ctx *c, int n_vcf_samples, int n_sel, const int32_t *sel_col, const int32_t *sel_ploidy, int flags,
                                 double min_qual, int max_ref_len, const pg_vcf_filter *filters, int n_filters, const char *contigs,
                                 int n_contig_bytes, int contig_mode, char missing, char sep, int add_ref, int *taken_out,
                                 const char **why_out) {
    if (!c || !taken_out) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_config: null argument");
    *taken_out = 0;
    c->vcf.configured = false;
    if (contig_mode < 0 || contig_mode > 2 || (contig_mode && !contigs) || n_contig_bytes < 0) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_config: bad contig list");
    PgvTables tab;
    const char *why = "";
    const int ok = pgv_make_config(n_vcf_samples, n_sel, sel_col, sel_ploidy, flags, min_qual, max_ref_len, filters, n_filters,
                                   n_contig_bytes, contig_mode, missing, sep, add_ref, &c->vcf.cfg, &tab, &why);
    if (why_out) *why_out = why;
    if (ok < 0) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_config: bad sample selection or filter list");
    if (ok == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream_up;
    HIPCHK(hipStreamSynchronize(st));                            // (tables of an earlier configuration may still be read)
    int rc;
    if ((rc = c->vcf.sel_col.ensure((size_t)n_sel)) != PG_OK || (rc = c->vcf.ploidy.ensure((size_t)n_sel)) != PG_OK ||
        (rc = c->vcf.fsel.ensure((size_t)n_sel)) != PG_OK || (rc = c->vcf.cell_off.ensure((size_t)n_sel)) != PG_OK ||
        (rc = c->vcf.contigs.ensure((size_t)n_contig_bytes + 1)) != PG_OK)
        return rc;
    HIPCHK(hipMemcpy(c->vcf.sel_col.p, tab.sel_col.data(), (size_t)n_sel * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->vcf.ploidy.p, tab.ploidy.data(), (size_t)n_sel, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->vcf.fsel.p, tab.fsel.data(), (size_t)n_sel, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->vcf.cell_off.p, tab.cell_off.data(), (size_t)n_sel * 4, hipMemcpyHostToDevice));
    if (n_contig_bytes) HIPCHK(hipMemcpy(c->vcf.contigs.p, contigs, (size_t)n_contig_bytes, hipMemcpyHostToDevice));
    // four lines per block while their tab positions fit 60 KB of LDS, else fewer
    const size_t per_wave = (size_t)n_vcf_samples * 4;
    c->vcf.waves_per_block = per_wave * 4 <= 60 * 1024 ? 4 : (per_wave * 2 <= 60 * 1024 ? 2 : 1);
    if ((rc = c->vcf.prevkey.ensure(sizeof(PgvKey))) != PG_OK) return rc;
    {
        PgvKey none;
        memset(&none, 0, sizeof(none));
        none.chrom_len = PGV_KEY_NONE;
        HIPCHK(hipMemcpy(c->vcf.prevkey.p, &none, sizeof(none), hipMemcpyHostToDevice));
    }
    c->vcf.configured = true;
    *taken_out = 1;
    return PG_OK;
}

// --excludeDuplicates: the CHROM and POS tokens of the data line before the next block submitted (a block the HOST parsed: the device
// carries the key on from the blocks it sees itself); chrom == NULL: none
extern "C" int pg_vcf_dev_set_prev(pg_ctx *c, const char *chrom, int chrom_len, const char *pos, int pos_len) {
    if (!c || !c->vcf.configured) return pg_fail(PG_ERR_STATE, "pg_vcf_dev_set_prev: pg_vcf_dev_config must be called first");
    if (chrom_len < 0 || pos_len < 0 || (chrom && !pos)) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_set_prev: bad argument");
    PgvKey key;
    memset(&key, 0, sizeof(key));
    if (!chrom) key.chrom_len = PGV_KEY_NONE;
    else if (chrom_len > PGV_KEY_MAX || pos_len > PGV_KEY_MAX) key.chrom_len = PGV_KEY_UNKNOWN;
    else {
        key.chrom_len = (uint32_t)chrom_len;
        key.pos_len = (uint32_t)pos_len;
        memcpy(key.chrom, chrom, (size_t)chrom_len);
        memcpy(key.pos, pos, (size_t)pos_len);
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream_up));                  // (behind every block already queued)
    HIPCHK(hipMemcpy(c->vcf.prevkey.p, &key, sizeof(key), hipMemcpyHostToDevice));
    return PG_OK;
}

// the key the block collected from `slot` started from (for a block that goes to the host parser under --excludeDuplicates):
// *chrom_len_out == -1: none; -2: a line the key could not hold -- the block it ended went to the host parser for that reason, whose
// last key the caller has.  chrom_out / pos_out: PGV_KEY_MAX (120) bytes each
extern "C" int pg_vcf_dev_prev(pg_ctx *c, int slot, char *chrom_out, int *chrom_len_out, char *pos_out, int *pos_len_out) {
    int rc = check_slot(c, slot, "pg_vcf_dev_prev");
    if (rc != PG_OK) return rc;
    if (!chrom_out || !chrom_len_out || !pos_out || !pos_len_out) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_prev: null argument");
    const PgvKey *key = reinterpret_cast<const PgvKey *>(c->vcf.s[slot].h_prev.p);
    *chrom_len_out = *pos_len_out = -1;
    if (!key || key->chrom_len == PGV_KEY_NONE) return PG_OK;
    if (key->chrom_len == PGV_KEY_UNKNOWN) {                     // the block before went to the host parser because of that very line: the caller holds its key
        *chrom_len_out = *pos_len_out = -2;
        return PG_OK;
    }
    *chrom_len_out = (int)key->chrom_len;
    *pos_len_out = (int)key->pos_len;
    memcpy(chrom_out, key->chrom, key->chrom_len);
    memcpy(pos_out, key->pos, key->pos_len);
    return PG_OK;
}

// A block of whole lines (the last byte a line feed) to the device, into text slot `slot`: from memory (`text`) or from a file
// (text == NULL: len bytes at file_offset of fd); returns when the bytes have landed, the line feeds are being counted.
extern "C" int pg_vcf_dev_submit(pg_ctx *c, int slot, const char *text, int fd, int64_t file_offset, int64_t len) {
    int rc = check_slot(c, slot, "pg_vcf_dev_submit");
    if (rc != PG_OK) return rc;
    if ((!text && fd < 0 && len) || file_offset < 0 || len < 0) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_submit: no text");
    pg_ctx::VcfDev::Slot &V = c->vcf.s[slot];
    V.text_len = len;
    V.no_final_newline = false;
    if (len > 0) {
        char last = 0;
        if (text) last = text[len - 1];
        else if (pread(fd, &last, 1, (off_t)(file_offset + len - 1)) != 1) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_submit: cannot read the input");
        V.no_final_newline = last != '\n';                      // (the end of a file without a final line feed: that block is the host's)
    }
    if ((rc = pg_tok_text_submit(c, slot, text, fd, file_offset, len)) != PG_OK) return rc;
    V.state = len ? 1 : 3;
    return PG_OK;
}

// The same for a block that is still bgzipped (the arguments of pg_tokenize_submit_bgzf): the members cross PCIe deflated, k_inflate
// writes their text behind `head` and lists its line feeds.  Asynchronous when comp is page-locked.
extern "C" int pg_vcf_dev_submit_bgzf(pg_ctx *c, int slot, const uint8_t *comp, int64_t comp_len, const uint32_t *in_off, const uint32_t *in_len,
                                      const uint32_t *out_len, const uint32_t *crc, int64_t n_members, const char *head, int64_t head_len,
                                      int64_t text_len, int64_t line_len_hint, int last_is_newline) {
    int rc = check_slot(c, slot, "pg_vcf_dev_submit_bgzf");
    if (rc != PG_OK) return rc;
    pg_ctx::VcfDev::Slot &V = c->vcf.s[slot];
    V.text_len = text_len;
    V.no_final_newline = text_len > 0 && !last_is_newline;
    if ((rc = pg_tok_bgzf_submit(c, slot, comp, comp_len, in_off, in_len, out_len, crc, n_members, head, head_len, text_len,
                                 line_len_hint > 0 ? line_len_hint : 64)) != PG_OK) return rc;
    V.state = text_len ? 1 : 3;
    return PG_OK;
}

// Queues the kernels of the block in `slot` (waits for the number of its lines only).
extern "C" int pg_vcf_dev_parse(pg_ctx *c, int slot) {
    int rc = check_slot(c, slot, "pg_vcf_dev_parse");
    if (rc != PG_OK) return rc;
    pg_ctx::VcfDev &D = c->vcf;
    pg_ctx::VcfDev::Slot &V = D.s[slot];
    if (V.state == 3) return PG_OK;
    if (V.state != 1) return pg_fail(PG_ERR_STATE, "pg_vcf_dev_parse: nothing submitted to slot %d", slot);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream_up;
    pg_ctx::TokSlot &T = c->tok[slot];
    int64_t n_lines = 0;
    if ((rc = pg_tok_lines(c, slot, &n_lines)) != PG_OK) { V.state = 0; return rc; }
    if ((rc = V.status.ensure(8)) != PG_OK || (rc = V.h_status.ensure(8)) != PG_OK) return rc;
    if (!V.done) HIPCHK(hipEventCreateWithFlags(&V.done, hipEventDisableTiming));
    V.h_status.p[0] = V.no_final_newline || n_lines == 0 ? PGV_ST_HOST : 0;     // (no line feed at all: one unfinished line)
    V.h_status.p[1] = V.no_final_newline || n_lines == 0 ? 0 : 0x7fffffffffffffffll;
    V.h_status.p[2] = V.h_status.p[3] = V.h_status.p[4] = 0;
    if (D.cfg.flags & PGV_EXCLUDE_DUPLICATES) {                  // (the key this block starts from, for a block that goes to the host after all)
        if ((rc = V.h_prev.ensure(sizeof(PgvKey))) != PG_OK) return rc;
        HIPCHK(hipMemcpyAsync(V.h_prev.p, D.prevkey.p, sizeof(PgvKey), hipMemcpyDeviceToHost, st));
    }
    if (V.h_status.p[0]) {                                       // nothing to queue: collect reports the block as the host's
        V.state = 2;
        HIPCHK(hipEventRecord(V.done, st));
        return PG_OK;
    }
    if ((rc = V.lines.ensure_roomy((size_t)n_lines * sizeof(PgvLine))) != PG_OK || (rc = V.rlen.ensure_roomy((size_t)n_lines)) != PG_OK ||
        (rc = V.roff.ensure_roomy((size_t)n_lines)) != PG_OK)
        return rc;
    // rows of single bases are shorter than their lines; complex rows (allele strings per call) can be longer: room for the text's
    // length again on top of the plain rows' bound, a total beyond it sends the block to the host
    V.out_cap = T.len + n_lines * (int64_t)(D.cfg.plain_cells + 64) + 4096;
    if ((rc = V.out.ensure_roomy((size_t)V.out_cap)) != PG_OK) return rc;
    HIPCHK(hipMemcpyAsync(V.status.p, V.h_status.p, 40, hipMemcpyHostToDevice, st));
    PgvLine *lines = reinterpret_cast<PgvLine *>(V.lines.p);
    long long *status = reinterpret_cast<long long *>(V.status.p);
    const PgvKey *prev = reinterpret_cast<const PgvKey *>(D.prevkey.p);
    hipLaunchKernelGGL(k_vcf_heads, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, st, T.tp, T.nl.p, n_lines, D.cfg, D.contigs.p, lines,
                       V.rlen.p, status, prev);
    if (D.cfg.flags & PGV_EXCLUDE_DUPLICATES)
        hipLaunchKernelGGL(k_vcf_lastkey, dim3(1), dim3(64), 0, st, T.tp, T.nl.p, n_lines, reinterpret_cast<PgvKey *>(D.prevkey.p), status);
    const int wpb = D.waves_per_block;
    const dim3 grid((unsigned)((n_lines + wpb - 1) / wpb));
    const size_t lds = (size_t)wpb * (size_t)D.cfg.n_vcf_samples * 4;
    hipLaunchKernelGGL((k_vcf_cells<0>), grid, dim3(64 * wpb), lds, st, T.tp, T.nl.p, n_lines, D.cfg, D.sel_col.p, D.ploidy.p, D.fsel.p,
                       D.cell_off.p, lines, V.rlen.p, V.roff.p, V.out.p, status, wpb);
    hipLaunchKernelGGL(k_vcf_scan, dim3(1), dim3(1024), 0, st, V.rlen.p, n_lines, V.roff.p, status, V.out_cap);
    hipLaunchKernelGGL((k_vcf_cells<1>), grid, dim3(64 * wpb), lds, st, T.tp, T.nl.p, n_lines, D.cfg, D.sel_col.p, D.ploidy.p, D.fsel.p,
                       D.cell_off.p, lines, V.rlen.p, V.roff.p, V.out.p, status, wpb);
    HIPCHK(hipGetLastError());
    // the rows deflated where they lie (k_deflate reads their size on the device; a raised status cancels it) -- on the context's second
    // stream, beside the next block's k_inflate and parse kernels on the copy stream: a member costs k_deflate's wavefront 7.6 ms
    // whatever else runs, and a block's rows seldom fill the chip's wave slots
    hipStream_t fin = st;
    if (D.bgzf_rows) {
        static const bool aside = !(getenv("PG_DEFLATE_STREAM") && atoi(getenv("PG_DEFLATE_STREAM")) == 0);
        if (aside && c->stream2) {
            if (!V.rows_ready) HIPCHK(hipEventCreateWithFlags(&V.rows_ready, hipEventDisableTiming));
            HIPCHK(hipEventRecord(V.rows_ready, st));
            HIPCHK(hipStreamWaitEvent(c->stream2, V.rows_ready, 0));
            fin = c->stream2;
        }
        if ((rc = pg_deflate_queue(c, fin, V.df, V.out.p, status + 2, V.out_cap, status, status + 4)) != PG_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(V.h_status.p, V.status.p, 40, hipMemcpyDeviceToHost, fin));
    HIPCHK(hipEventRecord(V.done, fin));
    V.state = 2;
    ++D.blocks;
    return PG_OK;
}

// Waits for the block's kernels.  *host_line_out < 0: the rows are ready (*out_len_out bytes, *n_rows_out rows: pg_vcf_dev_rows);
// else the block goes to the host parser -- line *host_line_out is the first the device does not take (or the rows would not fit).
extern "C" int pg_vcf_dev_collect(pg_ctx *c, int slot, int64_t *out_len_out, int64_t *n_rows_out, int64_t *host_line_out, int64_t *bgzf_len_out) {
    int rc = check_slot(c, slot, "pg_vcf_dev_collect");
    if (rc != PG_OK) return rc;
    if (!out_len_out || !n_rows_out || !host_line_out) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_collect: null argument");
    pg_ctx::VcfDev::Slot &V = c->vcf.s[slot];
    *out_len_out = *n_rows_out = 0;
    *host_line_out = -1;
    if (bgzf_len_out) *bgzf_len_out = 0;
    if (V.state == 3) { V.state = 0; return PG_OK; }
    if (V.state != 2) return pg_fail(PG_ERR_STATE, "pg_vcf_dev_collect: nothing parsed in slot %d", slot);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(V.done));
    V.state = 0;
    if ((rc = pg_tok_crc_result(c, slot)) != PG_OK) return rc;
    if (V.h_status.p[0]) {
        *host_line_out = (V.h_status.p[0] & PGV_ST_HOST) ? V.h_status.p[1] : 0;
        ++c->vcf.host_blocks;
        return PG_OK;
    }
    *out_len_out = V.h_status.p[2];
    *n_rows_out = V.h_status.p[3];
    if (bgzf_len_out && c->vcf.bgzf_rows) *bgzf_len_out = V.h_status.p[4];
    return PG_OK;
}

// bgzf_members != 0: the rows of every block parsed from now on are also deflated on the device (k_deflate: BGZF members of 65 280 bytes
// of rows, no end-of-file member); pg_vcf_dev_collect then reports their bytes and pg_vcf_dev_rows_bgzf fetches them
extern "C" int pg_vcf_dev_set_output(pg_ctx *c, int bgzf_members) {
    if (!c) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_set_output: null context");
    c->vcf.bgzf_rows = bgzf_members != 0;
    return PG_OK;
}

extern "C" int pg_vcf_dev_rows_bgzf(pg_ctx *c, int slot, uint8_t *dst, int64_t len) {
    int rc = check_slot(c, slot, "pg_vcf_dev_rows_bgzf");
    if (rc != PG_OK) return rc;
    pg_ctx::VcfDev::Slot &V = c->vcf.s[slot];
    if (len < 0 || (len && !dst) || (size_t)len > V.df.comp.cap) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_rows_bgzf: bad length");
    if (len == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    if (!c->tok_small) HIPCHK(hipStreamCreateWithFlags(&c->tok_small, hipStreamNonBlocking));
    HIPCHK(hipMemcpyAsync(dst, V.df.comp.p, (size_t)len, hipMemcpyDeviceToHost, c->tok_small));
    HIPCHK(hipStreamSynchronize(c->tok_small));
    return PG_OK;
}

// the rows of the collected block -> dst (len = what collect reported; page-locked dst: one DMA at the link's rate)
extern "C" int pg_vcf_dev_rows(pg_ctx *c, int slot, uint8_t *dst, int64_t len) {
    int rc = check_slot(c, slot, "pg_vcf_dev_rows");
    if (rc != PG_OK) return rc;
    pg_ctx::VcfDev::Slot &V = c->vcf.s[slot];
    if (len < 0 || (len && !dst) || (size_t)len > V.out.cap) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_rows: bad length");
    if (len == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    if (!c->tok_small) HIPCHK(hipStreamCreateWithFlags(&c->tok_small, hipStreamNonBlocking));
    HIPCHK(hipMemcpyAsync(dst, V.out.p, (size_t)len, hipMemcpyDeviceToHost, c->tok_small));     // (beside the next block's inflate on the copy stream)
    HIPCHK(hipStreamSynchronize(c->tok_small));
    return PG_OK;
}

// the text of the collected block -> dst (a block that goes to the host parser and whose text the host never had: BGZF)
extern "C" int pg_vcf_dev_text(pg_ctx *c, int slot, uint8_t *dst, int64_t len) {
    int rc = check_slot(c, slot, "pg_vcf_dev_text");
    if (rc != PG_OK) return rc;
    pg_ctx::TokSlot &T = c->tok[slot];
    if (len < 0 || (len && !dst) || len > T.len || !T.tp) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_text: bad length");
    if (len == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    if (!c->tok_small) HIPCHK(hipStreamCreateWithFlags(&c->tok_small, hipStreamNonBlocking));
    HIPCHK(hipMemcpyAsync(dst, T.tp, (size_t)len, hipMemcpyDeviceToHost, c->tok_small));
    HIPCHK(hipStreamSynchronize(c->tok_small));
    return PG_OK;
}

// blocks parsed on the device / handed to the host so far
extern "C" int pg_vcf_dev_stats(pg_ctx *c, int64_t *blocks_out, int64_t *host_blocks_out) {
    if (!c || !blocks_out || !host_blocks_out) return pg_fail(PG_ERR_ARG, "pg_vcf_dev_stats: null argument");
    *blocks_out = c->vcf.blocks;
    *host_blocks_out = c->vcf.host_blocks;
    return PG_OK;
}
