// The device path of the parseVCF.py drop-in walked on the host: the same pgv_head / pgv_cell / pgv_cell_put the kernels of
// genomics_general_amd/csrc/pg_vcf_dev.hip are made of (csrc/pg_vcf_core.h, compiled here by g++), configured by the same
// pgv_make_config, with the kernels' division of the work restated serially: k_vcf_heads = the loop over lines, k_vcf_cells<0> = the
// sizes of complex rows, k_vcf_scan = the running offset, k_vcf_cells<1> = the rows' text.  tests/test_vcf.py holds its output
// against the host parser (pg_encode_vcf + pg_vcf_render_rows) on every golden and on random files.  Test infrastructure.
#include "../genomics_general_amd/csrc/pg_vcf_cfg.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace {

// the tab positions of a kept line behind cells_off (what phase 1 of k_vcf_cells leaves in LDS): tabs[c] ends sample column c.  The
// header's n_cols sample columns must all be there, none of them empty, no other blank among them (str.split() would cut the line
// differently: the host's business); what follows them is not looked at, as the reference does not.  false: the line needs the host
bool line_tabs(const uint8_t *t, const PgvLine &L, int n_cols, std::vector<uint32_t> &tabs) {
    tabs.clear();
    for (uint32_t p = L.cells_off; p < L.line_len && (int)tabs.size() < n_cols; ++p) {
        const uint8_t c = t[p];
        if (c == '\t') tabs.push_back(p);
        else if (c < 0x21) return false;
    }
    if ((int)tabs.size() < n_cols - 1) return false;                         // fewer columns than the #CHROM line names
    for (int c = 0; c < n_cols; ++c) {
        const uint32_t a = c ? tabs[(size_t)c - 1] + 1 : L.cells_off;
        const uint32_t b = c < (int)tabs.size() ? tabs[(size_t)c] : L.line_len;
        if (b <= a) return false;
    }
    return true;
}

}  // namespace

// 0: *out_len bytes of rows in out (n_rows of them); 1: line *host_line needs the host parser; 2: out is too small; -1: bad arguments
extern "C" int pgv_emul_block(const uint8_t *text, int64_t len, int n_vcf_samples, int n_sel, const int32_t *sel_col,
                              const int32_t *sel_ploidy, int flags, double min_qual, int max_ref_len, const pg_vcf_filter *filters,
                              int n_filters, const char *contigs, int n_contig_bytes, int contig_mode, char missing, char sep,
                              int add_ref, const char *prev_chrom, int prev_chrom_len, const char *prev_pos, int prev_pos_len,
                              uint8_t *out, int64_t cap, int64_t *out_len, int64_t *n_rows, int64_t *host_line, int *taken) {
    PgvConfig cfg;
    PgvTables tab;
    const char *why = nullptr;
    const int ok = pgv_make_config(n_vcf_samples, n_sel, sel_col, sel_ploidy, flags, min_qual, max_ref_len, filters, n_filters,
                                   n_contig_bytes, contig_mode, missing, sep, add_ref, &cfg, &tab, &why);
    *taken = ok == 1;
    *out_len = *n_rows = 0;
    *host_line = -1;
    if (ok != 1) return ok < 0 ? -1 : 0;
    if (len == 0) return 0;
    if (text[len - 1] != '\n') { *host_line = 0; return 1; }
    std::vector<int64_t> start;
    for (int64_t p = 0; p < len;) {
        start.push_back(p);
        while (text[p] != '\n') ++p;
        ++p;
    }
    start.push_back(len);
    const int64_t n_lines = (int64_t)start.size() - 1;
    PgvKey key;                                                  // (what k_vcf_lastkey / pg_vcf_dev_set_prev leave on the device)
    memset(&key, 0, sizeof(key));
    key.chrom_len = PGV_KEY_NONE;
    if (prev_chrom && prev_pos) {
        if (prev_chrom_len > PGV_KEY_MAX || prev_pos_len > PGV_KEY_MAX) key.chrom_len = PGV_KEY_UNKNOWN;
        else {
            key.chrom_len = (uint32_t)prev_chrom_len;
            key.pos_len = (uint32_t)prev_pos_len;
            memcpy(key.chrom, prev_chrom, (size_t)prev_chrom_len);
            memcpy(key.pos, prev_pos, (size_t)prev_pos_len);
        }
    }
    std::vector<PgvLine> lines((size_t)n_lines);
    std::vector<uint32_t> tabs;
    int64_t at = 0, rows = 0;
    for (int64_t i = 0; i < n_lines; ++i) {
        const uint8_t *t = text + start[(size_t)i];
        const uint64_t n64 = (uint64_t)(start[(size_t)i + 1] - start[(size_t)i] - 1);
        if (n64 > 0xfffffff0ull) { *host_line = i; return 1; }
        PgvLine &L = lines[(size_t)i];
        if (pgv_head(t, (uint32_t)n64, cfg, reinterpret_cast<const uint8_t *>(contigs), &L) != PGV_OK) { *host_line = i; return 1; }
        if (!(L.flags & PGV_LINE_KEPT)) continue;
        if (cfg.flags & PGV_EXCLUDE_DUPLICATES) {
            const int dup = pgv_is_duplicate(t, L, key, [&](uint32_t j, const uint8_t **bt, uint32_t *bn) {
                if ((int64_t)j >= i) return false;
                const int64_t b = i - 1 - (int64_t)j;
                *bt = text + start[(size_t)b];
                *bn = (uint32_t)(start[(size_t)b + 1] - start[(size_t)b] - 1);
                return true;
            });
            if (dup == 2) { *host_line = i; return 1; }
            if (dup == 1) continue;
        }
        if (!line_tabs(t, L, n_vcf_samples, tabs)) { *host_line = i; return 1; }
        const bool cx = (L.flags & PGV_LINE_COMPLEX) != 0;
        // sizes first (the kernels know a row's place before they write it)
        std::vector<PgvCell> cells((size_t)n_sel);
        uint32_t bytes = L.fixed_len + (cx ? 0u : (uint32_t)cfg.plain_cells);
        for (int s = 0; s < n_sel; ++s) {
            const int col = tab.sel_col[(size_t)s];
            const uint32_t a = col ? tabs[(size_t)col - 1] + 1 : L.cells_off;
            const uint32_t b = col < (int)tabs.size() ? tabs[(size_t)col] : L.line_len;
            if (pgv_cell(t, a, b, L, cfg, tab.ploidy[(size_t)s], tab.fsel[(size_t)s], &cells[(size_t)s]) != PGV_OK) { *host_line = i; return 1; }
            if (cx) bytes += pgv_cell_bytes(L, cells[(size_t)s], tab.ploidy[(size_t)s]);
        }
        if (at + bytes > cap) return 2;
        uint8_t *o = out + at;
        for (uint32_t k = 0; k < L.chrom_len; ++k) *o++ = t[k];
        *o++ = (uint8_t)cfg.sep;
        for (uint32_t k = 0; k < L.pos_len; ++k) *o++ = t[L.pos_off + k];
        *o++ = (uint8_t)cfg.sep;
        if (cfg.add_ref) {
            for (uint32_t k = 0; k < L.al_len[0]; ++k) *o++ = t[L.al_off[0] + k];
            *o++ = (uint8_t)cfg.sep;
        }
        for (int s = 0; s < n_sel; ++s) {
            uint8_t *w = cx ? o : out + at + L.fixed_len + tab.cell_off[(size_t)s];
            o = pgv_cell_put(t, L, cfg, cells[(size_t)s], tab.ploidy[(size_t)s], cx, s + 1 == n_sel, w);
        }
        if (o != out + at + bytes) { fprintf(stderr, "pgv_emul_block: row %lld wrote %lld bytes, sized %u\n", (long long)i, (long long)(o - out - at), bytes); return -1; }
        at += bytes;
        ++rows;
    }
    *out_len = at;
    *n_rows = rows;
    return 0;
}
