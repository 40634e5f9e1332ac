// VCF lines on the device: the per-line and per-cell halves of the parseVCF.py drop-in as plain functions of one line / one sample
// column, written once and compiled twice -- by hipcc into k_vcf_heads / k_vcf_cells (pg_vcf_dev.hip: a thread per line, a lane per
// cell), and by g++ into tests/vcf_emul.cpp, which walks the same functions line by line on the host so that the CPU suite holds
// them against the host parser pg_encode_vcf + pg_vcf_render_rows (csrc/pg_vcf.cpp) on every golden and on random files.
//
// What they restate (the same reference lines pg_vcf.cpp cites):
//   VcfSite.__init__ / getSiteType / getGenotype      VCF_processing/parseVCF.py:49-191
//   the per-site filters of the main loop               VCF_processing/parseVCF.py:367-370
//   the output line                                     VCF_processing/parseVCF.py:380-383
// The device takes the REGULAR spelling of a VCF line only: columns separated by single tabs, POS as plain digits, QUAL and the
// filtered FORMAT values as plain decimals of up to 15 digits, at most PGV_MAX_ALLELES alleles, genotype allele tokens of up to
// three characters.  Anything else -- and everything the host parser answers with an error -- returns PGV_HOST: the block then goes
// through pg_encode_vcf, which knows every spelling and words the errors.  Nothing is guessed here.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PGV_HD __host__ __device__ inline
#else
#define PGV_HD inline
#endif

#define PGV_MAX_FILTERS 8
#define PGV_MAX_ALLELES 16
#define PGV_FLAG_LEN 16

#define PGV_OK 0
#define PGV_HOST 1          // this line needs the host parser

// option bits: those of pg_encode_vcf (include/popgen_hip.h)
#define PGV_SKIP_INDELS 1
#define PGV_KEEP_PARTIAL 2
#define PGV_MISMATCH_TO_MISSING 4
#define PGV_EXCLUDE_DUPLICATES 8

#define PGV_KEY_MAX 120
#define PGV_KEY_NONE 0xffffffffu         // no data line yet
#define PGV_KEY_UNKNOWN 0xfffffffeu      // a data line whose tokens the key does not hold: the host decides
// the CHROM and POS tokens of a data line (--excludeDuplicates compares a line with the data line before it, parseVCF.py:367)
struct PgvKey {
    uint32_t chrom_len, pos_len;
    uint8_t chrom[PGV_KEY_MAX], pos[PGV_KEY_MAX];
};

#define PGV_LINE_KEPT 1u
#define PGV_LINE_COMPLEX 2u      // some allele is not one base long: the row's cells are put together from the allele strings

struct PgvFilter {
    double min, max;
    int32_t site_types, gt_types;      // 0 = no selector
    int32_t flag_len;
    char flag[PGV_FLAG_LEN];
};

struct PgvConfig {
    int32_t n_vcf_samples, n_sel, flags, n_filters, max_ref_len, contig_mode, n_contig_bytes, add_ref;
    int32_t plain_cells;               // bytes of the cells (+ separators + line feed) of a row whose alleles are single bases
    char missing, sep;
    char pad[2];
    double min_qual;
    PgvFilter f[PGV_MAX_FILTERS];
};

// what k_vcf_heads leaves per line for k_vcf_cells (offsets are relative to the line's first byte)
struct PgvLine {
    uint32_t flags;
    uint32_t chrom_len, pos_off, pos_len, cells_off, fixed_len, line_len, pad0;
    uint8_t n_all, site_type;
    int8_t gt_idx, pad1;
    int8_t fidx[PGV_MAX_FILTERS];
    uint32_t al_off[PGV_MAX_ALLELES];
    uint16_t al_len[PGV_MAX_ALLELES];
    uint8_t al_chr[PGV_MAX_ALLELES];
};

struct PgvCell {
    uint8_t c0, c1, phase;
    int8_t a0, a1;
};

PGV_HD double pgv_p10(int k) {
    // (exact powers of ten; a switch, so that neither side needs a table in memory)
    switch (k) {
    case 0: return 1e0; case 1: return 1e1; case 2: return 1e2; case 3: return 1e3; case 4: return 1e4; case 5: return 1e5;
    case 6: return 1e6; case 7: return 1e7; case 8: return 1e8; case 9: return 1e9; case 10: return 1e10; case 11: return 1e11;
    case 12: return 1e12; case 13: return 1e13; case 14: return 1e14; default: return 1e15;
    }
}

// a plain decimal (digits, at most one '.', 1 .. 15 digits): 1 and *v, exactly what pg_vcf.cpp's py_float makes of it (the digits as
// an integer below 2^53 over an exact power of ten: one correctly rounded division); 0: no number in any spelling (no digit at all, or a
// second '.'); 2: a spelling the host decides (another character, more than 15 digits)
PGV_HD int pgv_float(const uint8_t *p, uint32_t n, double *v) {
    uint64_t m = 0;
    int digits = 0, frac = -1, dots = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint8_t c = p[k];
        if (c >= '0' && c <= '9') {
            if (digits < 16) m = m * 10 + (uint64_t)(c - '0');
            ++digits;
            if (frac >= 0) ++frac;
        } else if (c == '.') {
            if (frac < 0) frac = 0;
            ++dots;
        } else
            return 2;
    }
    if (digits == 0 || dots > 1) return 0;
    if (digits > 15) return 2;
    *v = frac > 0 ? (double)m / pgv_p10(frac) : (double)m;
    return 1;
}

PGV_HD bool pgv_eq(const uint8_t *a, const char *b, uint32_t n) {
    for (uint32_t k = 0; k < n; ++k)
        if (a[k] != (uint8_t)b[k]) return false;
    return true;
}

// the first two tokens of the line [t, t + n): 0 and their places (CHROM starts the line), 1 = not a data line (empty, or a '#' line:
// the reference skips those), 2 = a spelling the host reads
PGV_HD int pgv_line_key(const uint8_t *t, uint32_t n, uint32_t *chrom_len, uint32_t *pos_off, uint32_t *pos_len) {
    if (n == 0 || t[0] == '#') return 1;
    if (t[0] < 0x21) return 2;
    uint32_t p = 0;
    while (p < n && t[p] >= 0x21) ++p;
    if (p >= n || t[p] != '\t') return 2;
    *chrom_len = p;
    const uint32_t a = ++p;
    while (p < n && t[p] >= 0x21) ++p;
    if (p >= n || t[p] != '\t' || p == a) return 2;
    *pos_off = a;
    *pos_len = p - a;
    return 0;
}

// Is the kept line `t` (tokens from its PgvLine) a duplicate of the data line before it?  `before` walks back: before(j, &bt, &bn) gives
// line i - 1 - j (false: no more lines in this block -- then `key` decides, the last data line of the blocks before).
// 0 no, 1 yes, 2 the host decides.
template <class Before>
PGV_HD int pgv_is_duplicate(const uint8_t *t, const PgvLine &L, const PgvKey &key, Before before) {
    for (uint32_t j = 0;; ++j) {
        const uint8_t *bt;
        uint32_t bn;
        if (!before(j, &bt, &bn)) break;
        uint32_t cl, po, pl;
        const int r = pgv_line_key(bt, bn, &cl, &po, &pl);
        if (r == 1) continue;
        if (r == 2) return 2;
        if (cl != L.chrom_len || pl != L.pos_len) return 0;
        for (uint32_t k = 0; k < cl; ++k)
            if (bt[k] != t[k]) return 0;
        for (uint32_t k = 0; k < pl; ++k)
            if (bt[po + k] != t[L.pos_off + k]) return 0;
        return 1;
    }
    if (key.chrom_len == PGV_KEY_NONE) return 0;
    if (key.chrom_len == PGV_KEY_UNKNOWN) return 2;
    if (key.chrom_len != L.chrom_len || key.pos_len != L.pos_len) return 0;
    for (uint32_t k = 0; k < L.chrom_len; ++k)
        if (key.chrom[k] != t[k]) return 0;
    for (uint32_t k = 0; k < L.pos_len; ++k)
        if (key.pos[k] != t[L.pos_off + k]) return 0;
    return 1;
}

// One line [t, t + n) (no line feed).  PGV_OK: L->flags says whether the site is kept and whether its row is a complex one;
// PGV_HOST: see the head of the file.  contigs: the names of --include / --exclude separated by '\n'.
PGV_HD int pgv_head(const uint8_t *t, uint32_t n, const PgvConfig &cfg, const uint8_t *contigs, PgvLine *L) {
    L->flags = 0;
    L->line_len = n;
    if (n == 0 || t[0] == '#') return PGV_OK;                 // (skipped by the reference: parseVCF.py:228, 364)
    if (t[0] < 0x21) return PGV_HOST;                         // leading blanks: str.split() would skip them
    uint32_t ts[9], te[9];
    uint32_t p = 0;
    for (int k = 0; k < 9; ++k) {
        ts[k] = p;
        while (p < n && t[p] >= 0x21) ++p;
        te[k] = p;
        if (p == ts[k]) return PGV_HOST;                       // an empty token (two tabs in a row)
        if (p >= n) return PGV_HOST;                           // fewer than ten columns (an error or a line of another kind: the host words it)
        if (t[p] != '\t') return PGV_HOST;                     // another separator
        ++p;
    }
    if (p >= n) return PGV_HOST;
    // ---- the per-site filters (parseVCF.py:367-370) ----
    if (cfg.contig_mode) {
        bool listed = false;
        const uint32_t cn = te[0] - ts[0];
        uint32_t a = 0;
        const uint32_t ce = (uint32_t)cfg.n_contig_bytes;
        while (a <= ce && !listed) {
            uint32_t b = a;
            while (b < ce && contigs[b] != '\n') ++b;
            if (b - a == cn) {
                bool same = true;
                for (uint32_t k = 0; k < cn && same; ++k) same = contigs[a + k] == t[ts[0] + k];
                listed = same;
            }
            if (b >= ce) break;
            a = b + 1;
        }
        if (cfg.contig_mode == 1 && !listed) return PGV_OK;
        if (cfg.contig_mode == 2 && listed) return PGV_OK;
    }
    if (cfg.min_qual > 0) {
        double q = 0;
        const int r = pgv_float(t + ts[5], te[5] - ts[5], &q);
        if (r == 2) return PGV_HOST;
        if (r == 1 && q < cfg.min_qual) return PGV_OK;
    }
    const uint32_t ref_len = te[3] - ts[3];
    if (cfg.max_ref_len > 0 && ref_len > (uint32_t)cfg.max_ref_len) return PGV_OK;
    // ---- a kept site ----
    {
        const uint32_t pn = te[1] - ts[1];                     // POS as int() prints it again: plain digits, no leading zero
        if (pn > 18 || (pn > 1 && t[ts[1]] == '0')) return PGV_HOST;
        for (uint32_t k = 0; k < pn; ++k)
            if (t[ts[1] + k] < '0' || t[ts[1] + k] > '9') return PGV_HOST;
        L->pos_off = ts[1];
        L->pos_len = pn;
    }
    L->chrom_len = te[0];
    // alleles: REF + ALT.split(",") (ALT "." = none)
    int n_all = 1;
    if (ref_len > 65535) return PGV_HOST;
    L->al_off[0] = ts[3];
    L->al_len[0] = (uint16_t)ref_len;
    L->al_chr[0] = t[ts[3]];
    bool all_match = true, complex_row = ref_len != 1;
    if (!(te[4] - ts[4] == 1 && t[ts[4]] == '.')) {
        uint32_t a = ts[4];
        for (;;) {
            uint32_t b = a;
            while (b < te[4] && t[b] != ',') ++b;
            if (n_all >= PGV_MAX_ALLELES || b - a > 65535) return PGV_HOST;
            L->al_off[n_all] = a;
            L->al_len[n_all] = (uint16_t)(b - a);
            L->al_chr[n_all] = b > a ? t[a] : 0;
            all_match = all_match && b - a == ref_len;
            complex_row = complex_row || b - a != 1;
            ++n_all;
            if (b >= te[4]) break;
            a = b + 1;
        }
    }
    L->n_all = (uint8_t)n_all;
    L->site_type = (uint8_t)(n_all == 1 ? 1 : (all_match ? 2 : 4));         // MONO / SNP / INDEL
    // FORMAT: the LAST piece of each name (dict(zip()) keeps the last duplicate)
    int gt_idx = -1;
    int fi[PGV_MAX_FILTERS];
    for (int f = 0; f < PGV_MAX_FILTERS; ++f) fi[f] = -1;
    {
        uint32_t a = ts[8];
        int k = 0;
        for (;;) {
            uint32_t b = a;
            while (b < te[8] && t[b] != ':') ++b;
            if (b - a == 2 && t[a] == 'G' && t[a + 1] == 'T') gt_idx = k;
            for (int f = 0; f < cfg.n_filters; ++f)
                if ((int)(b - a) == cfg.f[f].flag_len && pgv_eq(t + a, cfg.f[f].flag, b - a)) fi[f] = k;
            ++k;
            if (b >= te[8]) break;
            a = b + 1;
            if (k > 120) return PGV_HOST;
        }
    }
    if (gt_idx < 0) return PGV_HOST;                           // (KeyError in the reference)
    L->gt_idx = (int8_t)gt_idx;
    for (int f = 0; f < PGV_MAX_FILTERS; ++f) L->fidx[f] = (int8_t)fi[f];
    L->cells_off = p;
    L->fixed_len = te[0] + 1 + L->pos_len + 1 + (cfg.add_ref ? ref_len + 1 : 0);
    L->flags = PGV_LINE_KEPT | (complex_row ? PGV_LINE_COMPLEX : 0u);
    return PGV_OK;
}

// One sample column [a, b) of a kept line (t = the line's first byte, offsets relative to it).  ploidy 1 or 2; fsel: bit f set when
// filter f applies to this sample (its `samples` selector).  The printed characters and the allele indices behind them, as
// pg_encode_vcf's walk() leaves them in chars_out / idx_out / phase_out.
PGV_HD int pgv_cell(const uint8_t *t, uint32_t a, uint32_t b, const PgvLine &L, const PgvConfig &cfg, int ploidy, uint32_t fsel, PgvCell *out) {
    if (b <= a) return PGV_HOST;                                 // an empty column cannot come out of str.split()
    int k = 0, host = 0;
    const int gt_idx = L.gt_idx;
    bool in_gt = gt_idx == 0;
    uint32_t fm = 0;
    for (int f = 0; f < cfg.n_filters; ++f)
        if (L.fidx[f] == 0) fm |= 1u << f;
    uint32_t reached = 0, bad = 0;
    // the number being read (a filtered field is a comma-separated list of them)
    uint64_t m = 0;
    int digits = 0, frac = -1, dots = 0;
    bool other = false;
    // the genotype
    int na = 1, cur = 0;
    bool phased = false;
    uint32_t raw0 = 0, raw1 = 0;
    int alen0 = 0, alen1 = 0;
    for (uint32_t p = a;; ++p) {
        const bool end = p >= b;
        const uint8_t c = end ? (uint8_t)':' : t[p];
        if (c < 0x21) return PGV_HOST;                           // a blank inside the columns: another way of splitting the line
        if (fm && (c == ':' || c == ',')) {
            // np.array(value.split(","), dtype=float): every piece must be a number inside [min, max]
            if (other || digits > 15) host = 1;
            else if (digits == 0 || dots > 1) bad |= fm;
            else {
                const double v = frac > 0 ? (double)m / pgv_p10(frac) : (double)m;
                for (int f = 0; f < cfg.n_filters; ++f)
                    if ((fm >> f & 1u) && !(cfg.f[f].min <= v && v <= cfg.f[f].max)) bad |= 1u << f;
            }
            m = 0; digits = 0; frac = -1; dots = 0; other = false;
        }
        if (c == ':') {
            reached |= fm;
            if (end) break;
            ++k;
            in_gt = k == gt_idx;
            fm = 0;
            for (int f = 0; f < cfg.n_filters; ++f)
                if (L.fidx[f] == k) fm |= 1u << f;
            continue;
        }
        if (in_gt) {
            // alleles = re.split("[/|]", GT); phase = "|" if "|" in GT else "/"
            if (c == '/' || c == '|') {
                phased = phased || c == '|';
                ++na;
                ++cur;
            } else if (cur == 0) {
                if (alen0 < 4) raw0 = (raw0 << 8) | c;
                ++alen0;
            } else if (cur == 1) {
                if (alen1 < 4) raw1 = (raw1 << 8) | c;
                ++alen1;
            }
        }
        if (fm && c != ',') {
            if (c >= '0' && c <= '9') {
                if (digits < 16) m = m * 10 + (uint64_t)(c - '0');
                ++digits;
                if (frac >= 0) ++frac;
            } else if (c == '.') {
                if (frac < 0) frac = 0;
                ++dots;
            } else
                other = true;
        }
    }
    if (k < gt_idx) return PGV_HOST;                             // a genotype without its GT piece (KeyError in the reference)
    if (alen0 > 3 || alen1 > 3) return PGV_HOST;
    // GTtype (parseVCF.py:13-18): only genotypes of the expected ploidy get as far as using it
    const bool distinct = na == 2 && !(alen0 == alen1 && raw0 == raw1);
    const bool has0 = (alen0 == 1 && raw0 == '0') || (na == 2 && alen1 == 1 && raw1 == '0');
    const bool hasdot = (alen0 == 1 && raw0 == '.') || (na == 2 && alen1 == 1 && raw1 == '.');
    const int gt_type = distinct ? 1 : (has0 ? 2 : (hasdot ? 4 : 8));
    bool passed = true;
    for (int f = 0; f < cfg.n_filters; ++f) {
        const PgvFilter &F = cfg.f[f];
        if (F.site_types && !(F.site_types & L.site_type)) continue;
        if (F.gt_types && !(F.gt_types & gt_type)) continue;
        if (!(fsel >> f & 1u)) continue;
        if (L.fidx[f] < 0 || !(reached >> f & 1u) || (bad >> f & 1u)) passed = false;
    }
    if (na != ploidy) {
        if (cfg.flags & PGV_MISMATCH_TO_MISSING) passed = false;
        else return PGV_HOST;                                    // (ValueError in the reference: the host words it)
    }
    if (host && passed) return PGV_HOST;                         // (a number of another spelling could only turn `passed` off)
    uint8_t o0 = (uint8_t)cfg.missing, o1 = (uint8_t)cfg.missing;
    int i0 = -1, i1 = -1;
    if (passed) {
        bool any_missing = false, bad_key = false;
        for (int i = 0; i < ploidy; ++i) {
            const uint32_t raw = i ? raw1 : raw0;
            const int al = i ? alen1 : alen0;
            // alleleDict[a]: a must be the decimal index of an allele, written as str(i) writes it
            int idx = -1;
            if (al >= 1 && !(al > 1 && (raw >> (8 * (al - 1)) & 0xff) == '0')) {
                idx = 0;
                for (int j = al - 1; j >= 0; --j) {
                    const uint32_t ch = raw >> (8 * j) & 0xff;
                    if (ch < '0' || ch > '9') { idx = -1; break; }
                    idx = idx * 10 + (int)(ch - '0');
                }
            }
            if (idx < 0 || idx >= (int)L.n_all) { bad_key = true; break; }
            if ((cfg.flags & PGV_SKIP_INDELS) && L.al_len[idx] != L.al_len[0]) { any_missing = true; continue; }
            if (i) i1 = idx; else i0 = idx;
            if (L.al_len[idx] == 1) {
                const uint8_t ch = L.al_chr[idx];
                if (i) o1 = ch; else o0 = ch;
                if (ch == (uint8_t)cfg.missing) any_missing = true;          // `missing not in sampleAlleles`
            }
        }
        if (bad_key || (any_missing && !(cfg.flags & PGV_KEEP_PARTIAL))) {
            o0 = o1 = (uint8_t)cfg.missing;
            i0 = i1 = -1;
        }
    }
    out->c0 = o0;
    out->c1 = o1;
    out->phase = phased ? (uint8_t)'|' : (uint8_t)'/';
    out->a0 = (int8_t)i0;
    out->a1 = (int8_t)(ploidy > 1 ? i1 : -1);
    return PGV_OK;
}

// bytes of a cell's text in a complex row (allele strings joined by the phase character, then the separator or line feed)
PGV_HD uint32_t pgv_cell_bytes(const PgvLine &L, const PgvCell &c, int ploidy) {
    uint32_t n = (uint32_t)ploidy;                               // ploidy - 1 phase characters + the separator
    n += c.a0 >= 0 ? L.al_len[c.a0] : 1u;
    if (ploidy > 1) n += c.a1 >= 0 ? L.al_len[c.a1] : 1u;
    return n;
}

// the cell's text at o (t = the line's first byte); `last`: the row's last cell ends with the line feed
PGV_HD uint8_t *pgv_cell_put(const uint8_t *t, const PgvLine &L, const PgvConfig &cfg, const PgvCell &c, int ploidy, bool complex_row, bool last,
                             uint8_t *o) {
    if (!complex_row) {
        *o++ = c.c0;
        if (ploidy > 1) { *o++ = c.phase; *o++ = c.c1; }
    } else {
        for (int i = 0; i < ploidy; ++i) {
            if (i) *o++ = c.phase;
            const int a = i ? c.a1 : c.a0;
            if (a >= 0) {
                const uint8_t *s = t + L.al_off[a];
                for (uint32_t k = 0; k < L.al_len[a]; ++k) *o++ = s[k];
            } else
                *o++ = (uint8_t)cfg.missing;
        }
    }
    *o++ = last ? (uint8_t)'\n' : (uint8_t)cfg.sep;
    return o;
}
