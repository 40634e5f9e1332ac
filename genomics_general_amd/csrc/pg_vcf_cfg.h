// The option set of a parseVCF run as the device path holds it (PgvConfig + per-sample tables), built from pg_encode_vcf's
// arguments.  Host code only; shared by pg_vcf_dev.hip and tests/vcf_emul.cpp so that the emulation the CPU suite runs is
// configured exactly as the kernels are.
#pragma once
#include "../../include/popgen_hip.h"
#include "pg_vcf_core.h"

#include <cstring>
#include <vector>

struct PgvTables {
    std::vector<int32_t> sel_col;
    std::vector<uint8_t> ploidy, fsel;
    std::vector<uint32_t> cell_off;          // where the cell of selected sample s starts among the cells of a plain row
    int max_col = 0;                         // the last VCF sample column a selected sample lives in
};

#define PGV_MAX_VCF_SAMPLES 14000            // (tab positions of a line: 4 bytes per sample column in LDS, one wavefront per block)

// 1: the device path takes this option set; 0: it does not (why, for a PG_TIMING line); < 0: bad arguments
inline int pgv_make_config(int n_vcf_samples, int n_sel, const int32_t *sel_col, const int32_t *sel_ploidy, int flags, double min_qual,
                           int max_ref_len, const pg_vcf_filter *filters, int n_filters, int n_contig_bytes, int contig_mode,
                           char missing, char sep, int add_ref, PgvConfig *cfg, PgvTables *tab, const char **why) {
    static const char *none = "";
    if (why) *why = none;
    if (n_sel < 0 || n_vcf_samples < 0 || (n_sel > 0 && (!sel_col || !sel_ploidy)) || n_filters < 0 || (n_filters > 0 && !filters)) return -1;
    auto no = [&](const char *w) { if (why) *why = w; return 0; };
    if (n_filters > PGV_MAX_FILTERS) return no("more than eight genotype filters");
    if (n_vcf_samples < 1 || n_vcf_samples > PGV_MAX_VCF_SAMPLES) return no("no sample columns, or more than the device keeps tab positions for");
    if (n_sel < 1) return no("no selected sample");
    if (n_contig_bytes > (1 << 16)) return no("a contig list of more than 64 KiB");
    memset(cfg, 0, sizeof(*cfg));
    cfg->n_vcf_samples = n_vcf_samples;
    cfg->n_sel = n_sel;
    cfg->flags = flags & (PG_VCF_SKIP_INDELS | PG_VCF_KEEP_PARTIAL | PG_VCF_MISMATCH_TO_MISSING | PG_VCF_EXCLUDE_DUPLICATES);
    cfg->n_filters = n_filters;
    cfg->max_ref_len = max_ref_len;
    cfg->contig_mode = contig_mode;
    cfg->n_contig_bytes = n_contig_bytes;
    cfg->add_ref = add_ref ? 1 : 0;
    cfg->missing = missing ? missing : 'N';
    cfg->sep = sep;
    cfg->min_qual = min_qual;
    tab->sel_col.assign(sel_col, sel_col + n_sel);
    tab->ploidy.resize((size_t)n_sel);
    tab->fsel.assign((size_t)n_sel, 0);
    tab->cell_off.resize((size_t)n_sel);
    tab->max_col = 0;
    uint32_t at = 0;
    for (int s = 0; s < n_sel; ++s) {
        if (sel_col[s] < 0 || sel_col[s] >= n_vcf_samples || sel_ploidy[s] < 1 || sel_ploidy[s] > 2) return -1;
        tab->ploidy[(size_t)s] = (uint8_t)sel_ploidy[s];
        tab->cell_off[(size_t)s] = at;
        at += sel_ploidy[s] == 2 ? 4u : 2u;
        if (sel_col[s] > tab->max_col) tab->max_col = sel_col[s];
    }
    cfg->plain_cells = (int32_t)at;
    for (int f = 0; f < n_filters; ++f) {
        if (!filters[f].flag) return -1;
        const size_t n = strlen(filters[f].flag);
        if (n < 1 || n >= PGV_FLAG_LEN) return no("a genotype filter on a FORMAT name of sixteen characters or more");
        PgvFilter &F = cfg->f[f];
        F.min = filters[f].min;
        F.max = filters[f].max;
        F.site_types = filters[f].site_types;
        F.gt_types = filters[f].gt_types;
        F.flag_len = (int32_t)n;
        memcpy(F.flag, filters[f].flag, n);
        for (int s = 0; s < n_sel; ++s)
            if (!filters[f].samples || filters[f].samples[s]) tab->fsel[(size_t)s] |= (uint8_t)(1u << f);
    }
    return 1;
}
