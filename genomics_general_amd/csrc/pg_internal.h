// Internal declarations shared by the HIP kernels (pg_kernels.hip) and the C-ABI host layer (pg_abi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/popgen_hip.h"

#define PG_MAX_POPS 16          // populations handled by the site-statistics kernels (K3/K6 take any number)
#define PG_SITES_PER_BLOCK 1024 // sites reduced by one block of the site-statistics kernels
#define PG_ABBA_SITES_PER_BLOCK 4096   // k_abba_q: fewer, longer blocks (prologue masks + epilogue reduction per block)
#define PG_ABBA_NSUM 6
#define PG_FOURPOP_NSUM 14
#define PG_XV_PLANES 2        // planes per word of virtual biallelic sites: x ("carries the tested allele"), v (called, not excluded)

// Input words (of 32 sites) per compaction group of the pack kernels = per block: 64 (2048 sites), or 128 when that still leaves
// the chip several times oversubscribed with blocks (pg_pick_group): larger groups end in fewer partial XV words (k_pairD's work)
// and amortise a block's start-up; measured on the north-star shape (50 000 -> 25 000 two-wave blocks): k_pack3 -5 %, k_pairD
// -3.5 %; on C2 (5000 -> 2600 one-wave blocks, fewer than the chip holds) k_pack3 +5 %.
#define PG_GROUP 64
#define PG_GROUP_MAX 128
// words of virtual sites reserved per group: worst case (every site with four alleles) / default
// Words reserved per group by default: enough whenever a window has no more virtual sites than sites (a biallelic site is
// one virtual site; + 1 for the group's partial last word).  k_pack2 / k_pack3 raise bit 1 of the flag word when a window needs
// more; the host then repeats the call with PG_XV_CAP (and keeps that reservation for the rest of the context's life).
#define PG_XV_CAP(grp) (3 * (grp))
#define PG_XV_CAP_DEFAULT(grp) ((grp) + 1)
// windows of up to this many sites get their float64 sums in NumPy's order (k_popdist_np, k_quartet_np): quotients and products of
// small integers sit on rounding ties of the printed digit, differences of equal means are +-0.0
#define PG_NP_MAX_SITES 256
#define PG_FLAG_MISMATCH 1      // some individual's two haplotypes differ in calledness: the diploid shortcut does not apply
#define PG_FLAG_XV_OVERFLOW 2   // a window produced more XV words than reserved

struct PgTask2 {                // one block of k_pairC (8 rows) / k_pairD (16 rows): circulant task, see pair_store_circ
    int32_t row0, nsub, col0, lower;   // rows row0.., columns col0, col0+1, ... (mod n), nsub = number of valid columns; lower = 2
};

struct PgSynthParams {
    uint64_t seed;
    int64_t first_site_index, scaf_len;
    int32_t n_dip, n_pops, var_thr, miss_thr;
};

// ---- launchers (all asynchronous on `st`) -----------------------------------------------------------
void pg_launch_synth(hipStream_t st, int8_t *gt, int S, int n_hap, int64_t site0, int64_t n_sites,
                     const int32_t *slot_gen_hap, PgSynthParams p);

void pg_launch_popdist_fin(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                           const int32_t *pop_start, int n_pops, int min_pair_sites, double *sum_out,
                           int64_t *cnt_out, int all_diploid, const int64_t *win_lo = nullptr, const int64_t *win_hi = nullptr,
                           long long skip_upto = -1);          // windows of up to skip_upto sites are left to k_popdist_np

// pi / dxy / Fst with the sums in NumPy's pairwise order (k_popdist_np + k_popstats_np); sums / cnts: [n_win][n_pops^2] scratch
void pg_launch_popdist_np(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                          const int32_t *pop_start, int n_pops, const int32_t *ref_row, const int32_t *pop_rank,
                          const int32_t *task_tree, const int32_t *trees, int max_leaves, int max_side, int min_pair_sites,
                          double min_data, int do_pairs, double *sums, int64_t *cnts, double *out, const int64_t *win_lo,
                          const int64_t *win_hi, long long max_sites);

void pg_launch_indpair_fin(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                           const int32_t *samp_start, const int32_t *samp_rank, int n_samp, int min_pair_sites, double *sum_out,
                           int64_t *cnt_out, int mean_mode);


void pg_launch_abba(hipStream_t st, const int8_t *gt, int S, const int64_t *win_lo, const int64_t *win_hi,
                    int n_win, int max_chunks, const int32_t *pop_start, int p1, int p2, int p3, int p4,
                    double min_data, int sel, int nsum, double *part_sums, int64_t *part_used, double *sums_out,
                    int64_t *used_out, uint32_t *flags, int64_t base, long long max_sites);

void pg_launch_popfreq(hipStream_t st, const int8_t *gt, int S, int n_hap, const int64_t *win_lo,
                       const int64_t *win_hi, int n_win, int max_chunks, const int32_t *pop_start, int n_pops,
                       unsigned long long *l_out, unsigned long long *S_out, unsigned long long *pairsum_out,
                       uint32_t *flags, int64_t base);
void pg_launch_popfreq_ordered(hipStream_t st, const int8_t *gt, int S, const int64_t *win_lo, const int64_t *win_hi, int n_win,
                               const int32_t *pop_start, int n_pops, const uint32_t *flags, int64_t base, double *theta_out);

void pg_launch_site_counts(hipStream_t st, const int8_t *gt, int S, int64_t site_lo, int64_t site_hi,
                           const int32_t *pop_start, int n_pops, int32_t *cnt_out);
void pg_launch_site_target(hipStream_t st, const int32_t *cnt, int64_t n_sites, int n_pops, int target, double min_data, int as_counts,
                           int has_threshold, double threshold, double *f_out, long long *i_out, uint8_t *keep_out);

void pg_launch_hap_called(hipStream_t st, const int8_t *gt, int S, int n_hap, const int64_t *win_lo,
                          const int64_t *win_hi, int n_win, int max_chunks, unsigned long long *out);

// ---- pairwise pipeline (pg_pair2.hip) ---------------------------------------------------------------
void pg_launch_pack2(hipStream_t st, const int8_t *gt, int S, const int64_t *win_lo, const int64_t *win_hi,
                     const int64_t *goff, const int64_t *vgoff, int n_win, int max_groups, int64_t total_groups, uint32_t *Vp,
                     int NPv, uint32_t *XV, int NP, int32_t *nw, int dip, int32_t *mismatch, uint32_t *pres, int capg, int grp);
bool pg_pack_needs_presence(int NP);
void pg_launch_expand(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                      int32_t *Cfull, int32_t *Dfull);
void pg_launch_pairC(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, const PgTask2 *tasks, int n_tasks,
                     int NPv, int n_units, int diag, int64_t avg_wq, int32_t *Cmat);
void pg_launch_pairD(hipStream_t st, const uint32_t *XY, const int32_t *nw, const int64_t *goff, int n_win,
                     const PgTask2 *tasks, int n_tasks, int NP, int N, int64_t avg_groups, int32_t *Dmat, int capg);
// the same counts on the matrix cores (pg_pair_mfma.hip): exact MX fp4 products of the bit planes, expanded in registers, one wave per block
void pg_launch_pairC_mfma(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, int NPv, int n_units, int diag,
                          int64_t avg_wq, int64_t max_sites, int32_t *Cmat);
void pg_launch_pairD_mfma(hipStream_t st, const uint32_t *XV, const int32_t *nw, const int64_t *goff, int n_win, int NP, int N,
                          int64_t avg_words, int64_t max_vsites, int32_t *Dmat, int capg);

// the called-count products with the planes staged through LDS, one block per window part (pg_pair_tile.hip); 0 = launched
bool pg_pair_tile_fits(int NPv);
int pg_launch_pairC_tile(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, int NPv, int n_units, int diag,
                         int64_t avg_wq, int64_t max_sites, int32_t *Cmat);

// called counts with one wave per SIMD and up to 14 tiles per wave (pg_pair_big.hip): planes of up to 224 units
bool pg_pair_big_fits(int NPv, int n_units);
void pg_launch_pairC_big(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, int NPv, int n_units, int diag,
                         int64_t avg_wq, int64_t max_sites, int32_t *Cmat);

void pg_launch_sample_het(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                          const int32_t *samp_start, int n_samp, int min_pair_sites, double *out);
void pg_launch_hapstats(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                        const int32_t *pop_start, int n_pops, int max_pop, const int32_t *order, int min_pair_sites, int diag_nan,
                        double max_dist, uint32_t *bits, size_t bits_per_window, double *out);
void pg_launch_unpack(hipStream_t st, const uint8_t *cells, int n_cols, int64_t n_rows, const int32_t *slot_src, int n_hap,
                      int8_t *gt, int S);
void pg_launch_flag_export(hipStream_t st, int32_t *flag, double *dst);
void pg_launch_popstats(hipStream_t st, const double *sums, const int64_t *cnts, int n_win, const int32_t *pop_start,
                        int n_pops, double min_data, int do_pairs, double *out, const int64_t *win_lo = nullptr,
                        const int64_t *win_hi = nullptr, long long skip_upto = -1);
