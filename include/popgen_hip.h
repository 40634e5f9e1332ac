/* popgen_hip.h -- C ABI of libpopgen_hip.so, the MI355X (gfx950) engine behind the per-window
 * statistics path of simonhmartin/genomics_general.
 *
 * The reference has no FFI: its CLI drivers call genomics.py functions directly.  Each entry point
 * below names the reference call site(s) it replaces (paths relative to the reference root); the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and lives in
 * genomics_general_amd/_lib.py.
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions across the boundary;
 *   - every function returns 0 on success, a negative pg_status on failure; the message for the last
 *     failure on the calling thread is pg_last_error();
 *   - the caller owns every host array passed in or out; the library owns all device memory of a ctx
 *     and frees it in pg_ctx_destroy;
 *   - a ctx is bound to one device and is not thread-safe; distinct ctxs may be used from distinct threads;
 *   - allele codes are one-hot int8: A=1 C=2 G=4 T=8, 0 = missing (anything that is not ACGT);
 *   - genotype blocks are site-major: gt[site][hap], haplotypes in *device slot order*, which the host
 *     chooses (populations contiguous, haplotypes of one individual adjacent);
 *   - a window is a half-open range [lo, hi) of site indices in the ctx's resident site buffer.
 */
#ifndef POPGEN_HIP_H
#define POPGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 1

typedef struct pg_ctx pg_ctx;

enum pg_status {
    PG_OK = 0,
    PG_ERR_ARG = -1,       /* bad argument / precondition */
    PG_ERR_HIP = -2,       /* HIP runtime error (message carries hipGetErrorString) */
    PG_ERR_NODEV = -3,     /* no usable GPU */
    PG_ERR_PARSE = -4,     /* malformed .geno text */
    PG_ERR_RCCL = -5,      /* RCCL error */
    PG_ERR_STATE = -6      /* call made in the wrong state (e.g. no samples set) */
};

/* genotype text formats, reference `-f/--genoFormat` (popgenWindows.py:205, genomics.py:390-396) */
enum pg_geno_format { PG_FMT_PHASED = 0, PG_FMT_PAIRS = 1, PG_FMT_HAPLO = 2, PG_FMT_DIPLO = 3 };
/* OR'ed into pg_encode_text's fmt: a cell may hold FEWER alleles than its column's ploidy (the remaining slots stay missing) --
 * rows tokenised once under the widest layout of a file whose ploidy changes along it (--inferPloidy, genomics.py:1108-1111). */
#define PG_FMT_NARROW_OK 0x100

/* kernels whose HIP-event timings pg_kernel_time reports */
enum pg_kernel_id { PG_K_PACK = 0, PG_K_PAIRWISE = 1 /* the called-count kernel (k_pairC*) */, PG_K_POPDIST_FIN = 2,
                    PG_K_SITESTATS = 3, PG_K_SYNTH = 4, PG_K_PAIRD = 5 /* the difference-count kernel (k_pairD*) */,
                    PG_K_INDPAIR_FIN = 6, PG_K_RESULT_D2H = 7 /* copy of a large result table back to the host (indPair means) */,
                    PG_K_ORDERED = 8 /* k_popfreq_ordered: thetaPi as the reference's site-by-site sum */, PG_K_COUNT_ = 9 };

int pg_abi_version(void);
const char *pg_last_error(void);

/* ---- device / context -------------------------------------------------------------------------- */
int pg_device_count(int *n_out);
int pg_ctx_create(pg_ctx **out, int device);
int pg_ctx_destroy(pg_ctx *ctx);
int pg_sync(pg_ctx *ctx);

/* Haplotype -> population / individual maps in device slot order.
 * Replaces the groups/sampleNames arrays of the Alignment built by genomics.genoToAlignment
 * (genomics.py:1101-1127) and SampleData.getPop (genomics.py:1282-1286).
 * hap_pop[h] in [0,n_pops) or -1 (no population); slots of one population must be contiguous and
 * populations must appear in increasing id order, -1 slots last.  hap_sample[h] = individual index in
 * [0,n_samples); slots of one individual must be contiguous. */
int pg_set_samples(pg_ctx *ctx, int n_hap, const int32_t *hap_pop, const int32_t *hap_sample, int n_pops);

/* The order the reference's float64 sums run in (optional; identity after pg_set_samples).  pop_row_order[slots in populations]:
 * for each population the slots of its haplotypes in the order of the Alignment's rows (haplotype names sorted,
 * genomics.py:1122), concatenated in population order; pop_name_rank[n_pops]: the position of each population among np.unique's
 * sorted labels (genomics.py:965).  pg_popdist_stats then forms every np.nanmean of genomics.py:976-992 over the flattened block in
 * NumPy's pairwise-summation order -- pi / dxy / Fst equal the reference's to the last bit (populations of up to a few hundred
 * haplotypes; beyond, upper-triangle sums in a fixed tree: equal within 1e-15). */
int pg_set_reference_order(pg_ctx *ctx, const int32_t *pop_row_order, const int32_t *pop_name_rank);

/* rank[n_samples]: of an individual pair, the one with the smaller rank supplies the ROWS of the haplotype block whose np.nanmean
 * pg_indpairdist_mean forms (genomics.py:946-947: `out[s][t]`, rows = the haplotypes of s; a 2 x 2 block added up row by row).
 * popgenWindows.py reads `pairDistDict[i][j]` with i before j in sorted names (popgenWindows.py:55-57), distMat.py in the order of
 * its samples (distMat.py:44-45).  Slot order after pg_set_samples. */
int pg_set_sample_rank(pg_ctx *ctx, const int32_t *rank);

/* Which windows get their float64 sums in NumPy's order (pg_popdist_stats, pg_abbababa, pg_fourpop): 0 = those of up to 256 sites (PG_NP_MAX_SITES)
 * (default: where the last bit shows in the printed digits), 1 = all (a caller that found a value of a long window within reach of
 * a rounding tie asks again for that window: cli.py), 2 = none (fixed reduction trees, within 1e-15). */
int pg_set_sum_order(pg_ctx *ctx, int mode);

/* The pairwise-summation tree of n values as the NumPy-order kernels walk it (no device needed): out[0 .. *len) =
 * { L, n_inner, n_levels, run_offset[L + 1], node_left[n_inner], node_right[n_inner], level_start[n_levels + 1] }; runs of at most
 * 128 values (slots 0 .. L - 1), inner node k (slot L + k) = slot node_left[k] + slot node_right[k], the inner nodes ordered by
 * height, the root last.  *len is set even when cap is too small. */
int pg_np_tree(int n, int32_t *out, int64_t cap, int64_t *len);

/* ---- resident site buffer ------------------------------------------------------------------------ */
int pg_reserve_sites(pg_ctx *ctx, int64_t n_sites);
/* pg_reserve_sites for a large reservation (>= 4 GiB) with a choice of physical placement: up to max_trials (<= 8) allocations are
 * held together, the regular pack + pair path is timed on each while it is still empty, the fastest is kept (the pack kernel's
 * time moves by +-6 % with the pages behind the rows; the probe predicts it).  probe_ms_out[max_trials] (may be NULL) receives the
 * probe times, *n_trials_out the number of candidates, *chosen_out the index kept.  Smaller reservations, or when memory does not
 * hold two candidates beside the scratch budget: exactly pg_reserve_sites. */
int pg_reserve_sites_tuned(pg_ctx *ctx, int64_t n_sites, int max_trials, double *probe_ms_out, int *n_trials_out, int *chosen_out);
/* The same choice for the planes the pack kernel writes (with the rows where they are, its time still moves by up to 4 % with the
 * pages behind the planes): up to max_trials (<= 8) sets of planes are held together, the regular pack + pair path over windows of
 * window_sites (<= 0: 50 000) of rows [0, n_sites) of the reservation -- empty or filled, they are only read -- is timed on each, the
 * fastest set is kept, the others released.  Candidate 0 = the planes the context holds when called.  Outputs as above
 * (probe_ms_out[max_trials]).  A choice made on the rows as they will be read is worth more than one made on empty rows: call it
 * once a long-lived data set is loaded.  Results do not depend on it; a later pass that needs larger planes allocates new ones. */
int pg_tune_planes(pg_ctx *ctx, int64_t n_sites, int64_t window_sites, int max_trials, double *probe_ms_out, int *n_trials_out,
                   int *chosen_out);
/* Copy gt[n_sites][n_hap] (tightly packed rows) to sites [site_offset, site_offset+n_sites). */
int pg_upload_sites(pg_ctx *ctx, int64_t site_offset, const int8_t *gt, int64_t n_sites);
int pg_download_sites(pg_ctx *ctx, int64_t site_offset, int8_t *gt_out, int64_t n_sites);
/* ---- streamed ingestion: uploads that overlap the kernels of the previous input block ------------------------------------ */
/* Bytes per resident row (n_hap rounded up to 16; pad bytes zero).  A tokenizer / decoder that is told n_hap = this pitch
 * writes rows that pg_upload_sites_async copies in one piece. */
int pg_row_pitch(pg_ctx *ctx, int *pitch_out);
/* pg_upload_sites on the context's copy stream, returning at once: gt[n_sites] rows of row_pitch bytes (>= n_hap; == pg_row_pitch
 * for a single linear copy), which must stay valid -- page-locked (pg_host_alloc) for a real DMA -- until pg_upload_wait.  The
 * window calls of the context do not wait for it: upload block k+1 into rows the windows of block k do not touch (e.g. the
 * other half of the reserved rows), call pg_upload_wait before using them.  Replaces, together with pg_encode_text, the
 * reference's serial parse -> queue -> genoToAlignment hand-over (popgenWindows.py:386-403, 445-447, genomics.py:1101-1127). */
int pg_upload_sites_async(pg_ctx *ctx, int64_t site_offset, const int8_t *gt, int64_t n_sites, int64_t row_pitch);
/* The same for packed genotype cells (the `.pgeno` payload: cells[n_sites][n_cols], one byte per cell = first allele's one-hot
 * code | second allele's << 4): the cells cross PCIe as they are (0.5 byte per allele call of a diploid) and are expanded into
 * resident rows on the device (k_unpack; the host half of genomics.py:390-396, 74-77).  slot_src[n_hap]: 2 * column + allele
 * index (0/1) feeding each slot, -1 = slot unused. */
int pg_upload_packed_async(pg_ctx *ctx, int64_t site_offset, const uint8_t *cells, int64_t n_sites, int n_cols,
                           const int32_t *slot_src);
/* Block until every queued asynchronous upload of the context has landed. */
int pg_upload_wait(pg_ctx *ctx);

/* K0 on the device: `len` bytes of complete `.geno` data lines (no header; the last byte a line feed) are copied to the GPU
 * (page-locked staging, host threads) and tokenised there straight into resident rows row_offset .. (+ n lines): line feeds
 * located by a count / scan / write pass, one wavefront per line.  The same cell decoding as pg_encode_text (genomics.py:390-396,
 * 407-408, 74-77, 1884-1904), for the regular layout only: one separator character between cells, every column's cells of one
 * width (the widths are read off the block's first line; a wanted column's width must be what its ploidy says, so a file of mixed
 * ploidy -- narrower cells for its haploid samples -- is regular too), no comment or blank lines in the block.  *ok_out = 1 when the layout was regular and the rows
 * are valid; 0 otherwise (nothing may be assumed about the rows: tokenise the block with pg_encode_text and upload it).
 * Outputs: pos_out[row_capacity] the positions; the scaffold runs of the block as (first row, offset and length of the scaffold
 * token in text), sorted by row, run_capacity entries each (*ok_out = 0 when there are more).  Call pg_count_lines first to
 * reserve the rows.  Blocks on the copy stream. */
int pg_tokenize_text(pg_ctx *ctx, const char *text, int64_t len, int fmt, int n_cols, int max_ploidy, const int32_t *col_slot,
                     const int32_t *col_ploidy, int64_t row_offset, int64_t *pos_out, int64_t row_capacity, int64_t *run_row_out,
                     int64_t *run_off_out, int32_t *run_len_out, int64_t run_capacity, int64_t *n_rows_out, int64_t *n_runs_out,
                     int *ok_out);
/* The same for `len` bytes at offset file_offset of an open file (plain text on disk): the staging threads pread() the text from
 * the page cache straight into their page-locked buffers, so the block never has to be mapped, faulted in or walked by the host.
 * Offsets in run_off_out are relative to file_offset. */
int pg_tokenize_file(pg_ctx *ctx, int fd, int64_t file_offset, int64_t len, int fmt, int n_cols, int max_ploidy,
                     const int32_t *col_slot, const int32_t *col_ploidy, int64_t row_offset, int64_t *pos_out, int64_t row_capacity,
                     int64_t *run_row_out, int64_t *run_off_out, int32_t *run_len_out, int64_t run_capacity, int64_t *n_rows_out,
                     int64_t *n_runs_out, int *ok_out);
/* The device tokenizer in three steps, two blocks in flight (slot 0 / 1), for a caller that overlaps the kernels of one block with
 * the copies of the next:   parse(k) -> submit(k+1) -> collect(k).
 *   submit   text (memory when `text` is given, else `len` bytes at file_offset of fd) -> the slot's device buffer, line feeds
 *            counted behind the copies; *ok_out = 0: not the regular layout (see pg_tokenize_text), nothing submitted
 *   parse    line-feed positions, rows row_offset .. cleared, parse kernel: queued on the copy stream, not waited for; waits only
 *            for the line count (*n_rows_out); *ok_out = 0 when the rows do not fit row_capacity / the reservation
 *   collect  waits for the parse; outputs as pg_tokenize_text (offsets relative to the block's first byte) */
int pg_tokenize_submit(pg_ctx *ctx, int slot, const char *text, int fd, int64_t file_offset, int64_t len, int fmt, int n_cols,
                       int max_ploidy, const int32_t *col_slot, const int32_t *col_ploidy, int *ok_out);
int pg_tokenize_parse(pg_ctx *ctx, int slot, int64_t row_offset, int64_t row_capacity, int64_t run_capacity, int64_t *n_rows_out,
                      int *ok_out);
int pg_tokenize_collect(pg_ctx *ctx, int slot, int64_t *pos_out, int64_t pos_capacity, int64_t *run_row_out, int64_t *run_off_out,
                        int32_t *run_len_out, int64_t run_capacity, int64_t *n_rows_out, int64_t *n_runs_out, int *ok_out);
/* Packed cells (`.pgeno` with codec none: one byte per genotype) straight from the file, by the staging
 * threads of the device tokenizer: pg_stage_file brings `len` bytes at file_offset of fd to byte dst_offset of staging slot 0 / 1
 * on the device (`capacity` bytes are made sure of when dst_offset == 0; returns when the bytes have landed); pg_unpack_staged
 * queues the expansion of n_rows rows of n_cols cells at byte src_offset of the slot into resident rows row_offset .. (slot_src as
 * pg_upload_packed_async) on the copy stream; pg_stage_sync waits for that stream.  The caller alternates the slots so that the
 * expansion of one block runs while the next block crosses PCIe. */
int pg_stage_file(pg_ctx *ctx, int slot, int fd, int64_t file_offset, int64_t len, int64_t dst_offset, int64_t capacity);
int pg_unpack_staged(pg_ctx *ctx, int slot, int64_t src_offset, int64_t n_rows, int n_cols, const int32_t *slot_src, int64_t row_offset);
int pg_stage_sync(pg_ctx *ctx);
/* What the device tokenizer of this context has spent so far: wall seconds of the host -> device copies of the text (staging
 * threads start to join: the PCIe-bound part), wall seconds of everything behind them (line feeds, parse kernel, positions and
 * runs back), and the bytes of text tokenised. */
int pg_tokenize_stats(pg_ctx *ctx, double *stage_seconds_out, double *kernel_seconds_out, int64_t *bytes_out);
/* Copy n resident rows from src_row to dst_row (ranges may overlap): the rows carried over to the next block of a stream. */
int pg_move_rows(pg_ctx *ctx, int64_t src_row, int64_t dst_row, int64_t n);

/* Fill sites on device with the counter-based synthetic generator (genomics_general_amd/synth.py is
 * the specification).  Dense layout: site i is scaffold i / scaf_len, position i % scaf_len + 1.
 * slot_gen_hap[h] = generator haplotype index held by device slot h. */
int pg_synth_fill(pg_ctx *ctx, int64_t site_offset, int64_t n_sites, int64_t first_site_index, uint64_t seed,
                  int64_t scaf_len, int32_t n_dip, int32_t n_pops_gen, const int32_t *slot_gen_hap,
                  int32_t var_thr, int32_t miss_thr);

/* ---- K0: host tokenizer (no GPU needed) -------------------------------------------------------- */
/* Replaces GenoFileReader.nextSite / parseGenoLine (genomics.py:1940-1945, 1884-1904) + splitSeq /
 * forceHomo / seqArrayToNumArray (genomics.py:390-396, 407-408, 74-77) for a buffer of complete lines.
 *   buf,len      text (no header line); lines starting with '#' are skipped, an empty line ends input
 *   n_cols       number of genotype columns in the file
 *   col_slot     [n_cols][max_ploidy] device slot of each allele of each column, -1 = column/allele unused
 *   col_ploidy   [n_cols] expected ploidy of the column's cells (0 = column not wanted)
 *   gt_out       [cap_sites][n_hap] one-hot codes (rows are fully written, unused slots = 0)
 *   pos_out      [cap_sites]
 *   scaf_off/len [cap_sites] byte offset/length of the scaffold token of each row inside buf
 *   n_sites_out  rows produced.  n_threads <= 0 -> hardware concurrency.
 * Error PG_ERR_PARSE when a wanted cell does not have the width its ploidy/format implies (the
 * reference asserts "Sample ploidy doesn't match number of sequences", genomics.py:1111). */
int pg_encode_text(const char *buf, size_t len, int fmt, int n_cols, int max_ploidy, const int32_t *col_slot,
                   const int32_t *col_ploidy, int n_hap, int8_t *gt_out, int64_t *pos_out, int64_t *scaf_off,
                   int32_t *scaf_len, int64_t cap_sites, int64_t *n_sites_out, int n_threads);
/* --inferPloidy: the reference takes the ploidy of a sample in a window from the cells the window holds (genoToAlignment with
 * ploidy None, genomics.py:1108-1111; splitSeq zips them, 390-396), so the cell widths of the whole input decide.  This walks a
 * buffer of whole lines and reports the data rows at which the width of a watched column's cell changes: change_row_out[k] (index
 * among the buffer's data rows), change_width_out[k][n_cols] (the widths from that row on; unwatched columns 0).  state[n_cols]
 * carries the last row's widths from buffer to buffer (start with -1 everywhere).  *n_changes_out is the true count; when it
 * exceeds cap nothing is lost -- call again with more room (state is advanced only when all changes fitted). */
int pg_text_cell_widths(const char *buf, size_t len, int n_cols, const int32_t *col_watch, int32_t *state,
                        int64_t *change_row_out, int32_t *change_width_out, int64_t cap, int64_t *n_changes_out,
                        int64_t *n_rows_out);
/* Row indices at which the scaffold token changes (contiguous scaffold runs, the unit slidingCoordWindows
 * restarts its window on, genomics.py:2013-2017).  n_runs_out is always the true count. */
int pg_scaffold_runs(const char *buf, const int64_t *scaf_off, const int32_t *scaf_len, int64_t n_sites,
                     int64_t *run_start_out, int64_t max_runs, int64_t *n_runs_out);
/* The same on RAW text (before any tokenising): byte offsets, relative to buf, of the first data line of every run of data lines
 * ('#' lines and empty lines skipped, genomics.py:1943) that share their first field.  n_out is always the true count (call
 * again with a larger cap).  Used to plan the sharding of `--windType predefined` input, whose forward-only reader
 * (genomics.py:2112-2171) makes the windows depend on the order of the runs in the whole file. */
int pg_text_runs(const char *buf, size_t len, int64_t *starts_out, int64_t cap, int64_t *n_out);
/* Window-range cuts of the multi-GPU input plan on RAW text.  A coordinate window is a function of (scaffold, position)
 * (genomics.py:1988-2017) and the reference hands windows, not scaffolds, to its workers (popgenWindows.py:396-403, 445-447), so a
 * rank's share of the input may start and end inside a scaffold run.  pg_text_seek_pos walks the data lines of buf (offset 0 = a
 * line start) while their first field equals scaf[0..scaf_len) and their position is < pos_min.  *off_out = offset of the line
 * it stopped at, *state_out = 1 (that line is of the run and has position *pos_out >= pos_min), 0 (that line belongs to another
 * scaffold: the run is over) or -1 (end of the buffer; with whole = 0 an unterminated last line is left for the next call),
 * *rows_out = data lines walked over.  pg_text_skip_rows: offset of the data line that follows n_rows data lines (row-index cuts
 * of sites windows, genomics.py:2032-2108); *rows_out < n_rows when the buffer ends first (*off_out = len). */
int pg_text_seek_pos(const char *buf, size_t len, int whole, const char *scaf, size_t scaf_len, int64_t pos_min, int64_t *off_out,
                     int32_t *state_out, int64_t *pos_out, int64_t *rows_out);
int pg_text_skip_rows(const char *buf, size_t len, int64_t n_rows, int64_t *off_out, int64_t *rows_out);
/* CPUs the process may really use: logical CPUs, cut by the affinity mask and the cgroup CPU quota.  What the library's host
 * thread pools are sized from when PG_HOST_THREADS is not set. */
int pg_usable_cpus(void);
/* Count data rows (non-'#', non-empty) in a text buffer so the caller can size the outputs. */
int pg_count_lines(const char *buf, size_t len, int64_t *n_rows_out);

/* ---- VCF -> genotype cells (host only): the native half of the parseVCF.py drop-in ------------------------------------------- */
/* Replaces, for a buffer of complete VCF lines (header lines may be included, they are skipped), VcfSite.__init__ / getSiteType /
 * getGenotype and the per-site filters of VCF_processing/parseVCF.py:49-191, 367-377.  One output row per kept site:
 *   chars_out[cap][2 * n_sel]   the allele characters the reference would print for selected sample s at [2s], [2s+1] (missing =
 *                               `missing`, second character 0 for a haploid sample); phase_out[cap][n_sel] '/' or '|'
 *   idx_out[cap][2 * n_sel]     the allele indices behind those characters (-1 = missing); row_flag_out[cap] = 1 when some printed
 *                               allele of the row is longer than one character (its character is `missing`; a text renderer takes
 *                               the strings from REF / ALT through idx_out)
 *   pos_out, chrom_off/len, ref_off/len, alt_off/len   POS and the locations of the CHROM, REF and ALT tokens in buf
 * sel_col[n_sel]: index of each selected sample among the VCF's sample columns; sel_ploidy: its expected ploidy (1 or 2).
 * flags: PG_VCF_*.  min_qual <= 0 / max_ref_len <= 0: no such filter.  contigs: names separated by '\n', contig_mode 0 none,
 * 1 include list, 2 exclude list.  prev_chrom / prev_pos: CHROM and POS tokens of the data line before the buffer (for
 * --excludeDuplicates across blocks; NULL at the start).  n_multibase_out: printed allele calls longer than one character.
 * cap_sites = 0: only count the kept sites. */
#define PG_VCF_SKIP_INDELS 1
#define PG_VCF_KEEP_PARTIAL 2
#define PG_VCF_MISMATCH_TO_MISSING 4
#define PG_VCF_EXCLUDE_DUPLICATES 8
typedef struct pg_vcf_filter {      /* --gtf flag=X min=X max=X siteTypes=.. gtTypes=.. samples=..  (parseVCF.py:255-266, 117-131) */
    const char *flag;               /* FORMAT field, e.g. "DP" */
    double min, max;
    int site_types;                 /* bit mask 1 MONO | 2 SNP | 4 INDEL, 0 = every site type */
    int gt_types;                   /* bit mask 1 Het | 2 HomRef | 4 Missing | 8 HomAlt, 0 = every genotype type */
    const uint8_t *samples;         /* per selected sample 0/1, NULL = every sample */
} pg_vcf_filter;
int pg_encode_vcf(const char *buf, size_t len, int n_vcf_samples, int n_sel, const int32_t *sel_col, const int32_t *sel_ploidy,
                  int flags, double min_qual, int max_ref_len, const pg_vcf_filter *filters, int n_filters, const char *contigs,
                  int n_contig_bytes, int contig_mode, char missing, const char *prev_chrom, int prev_chrom_len, const char *prev_pos,
                  int prev_pos_len, uint8_t *chars_out, int8_t *idx_out, uint8_t *phase_out, uint8_t *row_flag_out, int64_t *pos_out,
                  int64_t *chrom_off, int32_t *chrom_len, int64_t *ref_off, int32_t *ref_len, int64_t *alt_off, int32_t *alt_len,
                  int64_t cap_sites, int64_t *n_sites_out, int64_t *n_multibase_out, int n_threads);
/* The `.geno` text of rows [0, n_rows) of pg_encode_vcf's outputs, as the reference prints them (VCF_processing/parseVCF.py:151-169
 * the cells, 380-383 the line: CHROM, POS[, REF when add_ref], one cell per selected sample, joined by sep): a cell = the sample's
 * allele characters joined by its phase character; in a row whose row_flag is set the cells are put together from the REF / ALT
 * strings in buf through idx.  out == NULL: only *out_len_out (the bytes the rows take) is computed. */
int pg_vcf_render_rows(const char *buf, int64_t n_rows, int n_sel, const int32_t *sel_ploidy, const uint8_t *chars, const int8_t *idx,
                       const uint8_t *phase, const uint8_t *row_flag, const int64_t *pos, const int64_t *chrom_off,
                       const int32_t *chrom_len, const int64_t *ref_off, const int32_t *ref_len, const int64_t *alt_off,
                       const int32_t *alt_len, char sep, char missing, int add_ref, uint8_t *out, int64_t out_cap,
                       int64_t *out_len_out, int n_threads);

/* ---- VCF lines -> `.geno` rows on the device (csrc/pg_vcf_dev.hip): pg_encode_vcf + pg_vcf_render_rows in one, for the regular
 * spelling of a VCF line (single tabs, POS as plain digits, QUAL and the filtered FORMAT values as plain decimals of up to 15 digits,
 * at most 16 alleles).  Replaces the same reference lines (VCF_processing/parseVCF.py:49-191, 367-370, 380-383).  A bgzipped VCF never
 * crosses PCIe as text: k_inflate writes it into the tokenizer's text slot, only the rows come back.
 *   pg_vcf_dev_config        the option set: pg_encode_vcf's arguments + the rows' separator and --addRefTrack.  *taken_out = 0 (with
 *                            *why_out, a static string): an option set the device does not take (> 8 genotype filters, > 14 000 sample
 *                            columns, ...) -- the caller stays on pg_encode_vcf.
 *   pg_vcf_dev_submit        a block of whole lines -> text slot `slot` (0 / 1): from memory, or len bytes at file_offset of fd
 *   pg_vcf_dev_submit_bgzf   the same for a block of BGZF members (table: pg_bgzf_walk; head = text the caller holds in front of them,
 *                            text_len = head + members cut behind the block's last line feed; line_len_hint = bytes of a typical line)
 *   pg_vcf_dev_parse         queues k_vcf_heads / k_vcf_cells / k_vcf_scan on the block of `slot`
 *   pg_vcf_dev_collect       waits for them.  *host_line_out < 0: *out_len_out bytes of rows (*n_rows_out rows) are ready for
 *                            pg_vcf_dev_rows; else line *host_line_out of the block is one the device does not take (or the host
 *                            parser would answer with an error): the BLOCK goes to pg_encode_vcf (pg_vcf_dev_text brings its text back
 *                            when the host never had it)
 * The ingestion loop of VCF_processing/parseVCF.py's drop-in runs   parse(k) -> submit(k+1) -> collect(k) -> rows(k). */
int pg_vcf_dev_config(pg_ctx *ctx, int n_vcf_samples, int n_sel, const int32_t *sel_col, const int32_t *sel_ploidy, int flags,
                      double min_qual, int max_ref_len, const pg_vcf_filter *filters, int n_filters, const char *contigs,
                      int n_contig_bytes, int contig_mode, char missing, char sep, int add_ref, int *taken_out, const char **why_out);
/* --excludeDuplicates (parseVCF.py:367: a line against the DATA line before it): the device carries the CHROM / POS tokens of the last
 * data line from block to block itself (k_vcf_lastkey); pg_vcf_dev_set_prev hands it the key of blocks the HOST parsed before the next
 * submit (chrom == NULL: none), pg_vcf_dev_prev returns the key the block collected from `slot` started from (120 bytes each; a
 * length of -1: none; -2: the block before ended in a line the key could not hold and went to the host parser for it: the caller has that key) -- what pg_encode_vcf needs as prev_chrom / prev_pos when that block goes to it. */
int pg_vcf_dev_set_prev(pg_ctx *ctx, const char *chrom, int chrom_len, const char *pos, int pos_len);
int pg_vcf_dev_prev(pg_ctx *ctx, int slot, char *chrom_out, int *chrom_len_out, char *pos_out, int *pos_len_out);
int pg_vcf_dev_submit(pg_ctx *ctx, int slot, const char *text, int fd, int64_t file_offset, int64_t len);
int pg_vcf_dev_submit_bgzf(pg_ctx *ctx, int slot, const uint8_t *comp, int64_t comp_len, const uint32_t *in_off, const uint32_t *in_len,
                           const uint32_t *out_len, const uint32_t *crc, int64_t n_members, const char *head, int64_t head_len,
                           int64_t text_len, int64_t line_len_hint, int last_is_newline);
int pg_vcf_dev_parse(pg_ctx *ctx, int slot);
int pg_vcf_dev_collect(pg_ctx *ctx, int slot, int64_t *out_len_out, int64_t *n_rows_out, int64_t *host_line_out, int64_t *bgzf_len_out);
int pg_vcf_dev_rows(pg_ctx *ctx, int slot, uint8_t *dst, int64_t len);
/* `-o out.geno.gz` (`parseVCF.py ... | bgzip`, VCF_processing/README.md:33): with pg_vcf_dev_set_output(ctx, 1) the rows of a block are
 * also deflated where they lie (csrc/pg_deflate.hip: k_deflate, a wavefront per member of 65 280 bytes of rows; no end-of-file member);
 * pg_vcf_dev_collect reports the members' bytes in *bgzf_len_out (may be NULL), pg_vcf_dev_rows_bgzf copies them to dst. */
int pg_vcf_dev_set_output(pg_ctx *ctx, int bgzf_members);
int pg_vcf_dev_rows_bgzf(pg_ctx *ctx, int slot, uint8_t *dst, int64_t len);
int pg_vcf_dev_text(pg_ctx *ctx, int slot, uint8_t *dst, int64_t len);
int pg_vcf_dev_stats(pg_ctx *ctx, int64_t *blocks_out, int64_t *host_blocks_out);

/* freq.py's output rows (freq.py:98-113) formatted on all host threads: "scaffold\tposition\tcell\tcell...\n" per kept site.
 * mode 0: values = int32 [n][n_pops][4], cells "a,c,g,t"; mode 1: values = int64 [n][n_pops]; mode 2: values = double [n][n_pops]
 * printed as NumPy prints a double rounded to four decimals (nan, 0.0, 0.3333).  run_of_row[i] indexes the scaffold names
 * (names[name_off[r] .. name_off[r+1])); keep[i] == 0 drops row i (NULL keeps all).  *out_len = bytes needed; PG_ERR_ARG when that
 * exceeds out_cap. */
int pg_format_freq_rows(int mode, int64_t n_rows, int n_pops, const void *values, const int64_t *pos, const int32_t *run_of_row,
                        const char *names, const int64_t *name_off, const uint8_t *keep, char *out, int64_t out_cap, int64_t *out_len,
                        int n_threads);
/* Rows of float64 values as Python prints them: what makeDistMatString / PhylipString / NexusString (genomics.py:2288-2306) get from
 * `distArray.round(roundTo).astype(str)`.  Value j of row i = v[i * row_len + j], rounded like np.round(x, round_to) when
 * round_to >= 0 (rint(x * 10^r) / 10^r), printed as repr(float) (shortest digits that read back; fixed notation for
 * 1e-4 <= |x| < 1e16 with ".0" behind integers, else d.ddde-XX; nan / inf / -0.0), separated by sep, a line feed behind the row;
 * prefix (may be NULL): the bytes prefix[prefix_off[i] .. prefix_off[i+1]) in front of row i (taxon labels).  out == NULL: only
 * *out_len. */
int pg_format_float_rows(const double *v, int64_t n_rows, int64_t row_len, int round_to, char sep, const char *prefix,
                         const int64_t *prefix_off, char *out, int64_t out_cap, int64_t *out_len, int n_threads);

/* Inflate the independently deflated chunks of a `.pgeno` block (zlib streams; genoio.PackedWriter) on all host threads: the
 * concatenated output goes to dst_a (first len_a bytes: the block's int32 positions) and dst_b (the rest: its cells, e.g. rows of
 * the page-locked array an upload will read).  src_off / src_len locate chunk i in src, raw_len[i] is its inflated size. */
int pg_inflate_chunks(const uint8_t *src, const int64_t *src_off, const int64_t *src_len, const int64_t *raw_len, int n_chunks,
                      uint8_t *dst_a, int64_t len_a, uint8_t *dst_b, int64_t len_b, int n_threads);

/* ---- `.geno.gz` written by bgzip (BGZF: independent gzip members of at most 64 KiB of text) ----------------------------------------
 * Replaces gzip.open() in GenoFileReader.__init__ (genomics.py:1917-1919; popgenWindows.py:313 "-g input.geno.gz" is the
 * reference's normal input, its producer `parseVCF.py ... | bgzip`, VCF_processing/README.md:33).
 *
 * pg_bgzf_walk: the member table of buf[0 .. len) (RFC 1952 headers parsed; each member must carry the 'B' 'C' size subfield).  It
 * stops in front of a member that is not complete in buf, after max_members members, or once the walked members hold >= max_text
 * bytes of text (max_text <= 0: no limit).  Per member k: its deflate stream buf[in_off[k] .. + in_len[k]), ISIZE out_len[k] and
 * CRC-32 crc[k] of its trailer.  *consumed_out = bytes of buf the walked members occupy, *text_out = sum of their ISIZE. */
int pg_bgzf_walk(const uint8_t *buf, int64_t len, int64_t max_members, int64_t max_text, uint32_t *in_off, uint32_t *in_len,
                 uint32_t *out_len, uint32_t *crc, int64_t *n_members_out, int64_t *consumed_out, int64_t *text_out);
/* the members inflated by a pool of host threads (zlib): member k -> dst[out_off[k] .. + out_len[k]); crc (may be NULL) is checked.
 * The route of blocks the device tokenizer refuses and of the readers' own small reads (header line, shard cuts). */
/* ONE gzip stream (`gzip file.geno`: the reference reads it with gzip.open, popgenWindows.py:313, genomics.py:1917): nothing to run in
 * parallel, so zlib's inflate() on the reader's thread, straight into the caller's block buffer.  pg_gzip_read_lines: at least `want`
 * bytes of text (fewer only at the end of the input) extended to the next line feed, in dst[0 .. *got_out); cap > want by the
 * longest line expected; *complete_out = 0: the last line did not end inside cap (call again, append); *eof_out = 1: nothing left.
 * Concatenated members are followed; a damaged stream is PG_ERR_PARSE. */
typedef struct pg_gz pg_gz;
int pg_gzip_open(const char *path, pg_gz **out);
int pg_gzip_read_lines(pg_gz *g, uint8_t *dst, int64_t cap, int64_t want, int64_t *got_out, int *complete_out, int *eof_out);
int pg_gzip_close(pg_gz *g);
/* out4: [0] threads of the chunk-parallel decoder (0: the serial decoder, -1: it took over after a batch found no chain of chunks,
 * -2: zlib), [1] batches decoded side by side, [2] members that went on serially, [3] bytes of text handed out so far */
int pg_gzip_stats(pg_gz *g, int64_t *out4);
int pg_inflate_members(const uint8_t *comp, const uint32_t *in_off, const uint32_t *in_len, const int64_t *out_off,
                       const uint32_t *out_len, const uint32_t *crc, int64_t n_members, uint8_t *dst, int n_threads);
/* text -> BGZF, what `bgzip` writes (tools/bgzip.py; bench.py's compressed samples; tests): members of `block` bytes of text (bgzip:
 * 65280) deflated at `level` by a pool of host threads, + the empty EOF member when eof_marker != 0. */
int pg_bgzf_compress(const uint8_t *text, int64_t len, int level, int block, int eof_marker, uint8_t *out, int64_t out_cap,
                     int64_t *out_len_out, int n_threads);
/* the members inflated ON THE DEVICE (k_inflate: one wavefront per member, canonical Huffman decoding by ballot, matches copied 64
 * bytes at a time; the trailers' CRC-32 are checked when crc != NULL -- inside k_inflate as the text leaves, or by k_crc32 with PG_BGZF_CRC_FOLD=0): comp[0 .. comp_len) -> dst[0 .. sum out_len).  kernel_ms_out (may
 * be NULL): device time of the kernels.  PG_ERR_PARSE names the first damaged member. */
/* text[0 .. len) -> BGZF members (no end-of-file member) at out, deflated on the device by k_deflate -- the text crosses PCIe, so
 * this is the entry point of tests and tools (tools/bgzip.py --device); a drop-in whose text is made on the device uses
 * pg_vcf_dev_rows_bgzf.  *out_len_out: the members' bytes; kernel_ms_out (may be NULL): device time of the three kernels.
 * Replaces bgzip in `parseVCF.py ... | bgzip` (VCF_processing/README.md:33). */
int pg_bgzf_compress_device(pg_ctx *ctx, const uint8_t *text, int64_t len, uint8_t *out, int64_t out_cap, int64_t *out_len_out,
                            double *kernel_ms_out);
int pg_inflate_device(pg_ctx *ctx, const uint8_t *comp, int64_t comp_len, const uint32_t *in_off, const uint32_t *in_len,
                      const uint32_t *out_len, const uint32_t *crc, int64_t n_members, uint8_t *dst, double *kernel_ms_out);
/* The submit step of the device tokenizer (pg_tokenize_submit) for a block that is still deflated -- comp[0 .. comp_len), or with
 * comp == NULL the comp_len bytes at file_offset of fd (read by the staging threads with pread) --: the members cross PCIe as they
 * are (10 - 26 x fewer bytes than their text) and are inflated into the slot's text buffer behind `head` (head_len bytes of text the
 * caller already holds: the unfinished line the previous block ended with).  The block's text = head + the members' text, cut to
 * text_len bytes (it must end with a line feed; the caller keeps what follows for the next block).  first_line: the block's first
 * line without its line feed (the cell widths are read off it).  parse / collect as for plain text; a damaged member makes
 * pg_tokenize_parse fail with PG_ERR_PARSE.  pg_tokenize_run_names reads the scaffold names of the runs pg_tokenize_collect reported
 * (offsets / lengths in the block's text) back from the device, one after the other into out: the host never had that text. */
int pg_tokenize_submit_bgzf(pg_ctx *ctx, int slot, const uint8_t *comp, int fd, int64_t file_offset, int64_t comp_len,
                            const uint32_t *in_off, const uint32_t *in_len,
                            const uint32_t *out_len, const uint32_t *crc, int64_t n_members, const char *head, int64_t head_len,
                            int64_t text_len, const char *first_line, int64_t first_line_len, int fmt, int n_cols, int max_ploidy,
                            const int32_t *col_slot, const int32_t *col_ploidy, int *ok_out);
int pg_tokenize_run_names(pg_ctx *ctx, int slot, const int64_t *run_off, const int32_t *run_len, int64_t n_runs, char *out,
                          int64_t out_capacity);

/* Packed `.pgeno` input (genomics_general_amd/genoio.py: a tokenised `.geno` file kept on disk, one byte per genotype cell =
 * first allele code | second allele code << 4, in file column order): block of cells -> one-hot codes in slot order, same
 * col_slot / col_ploidy tables as pg_encode_text. */
int pg_decode_packed(const uint8_t *cells, int64_t n_rows, int n_cols, int max_ploidy, const int32_t *col_slot,
                     const int32_t *col_ploidy, int n_hap, int8_t *gt_out, int n_threads);

/* ---- K2: pairwise matrices ----------------------------------------------------------------------- */
/* Replaces Alignment.distMatrix / pairDist / numHamming (genomics.py:907-916, 903-905, 1219-1221) and
 * Alignment.pairNonNan (genomics.py:1042-1047).  For each window: D[i][j] = #sites where i and j are
 * both called and differ, C[i][j] = #sites both called; full symmetric [n_hap][n_hap] int32 matrices with
 * zero diagonal, in device slot order.  distMat[i][j] of the reference == D/C (nan when C == 0). */
int pg_pairwise(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int32_t *D_out, int32_t *C_out);

/* ---- K2+K3: population distance sums -------------------------------------------------------------- */
/* Replaces Alignment.groupDistStats (genomics.py:956-995).  For every window and every unordered
 * population pair (x<=y), in the order (0,0),(0,1),..,(0,P-1),(1,1),.. : the float64 sum of D/C over
 * haplotype pairs {i in x, j in y, i<j when x==y} with C >= max(min_pair_sites,1), and the number of such
 * pairs.  nanmean / minData / pi_s / pi_t / Fst are formed from these on the host (engine.py). */
int pg_popdist(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int min_pair_sites,
               double *sum_out, int64_t *cnt_out);

/* ---- K2+K3 with the statistics finished on the device -------------------------------------------- */
/* Replaces Alignment.groupDistStats (genomics.py:956-995) end to end: stats_out[n_win][P + (do_pairs ? P*(P-1) : 0)] =
 * pi of every population (CLI order), then (do_pairs) dxy of every pair x<y (x-major), then Fst of every pair; nan exactly
 * where the reference's nanmean_min / Fst give nan.  Same float64 operations, in the same order, as the reference. */
int pg_popdist_stats(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int min_pair_sites,
                     double min_data, int do_pairs, double *stats_out);

/* ---- K2+K6: individual-pair distance sums -------------------------------------------------------- */
/* Replaces Alignment.indPairDists (genomics.py:934-954).  For every window and unordered individual pair
 * (s<=t), (0,0),(0,1).. : sum of D/C over haplotype pairs {a in s, b in t, a<b when s==t} with
 * C >= max(min_pair_sites,1), and their number. */
int pg_indpairdist(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int min_pair_sites,
                   double *sum_out, int64_t *cnt_out);

/* The finished means instead of sums and counts (one array, half the bytes to copy back): d_out[n_win][pairs] =
 * nanmean of the haplotype-pair block of individuals (s,t) exactly as Alignment.indPairDists forms it (genomics.py:946-947):
 * s != t: sum/count (nan when no haplotype pair qualifies); s == t: the symmetric block holds every pair twice, and with
 * diag_counts_zeros != 0 (includeSameWithSame, genomics.py:940) one zero per haplotype counts as data. */
int pg_indpairdist_mean(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int min_pair_sites,
                        int diag_counts_zeros, double *d_out);
/* Deferred result tables, for loops that compute large tables back to back (bench.py's distMat shape: 40 MB per pass): with on != 0,
 * pg_indpairdist_mean returns when its kernels are done and lets the copy of the table into d_out -- which must then be page-locked
 * (pg_host_alloc), else the call behaves as before -- run on a stream of its own, beside the kernels of the following calls (two
 * device-side buffers alternate).  d_out is complete after pg_results_wait or pg_sync. */
int pg_set_deferred_results(pg_ctx *ctx, int on);
int pg_results_wait(pg_ctx *ctx);

/* The same means from pair counts the CALLER supplies: D, C as pg_pairwise writes them ([n_win][n_hap][n_hap] int32, device slot
 * order).  Pair counts are sums over sites, so the counts of disjoint parts of a window add: this is how `distMat.py --windType cat`
 * (one window = the whole input, distMat.py:284-289 / genomics.py:1949-1967) runs on several GPUs -- every rank counts its share
 * of the lines (pg_pairwise), the counts are summed across the ranks, every rank finishes the means from the sums. */
int pg_indpairdist_mean_from_counts(pg_ctx *ctx, const int32_t *D, const int32_t *C, int n_win, int min_pair_sites,
                                    int diag_counts_zeros, double *d_out);

/* ---- indHet / hapStats finished on the device (no N x N matrices leave the GPU) ------------------------------- */
/* Replaces Alignment.sampleHet (genomics.py:918-929) on the reference worker's cached distance matrix:
 * het_out[n_win][n_individuals] (slot order of the individuals) = D/C of a diploid individual's two haplotypes where bit 1 of
 * their jointly called site count is set (the reference's `len(x)==2 & C >= _minSites` precedence), nan otherwise and where
 * C < min_pair_sites (the mask a preceding groupDistStats(minSites) leaves on the cache, genomics.py:959-961; 0 = none). */
int pg_sample_het(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int min_pair_sites, double *het_out);
/* Replaces Alignment.H12stats (genomics.py:1079-1098) + distMat_to_cluster_sizes (genomics.py:1239-1261):
 * h_out[n_win][n_pops][3] = H1, H12, H2 of every population.  pop_row_order[slots in populations]: for each population the
 * slots of its haplotypes in the reference's row order (haplotype names sorted, genomics.py:1122) -- the greedy clustering
 * breaks ties by row order.  min_pair_sites as above; diag_nan != 0 when an earlier step of the reference's worker left a nan
 * diagonal on the cached matrix (groupDistStats, or indPairDists without includeSameWithSame). */
int pg_hapstats(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int min_pair_sites, int diag_nan,
                double max_dist, const int32_t *pop_row_order, double *h_out);

/* ---- K1+K4: ABBA-BABA window sums ------------------------------------------------------------------ */
/* Replaces genomics.ABBABABA(polarize=True) (genomics.py:1647-1695) with f4/D/fd/fdm/ABBA/BABA
 * (genomics.py:1409-1475, 1565-1569).  sums_out[n_win][6] = { sum f4(p1,p2,p3,p4), sum (ABBA+BABA),
 * sum f4(p1,pd,pd,p4), sum f4(pdm1,pdm2,pdm3,p4), sum ABBA, sum BABA }; sites_used_out[n_win]: the sites that entered the
 * sums, or -1 for a window without one good site (biallelic with enough data in all four populations): there the reference
 * answers sitesUsed = nan (its zip() of six names with seven values drops the 0, genomics.py:1693-1695). */
int pg_abbababa(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int p1, int p2, int p3,
                int p4, double min_data, double *sums_out, int64_t *sites_used_out);

/* ---- K1+K4': four-population window sums (fourPopWindows.py) ---------------------------------------- */
/* Replaces genomics.fourPop (genomics.py:1585-1643) and its per-site terms (genomics.py:1409-1563).
 * allele_sel chooses the allele whose frequencies p1..p4 enter the terms:
 *   PG_SEL_MINOR    default of the reference: np.argsort(all4freqs)[:,2] (genomics.py:1615)
 *   PG_SEL_POLARIZE --polarize: allele absent from P4 (genomics.py:1610)
 *   PG_SEL_FIXED    --fixed: as polarize and fixed (0 or 1) in P1, P2, P3 (genomics.py:1611-1614)
 * sums_out[n_win][14] = { sum f4, sum (ABBA+BABA), sum f4(p1,pd,pd,p4), sum f4(pdm1,pdm2,pdm3,p4), sum ABBA,
 * sum BABA, sum f4_c, sum f4_c(p1,pd,pd,p4), sum f4_c(pdm1,pdm2,pdm3,p4), sum fdh-denominator, sum
 * fdh2-denominator, sum fh-denominator, sum ABAA, sum BAAA }; sites_used_out[n_win]. */
enum pg_allele_sel { PG_SEL_MINOR = 0, PG_SEL_POLARIZE = 1, PG_SEL_FIXED = 2 };
int pg_fourpop(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int p1, int p2, int p3,
               int p4, double min_data, int allele_sel, double *sums_out, int64_t *sites_used_out);

/* ---- K1+K5: site-frequency window sums -------------------------------------------------------------- */
/* Replaces Alignment.groupFreqStats (genomics.py:1002-1028) + baseCountPi (609-616).  Sites used are those
 * with no missing call in ANY haplotype slot.  l_out[n_win]; S_out[n_win][n_pops] = #sites with >1 allele
 * in the population; pairsum_out[n_win][n_pops] = sum over sites of sum_{a<b} c_a c_b (exact integers).
 * theta_pi_out[n_win][n_pops] (may be NULL): thetaPi as the reference forms it -- Python's sum() over the per-site values
 * `pairs / (.5*N*(N-1))` in site order (genomics.py:1016-1018), a sequential float64 sum, reproduced bit for bit (Tajima's D of
 * a population of three haplotypes is that sum's rounding noise over a variance of zero). */
int pg_popfreq(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int64_t *l_out,
               int64_t *S_out, int64_t *pairsum_out, double *theta_pi_out);

/* ---- K1 raw: per-site per-population base counts ------------------------------------------------- */
/* Replaces Alignment.siteFreqs(asCounts=True) / binBaseFreqs (genomics.py:1049-1052, 592-599) for every
 * population at once: cnt_out[n_sites][n_pops][4] (A,C,G,T) for sites [site_lo, site_hi). */
int pg_site_counts(pg_ctx *ctx, int64_t site_lo, int64_t site_hi, int32_t *cnt_out);
/* freq.py's per-site finaliser on those counts (freq.py:60-105 with derivedAllele / minorAllele, genomics.py:636-669), for sites
 * [site_lo, site_hi): target 1 = derived (the last population is the outgroup), 2 = minor.  Per site and population the count of the
 * target allele (as_counts: values_out is int64[n][n_pops], 0 where the site or the population's data fail) or its frequency
 * np.around(c / n, 4) (values_out is float64[n][n_pops], NaN there; has_threshold: frequencies >= threshold become 1, the others 0);
 * a population counts when its called alleles number >= min_data (freq.py:80 compares the COUNT).  keep_out[n]: 0 for a row that
 * is all NaN / all zero (what freq.py drops without --keepNanLines). */
int pg_site_target(pg_ctx *ctx, int64_t site_lo, int64_t site_hi, int target, double min_data, int as_counts, int has_threshold,
                   double threshold, void *values_out, uint8_t *keep_out);

/* ---- per-haplotype called-site counts ------------------------------------------------------------ */
/* Replaces Alignment.seqNonNan (genomics.py:1038-1040) as used by distMat.py:40 (--minPerInd):
 * called_out[n_win][n_hap] = number of sites of the window at which the haplotype slot is called. */
int pg_hap_called(pg_ctx *ctx, const int64_t *win_lo, const int64_t *win_hi, int n_win, int64_t *called_out);

/* ---- measurement ------------------------------------------------------------------------------------ */
/* Accumulated HIP-event time (ms) and launch count of a kernel family since the last reset. */
int pg_kernel_time(pg_ctx *ctx, int kernel_id, double *ms_out, int64_t *launches_out);
/* Which kernel families are bracketed by events (bit k = family k, default all).  Event records between kernels cost a few
 * microseconds of GPU idle time each; a throughput run can restrict them to the family it reports. */
int pg_kernel_time_select(pg_ctx *ctx, uint32_t mask);
int pg_kernel_time_reset(pg_ctx *ctx);
/* Placement experiments (tools/pack_variance.py; not used by the drivers): device address / size of a big buffer (which: 0 resident
 * rows, 1 called plane, 2 XV planes), and a way to make its next allocation start lead_bytes behind what hipMalloc returns. */
int pg_debug_address(pg_ctx *ctx, int which, uint64_t *addr_out, uint64_t *bytes_out);
int pg_debug_place(pg_ctx *ctx, int which, uint64_t lead_bytes);
/* CU partition experiment (tools/cu_split_sweep.py; not used by the drivers): the context's compute stream on pair_cus_per_xcd
 * compute units of every XCD, the pack stream of the two-stream pipeline (PG_OVERLAP=1) on the others; 0 = plain streams. */
int pg_debug_cu_split(pg_ctx *ctx, int pair_cus_per_xcd);
/* Where the creation of the process's first context went, in seconds: [0] hipGetDeviceCount (the runtime's start-up), [1]
 * hipSetDevice + the first stream, [2] the other streams and events (tools/ctx_time.py). */
int pg_ctx_create_times(double *out3);
/* scratch budget (bytes) for per-batch bit-planes + matrices; default 48 GiB (a job that fits runs as one batch; a larger one is
 * cut into at least eight sub-batches that alternate between two slots of half the budget each) */
int pg_set_scratch_limit(pg_ctx *ctx, int64_t bytes);

/* ---- page-locked host memory ---------------------------------------------------------------------------- */
/* Output arrays handed to this library may live anywhere; when they are page-locked (allocated here) the device-to-host
 * copies of large results (pg_indpairdist, pg_pairwise) run as direct DMA at PCIe speed instead of through the runtime's
 * bounce buffer.  Plain hipHostMalloc / hipHostFree, exposed so that a host without HIP bindings can use them. */
int pg_host_alloc(size_t bytes, void **ptr_out);
int pg_host_free(void *ptr);

/* ---- C1: multi-GPU result gather (RCCL over xGMI), one process per GPU ----------------------------- */
/* Replaces the sorter/writer threads' re-ordering role (popgenWindows.py:108-157).  uid is 128 bytes. */
int pg_comm_unique_id(void *uid128_out);
int pg_comm_init(pg_ctx *ctx, int n_ranks, int rank, const void *uid128);
/* Gather count doubles from every rank into recv[n_ranks*count] on every rank (ncclAllGather). */
int pg_comm_allgather_f64(pg_ctx *ctx, const double *send, double *recv, int64_t count);
int pg_comm_barrier(pg_ctx *ctx);
int pg_comm_destroy(pg_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* POPGEN_HIP_H */
