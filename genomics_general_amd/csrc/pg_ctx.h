// Context object and small device-buffer helper of the C-ABI host layer.
#pragma once
#include "../../include/popgen_hip.h"
#include "pg_internal.h"
#include "pg_inflate.h"
#include "pg_vcf_core.h"

#include <utility>
#include <vector>

#define PG_MAX_HAP 32768
#define PG_TOK_WORKERS 8       // staging threads of the device tokenizer at most (four already fill PCIe: 56 GB/s of text on the round-4 box)
#define PG_TOK_STREAMS 4       // copy streams they share (8 streams: 55.6 GB/s, 33 ms to create; 4: 54.8, 20 ms; 2: 50; 1: 36-45)

int pg_fail(int code, const char *fmt, ...);
// host threads the library may start for one call: every hardware thread, or PG_HOST_THREADS (the drivers set it to cores / ranks
// under a multi-rank launch, so that N ranks on one node do not start N x cores threads)
int pg_host_threads();

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return pg_fail(PG_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    T *base = nullptr;   // what hipMalloc returned: p = base + lead
    size_t cap = 0;      // elements
    size_t lead = 0;     // elements in front of p (placement experiments: tools/pack_variance.py, pg_debug_place)
    int alloc(size_t n) {
        release();
        if (n == 0) n = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&base), (n + lead) * sizeof(T));
        if (e != hipSuccess) {
            base = p = nullptr;
            return pg_fail(PG_ERR_HIP, "hipMalloc(%zu bytes): %s", (n + lead) * sizeof(T), hipGetErrorString(e));
        }
        p = base + lead;
        cap = n;
        return PG_OK;
    }
    int ensure(size_t n) { return n <= cap ? PG_OK : alloc(n); }
    // for buffers whose size follows the blocks of a streamed input (each a little different from the last): an eighth of headroom, so
    // that a block slightly larger than every block before it does not cost a hipFree (which waits for the device) + hipMalloc
    int ensure_roomy(size_t n) { return n <= cap ? PG_OK : alloc(n + n / 8 + 4096); }
    int upload(const T *h, size_t n, hipStream_t st) {
        int rc = ensure(n);
        if (rc != PG_OK) return rc;
        hipError_t e = hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return pg_fail(PG_ERR_HIP, "hipMemcpyAsync H2D: %s", hipGetErrorString(e));
        return PG_OK;
    }
    void release() {
        if (base) (void)hipFree(base);
        base = p = nullptr;
        cap = 0;
    }
};

// page-locked host staging (async copies from/to pageable memory go through a runtime bounce buffer and cost ~20 us each)
template <class T>
struct HostPin {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return PG_OK;
        release();
        if (n < 1024) n = 1024;
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), n * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return pg_fail(PG_ERR_HIP, "hipHostMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
        }
        cap = n;
        return PG_OK;
    }
    int ensure_roomy(size_t n) { return n <= cap ? PG_OK : ensure(n + n / 8 + 4096); }      // (see DevBuf::ensure_roomy)
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct pg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // samples
    int n_hap = 0, n_pops = 0, n_samp = 0;
    int S = 0;    // bytes per site row of the resident buffer (multiple of 16, pad = 0)
    int NP = 0;   // haplotype stride of the bit-planes (multiple of 32; of 64 for the popcount kernels)
    std::vector<int32_t> h_pop_start, h_samp_start;
    DevBuf<int32_t> hap_pop, pop_start, samp_start, slot_gen;
    DevBuf<PgTask2> tasks2;      // v2 pair kernels (haplotype units)
    int n_tasks2 = 0;
    DevBuf<PgTask2> tasksC;      // v2 k_pairC when its units are diploid individuals (diagonal included)
    int n_tasksC = 0;
    DevBuf<PgTask2> tasksCh;     // v2 k_pairC on haplotype units (no diagonal)
    int n_tasksCh = 0;
    bool all_diploid = false;    // every individual owns exactly slots (2k, 2k+1)
    bool pops_on_individuals = false;   // ... and every population begins and ends on an individual boundary
    DevBuf<uint32_t> Vp, XY;     // v2 planes (slot 0; also used by nothing else)
    // v2 software pipeline: k_pack2 of sub-batch k+1 (HBM-bound, stream2) overlaps the pair kernels of sub-batch k
    // (VALU-bound, stream).  Two slots of planes / window tables.
    hipStream_t stream2 = nullptr;
    // ingestion: host -> device copies (and the device-side expansion of packed cells) run on their own stream, so that the
    // upload of input block k+1 overlaps the kernels of block k; up_ev = end of the last queued upload
    hipStream_t stream_up = nullptr;
    hipEvent_t up_ev = nullptr;
    bool up_pending = false;
    DevBuf<uint8_t> cells_stage;     // packed cells of the upload in flight
    // BGZF members inflated on the device (pg_inflate.hip): compressed bytes, member table, status [error bits, first bad member]
    struct Inflate {
        DevBuf<uint32_t> comp, crc_tab, crc_fold;   // crc_tab: k_crc32's tables; crc_fold: pgi_make_crc_tables (the check inside k_inflate)
        DevBuf<uint8_t> text;              // pg_inflate_device only (the tokenizer inflates into its text slots)
        DevBuf<uint8_t> sink;              // 128 bytes per member: where the lanes of a copy step that have no byte store
        DevBuf<uint16_t> nl_list;          // the members' line feeds as k_inflate lists them (nl_cap offsets per member)
        DevBuf<int32_t> nl_cnt;            //   their number per member
        DevBuf<int64_t> mem_base;          //   line feeds in front of each member (k_member_scan)
        uint32_t nl_cap = 0;               //   0: the block's line feeds are found by passes over its text
        hipEvent_t ev_inflated = nullptr, ev_crc = nullptr;   // k_crc32 on a stream of its own (the tokenizer's blocks)
        bool crc_pending = false;
        DevBuf<PgiMember> members;
        HostPin<PgiMember> h_members;
        DevBuf<int32_t> status;
        HostPin<int32_t> h_status;
    } inf;
    // device-side tokenizer (pg_tokenize_*): two blocks in flight, each with its text, line feeds, per-line outputs
    struct TokSlot {
        DevBuf<uint8_t> text;
        uint8_t *tp = nullptr;             // the block's first byte: text.p, or a 16-byte aligned place inside it (BGZF: head + inflated members)
        Inflate inf;                       // the block arrived deflated (pg_tokenize_submit_bgzf)
        bool deflated = false;
        HostPin<uint8_t> h_head;
        DevBuf<uint8_t> names;             // pg_tokenize_run_names: the scaffold names of the block's runs, gathered
        DevBuf<int64_t> names_idx;
        DevBuf<int32_t> i32, dcols, pos;
        DevBuf<int64_t> i64, nl, off, pos64, cells_at;
        HostPin<int64_t> h_total;          // page-locked landing: [0] lines, [1] status | runs
        HostPin<int64_t> h_pos;
        HostPin<int32_t> h_cols;
        std::vector<int32_t> cols;         // col_slot | col_ploidy | cell offsets | cell widths of the submitted block
        hipEvent_t counted = nullptr, staged = nullptr, parsed = nullptr;   // line feeds counted / deflated bytes on the device / rows, positions, status there
        hipEvent_t heads_done = nullptr, pos_copied = nullptr;              // k_tok_heads finished / the positions are on the host (copied on the small stream)
        bool pos_pending = false;
        int state = 0;                     // 0 idle, 1 empty block, 2 submitted, 3 parse queued, 4 empty result
        int fmt = 0, n_cols = 0, max_ploidy = 0, cells_w = 0;
        int64_t len = 0, n_lines = 0, n_tiles = 0, run_cap = 0;
        int64_t n_members = 0, head_len = 0;   // (a block that arrived deflated)
    } tok[2];
    // text deflated on the device (pg_deflate.hip): token buffers of the persistent waves, the members' streams, sizes and CRC-32s, the
    // assembled BGZF members
    struct Deflate {
        DevBuf<uint32_t> tok, out_len, crc;
        DevBuf<uint8_t> slots, comp, text;      // text: pg_bgzf_compress_device only
        DevBuf<int64_t> totals;                 // [0] bytes of text, [1] bytes of the members
        HostPin<int64_t> h_totals;
        void release() { tok.release(); out_len.release(); crc.release(); slots.release(); comp.release(); text.release(); totals.release(); h_totals.release(); }
    } deflate;
    // VCF lines parsed on the device (pg_vcf_dev.hip): the option set of the run and, per text slot of the tokenizer, the per-line
    // records, the rows' sizes and places, the rows' text
    struct VcfDev {
        bool configured = false;
        PgvConfig cfg;
        int waves_per_block = 4;
        DevBuf<uint8_t> contigs, ploidy, fsel, prevkey;      // prevkey: the PgvKey of the last data line seen (--excludeDuplicates)
        DevBuf<int32_t> sel_col;
        DevBuf<uint32_t> cell_off;
        struct Slot {
            DevBuf<uint8_t> lines, out;
            DevBuf<uint32_t> rlen;
            DevBuf<int64_t> roff, status;        // status: [0] bits (1 a line needs the host, 2 the rows exceed `out`), [1] first such line, [2] bytes of the rows, [3] rows, [4] bytes of the rows as BGZF members
            Deflate df;                          // the rows deflated where they lie (pg_vcf_dev_set_output)
            HostPin<int64_t> h_status;
            HostPin<uint8_t> h_prev;             // the key the block started from
            hipEvent_t done = nullptr, rows_ready = nullptr;
            int state = 0;                       // 0 idle, 1 text on its way / there, 2 kernels queued, 3 empty block
            int64_t text_len = 0, out_cap = 0;
            bool no_final_newline = false;
        } s[2];
        bool bgzf_rows = false;                  // pg_vcf_dev_set_output: the rows leave the device as BGZF members
        double kernel_ms = 0;                    // pg_vcf_dev_stats
        int64_t blocks = 0, host_blocks = 0;
    } vcf;
    int64_t tok_nl_fallbacks = 0;                          // deflated blocks whose line feeds were found by passes over the text after all
    HostPin<uint8_t> tok_pin;                              // two 4 MiB page-locked buffers per staging thread
    hipStream_t tok_st[PG_TOK_WORKERS] = {};               // one copy stream per staging thread
    hipStream_t tok_crc = nullptr;                         // k_crc32 of a deflated block, beside the tokenizer's kernels
    hipStream_t tok_parse = nullptr;                       // PG_TOK_PARSE_STREAM=1: a block's parse kernels beside the next block's k_inflate
    hipStream_t tok_small = nullptr;                       // the collect step's few kilobytes (beside the next block's inflate on stream_up)
    hipEvent_t tok_wev[PG_TOK_WORKERS][2] = {};
    double tok_stage_s = 0, tok_kernel_s = 0;              // pg_tokenize_stats: wall seconds of the copies / of everything behind them
    int64_t tok_bytes = 0;
    DevBuf<int32_t> slot_src;        // pg_upload_packed_async: slot -> 2 * cell column + allele
    struct Slot {
        DevBuf<uint32_t> Vp, XV, pres;
        DevBuf<int64_t> win;
        HostPin<int64_t> host;            // pinned staging of [lo | hi | goff | vgoff], alive until its H2D copy completed
        hipEvent_t packed = nullptr, consumed = nullptr;
        bool used = false;
    } slot[2];
    DevBuf<int32_t> flag;        // v2: PG_FLAG_MISMATCH | PG_FLAG_XV_OVERFLOW, raised by the pack kernels
    bool xv_worst = false;       // XV words reserved per compaction group: PG_XV_CAP_DEFAULT, PG_XV_CAP after an overflow
    DevBuf<int32_t> Cfull, Dfull;  // pg_pairwise staging
    DevBuf<uint32_t> hapbits;      // k_hapstats: match matrices as bit rows
    DevBuf<int32_t> hap_order;     // k_hapstats: each population's slots in the reference's row order
    DevBuf<int32_t> site_tmp;      // pg_site_counts staging
    DevBuf<double> site_val;       // pg_site_target: the finished columns (float64 or int64, 8 bytes either way)
    DevBuf<uint8_t> site_keep;     //                 and the rows' keep flags
    DevBuf<uint32_t> site_flags;   // pg_popfreq: one bit per site (k_popfreq_ordered)
    // pi / dxy / Fst in NumPy's summation order (k_popdist_np): the reference's row order within the populations and the rank of
    // the population names (pg_set_reference_order; identity until set), the pairwise-summation trees of the block lengths
    DevBuf<int32_t> ref_row, pop_rank, np_trees, np_task_tree;
    DevBuf<int32_t> samp_rank;     // k_indpair_fin: which individual of a pair supplies the rows of its haplotype block (pg_set_sample_rank)
    int np_state = 0;              // 0: not built for the current samples; 1: usable; -1: a block has too many runs (old finisher)
    int np_max_leaves = 0, np_max_side = 0;
    int sum_order = 0;             // pg_set_sum_order: 0 = NumPy's order up to PG_NP_MAX_SITES sites a window, 1 = for every window, 2 = for none
    const int64_t *cur_win_lo = nullptr, *cur_win_hi = nullptr;   // pairwise_batches: the staged windows of the sub-batch being consumed
    // how the matrices of the last batch are laid out (set by pairwise_batches)
    int cN = 0, cshift = 0;
    // resident sites
    DevBuf<int8_t> gt;
    int64_t cap_sites = 0;
    // scratch
    int64_t scratch_limit = 48ll << 30;
    DevBuf<int32_t> Cmat, Dmat;
    DevBuf<int64_t> win;        // [lo | hi | woff]
    DevBuf<double> res_f64, part_f64, stats;
    // deferred result tables (pg_set_deferred_results): the finaliser writes into one of two device buffers, the copy to the caller's
    // page-locked table runs on res_stream beside the kernels of the next call
    hipStream_t res_stream = nullptr;
    DevBuf<double> res_alt[2];
    hipEvent_t res_fin[2] = {nullptr, nullptr}, res_copied[2] = {nullptr, nullptr};
    int res_flip = 0;
    bool defer_results = false;
    std::vector<hipEvent_t> event_pool;
    DevBuf<int64_t> res_i64, part_i64;
    HostPin<double> out_pin;     // pinned landing zone of small result tables
    HostPin<int64_t> win_pin;    // pinned staging of the window table of the site-statistics kernels
    hipEvent_t win_ev = nullptr; // completion of the last copy out of win_pin
    // timing
    uint32_t time_mask = 0xFFFFFFFFu;   // bit k: kernel family k is bracketed by HIP events (pg_kernel_time_select)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events[PG_K_COUNT_];
    double acc_ms[PG_K_COUNT_] = {};
    int64_t acc_launches[PG_K_COUNT_] = {};
    // RCCL (opaque here)
    void *comm = nullptr;
    int comm_ranks = 0, comm_rank = 0;
    DevBuf<double> comm_send, comm_recv;
    HostPin<double> comm_pin;    // page-locked staging of the all-gather
};

std::vector<PgTask2> pg_make_tasks2(int n, int max_nsub, int diag);
int pg_time_begin(pg_ctx *c, int k, hipEvent_t *e0, hipEvent_t *e1);
int pg_time_end(pg_ctx *c, int k, hipEvent_t e0, hipEvent_t e1, int launches);
