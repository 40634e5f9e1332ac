// C-ABI host layer of libpopgen_hip.so: context, resident site buffer, window batching, kernel timing.
// See include/popgen_hip.h for the contract of every entry point.
#include "pg_ctx.h"

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <thread>

static thread_local char g_err[1024] = "";

int pg_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// CPUs this process may really use: the logical CPUs, cut by the scheduler's affinity mask and by the cgroup's CPU quota (cpu.max
// of cgroup v2, cfs_quota_us / cfs_period_us of v1).  A container on a 256-thread host is often given a few CPUs' worth of time:
// 256 busy threads then share that time, and the CFS bandwidth controller stops ALL of them for the rest of every 100 ms period
// once the quota is used up (measured on the GPU box of round 4: cpu.max = 16 CPUs; 16 staging threads + the Python threads ran
// the text copies at 41 GB/s, 4 threads at 56).
static int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0 && a < n) n = a;
    }
    double quota = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[32] = {0};
        long long period = 0;
        if (fscanf(f, "%31s %lld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / (double)period;
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long long q = -1, period = 0;
        if (fscanf(g, "%lld", &q) != 1) q = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lld", &period) != 1) period = 0;
            fclose(h);
        }
        if (q > 0 && period > 0) quota = (double)q / (double)period;
    }
    if (quota >= 1.0 && quota < n) n = (int)(quota + 0.5);
    return n < 1 ? 1 : n;
}

int pg_host_threads() {
    static const int usable = usable_cpus();
    int nt = usable;
    if (const char *e = getenv("PG_HOST_THREADS")) {
        const int v = atoi(e);
        if (v > 0) nt = v;
    }
    return nt < 1 ? 1 : nt;
}

extern "C" int pg_usable_cpus(void) {
    static const int usable = usable_cpus();
    return usable;
}

extern "C" const char *pg_last_error(void) { return g_err; }
extern "C" int pg_abi_version(void) { return PG_ABI_VERSION; }

static double g_ctx_times[3] = {0.0, 0.0, 0.0};

extern "C" int pg_device_count(int *n_out) {
    if (!n_out) return pg_fail(PG_ERR_ARG, "pg_device_count: null output");
    int n = 0;
    const auto t_a = std::chrono::steady_clock::now();
    hipError_t e = hipGetDeviceCount(&n);
    if (g_ctx_times[0] == 0.0) g_ctx_times[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
    if (e != hipSuccess) {
        *n_out = 0;
        (void)hipGetLastError();
        return pg_fail(PG_ERR_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n_out = n;
    return PG_OK;
}

extern "C" int pg_ctx_create(pg_ctx **out, int device) {
    if (!out) return pg_fail(PG_ERR_ARG, "pg_ctx_create: null output");
    *out = nullptr;
    int n = 0;
    int rc = pg_device_count(&n);
    if (rc != PG_OK) return rc;
    if (n <= 0) return pg_fail(PG_ERR_NODEV, "no HIP device visible: the popgen engine needs an AMD GPU (gfx950)");
    if (device < 0 || device >= n) return pg_fail(PG_ERR_ARG, "device %d out of range [0,%d)", device, n);
    const auto t_a = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(device));
    pg_ctx *c = new pg_ctx();
    c->device = device;
    if (getenv("PG_SCRATCH_GIB")) c->scratch_limit = (int64_t)atol(getenv("PG_SCRATCH_GIB")) << 30;    // default 48 GiB
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    g_ctx_times[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();   // hipSetDevice + the first stream
    if (e != hipSuccess) {
        delete c;
        return pg_fail(PG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return pg_fail(PG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->up_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        (void)hipStreamDestroy(c->stream2);
        if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
        delete c;
        return pg_fail(PG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    g_ctx_times[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count() - g_ctx_times[1];   // the other two streams + an event
    *out = c;
    return PG_OK;
}

// where the creation of the first context went (seconds): [0] hipGetDeviceCount (the runtime's start-up: hipInit), [1] hipSetDevice +
// the first stream, [2] two more streams and an event (tools/ctx_time.py)
extern "C" int pg_ctx_create_times(double *out3) {
    if (!out3) return pg_fail(PG_ERR_ARG, "pg_ctx_create_times: null output");
    for (int k = 0; k < 3; ++k) out3[k] = g_ctx_times[k];
    return PG_OK;
}

static void drop_events(pg_ctx *c) {
    for (auto ev : c->event_pool) (void)hipEventDestroy(ev);
    c->event_pool.clear();
    for (int k = 0; k < PG_K_COUNT_; ++k) {
        for (auto &pr : c->events[k]) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        c->events[k].clear();
        c->acc_ms[k] = 0.0;
        c->acc_launches[k] = 0;
    }
}

extern "C" int pg_ctx_destroy(pg_ctx *c) {
    if (!c) return PG_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamSynchronize(c->stream_up);
    pg_comm_destroy(c);
    for (int k = 0; k < 2; ++k) {
        c->slot[k].Vp.release();
        c->slot[k].XV.release();
        c->slot[k].pres.release();
        c->slot[k].host.release();
        c->slot[k].win.release();
        if (c->slot[k].packed) (void)hipEventDestroy(c->slot[k].packed);
        if (c->slot[k].consumed) (void)hipEventDestroy(c->slot[k].consumed);
    }
    if (c->win_ev) (void)hipEventDestroy(c->win_ev);
    (void)hipStreamDestroy(c->stream2);
    (void)hipStreamDestroy(c->stream_up);
    if (c->up_ev) (void)hipEventDestroy(c->up_ev);
    for (int t = 0; t < PG_TOK_WORKERS; ++t) {
        if (c->tok_st[t]) (void)hipStreamDestroy(c->tok_st[t]);
        if (t == 0 && c->tok_small) (void)hipStreamDestroy(c->tok_small);
        if (t == 0 && c->tok_parse) (void)hipStreamDestroy(c->tok_parse);
        if (t == 0 && c->tok_crc) (void)hipStreamDestroy(c->tok_crc);
        for (int k = 0; k < 2; ++k)
            if (c->tok_wev[t][k]) (void)hipEventDestroy(c->tok_wev[t][k]);
    }
    c->cells_stage.release();
    c->slot_src.release();
    for (int k = 0; k < 2; ++k) {
        pg_ctx::TokSlot &T = c->tok[k];
        T.text.release(); T.i32.release(); T.dcols.release(); T.pos.release(); T.pos64.release(); T.cells_at.release(); T.i64.release(); T.nl.release(); T.off.release();
        T.h_total.release(); T.h_pos.release(); T.h_cols.release();
        T.h_head.release(); T.names.release(); T.names_idx.release();
        auto drop = [](pg_ctx::Inflate &I) {
            I.comp.release(); I.crc_tab.release(); I.crc_fold.release(); I.text.release(); I.sink.release(); I.members.release(); I.h_members.release(); I.status.release(); I.h_status.release();
            I.nl_list.release(); I.nl_cnt.release(); I.mem_base.release();
            if (I.ev_inflated) (void)hipEventDestroy(I.ev_inflated);
            if (I.ev_crc) (void)hipEventDestroy(I.ev_crc);
            I.ev_inflated = I.ev_crc = nullptr;
        };
        drop(T.inf);
        if (k == 0) drop(c->inf);
        if (T.counted) (void)hipEventDestroy(T.counted);
        if (T.staged) (void)hipEventDestroy(T.staged);
        if (T.parsed) (void)hipEventDestroy(T.parsed);
        if (T.heads_done) (void)hipEventDestroy(T.heads_done);
        if (T.pos_copied) (void)hipEventDestroy(T.pos_copied);
    }
    c->deflate.release();
    c->vcf.prevkey.release(); c->vcf.contigs.release(); c->vcf.ploidy.release(); c->vcf.fsel.release(); c->vcf.sel_col.release(); c->vcf.cell_off.release();
    for (int k = 0; k < 2; ++k) {
        pg_ctx::VcfDev::Slot &V = c->vcf.s[k];
        V.lines.release(); V.out.release(); V.rlen.release(); V.roff.release(); V.status.release(); V.h_status.release(); V.h_prev.release(); V.df.release();
        if (V.done) (void)hipEventDestroy(V.done);
        if (V.rows_ready) (void)hipEventDestroy(V.rows_ready);
    }
    c->tok_pin.release();
    drop_events(c);
    c->gt.release();
    c->hap_pop.release();
    c->pop_start.release();
    c->samp_start.release();
    c->tasks2.release();
    c->tasksC.release();
    c->tasksCh.release();
    c->flag.release();
    c->out_pin.release();
    c->win_pin.release();
    c->Cfull.release();
    c->Dfull.release();
    c->hapbits.release();
    c->hap_order.release();
    if (c->res_stream) { (void)hipStreamSynchronize(c->res_stream); (void)hipStreamDestroy(c->res_stream); }
    for (int f = 0; f < 2; ++f) {
        c->res_alt[f].release();
        if (c->res_fin[f]) (void)hipEventDestroy(c->res_fin[f]);
        if (c->res_copied[f]) (void)hipEventDestroy(c->res_copied[f]);
    }
    c->site_tmp.release();
    c->site_val.release();
    c->site_keep.release();
    c->site_flags.release();
    c->ref_row.release();
    c->samp_rank.release();
    c->pop_rank.release();
    c->np_trees.release();
    c->np_task_tree.release();
    c->Vp.release();
    c->XY.release();
    c->Cmat.release();
    c->Dmat.release();
    c->win.release();
    c->res_f64.release();
    c->stats.release();
    c->res_i64.release();
    c->part_f64.release();
    c->part_i64.release();
    c->slot_gen.release();
    (void)hipStreamDestroy(c->stream);
    delete c;
    return PG_OK;
}

extern "C" int pg_sync(pg_ctx *c) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->res_stream) HIPCHK(hipStreamSynchronize(c->res_stream));
    return PG_OK;
}

extern "C" int pg_set_deferred_results(pg_ctx *c, int on) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    if (c->res_stream) HIPCHK(hipStreamSynchronize(c->res_stream));
    c->defer_results = on != 0;
    return PG_OK;
}

extern "C" int pg_results_wait(pg_ctx *c) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    if (c->res_stream) HIPCHK(hipStreamSynchronize(c->res_stream));
    return PG_OK;
}

extern "C" int pg_host_alloc(size_t bytes, void **ptr_out) {
    if (!ptr_out) return pg_fail(PG_ERR_ARG, "pg_host_alloc: null argument");
    *ptr_out = nullptr;
    HIPCHK(hipHostMalloc(ptr_out, bytes ? bytes : 1, hipHostMallocPortable));       // usable by every device and host thread
    return PG_OK;
}

extern "C" int pg_host_free(void *ptr) {
    if (ptr) HIPCHK(hipHostFree(ptr));
    return PG_OK;
}

extern "C" int pg_set_scratch_limit(pg_ctx *c, int64_t bytes) {
    if (!c || bytes < (64ll << 20)) return pg_fail(PG_ERR_ARG, "scratch limit must be >= 64 MiB");
    c->scratch_limit = bytes;
    return PG_OK;
}

// ---- v2 pair-kernel task table -------------------------------------------------------------------------
// Units [0,n).  Full 64-column chunks cover columns [0, 64*floor(n/64)) with the rows above them (i<j); the remaining
// R = n mod 64 units are handled as ROWS against every column chunk (pairs j<i, written at (j,i)), so no wave runs with
// only R of its 64 lanes useful.  Rows come in sub-tiles of 8, up to max_nsub sub-tiles per wave.
// Circulant task table of k_pairC / k_pairD (see pair_store_circ): `rows`-row blocks x runs of 64 consecutive columns (mod n).
std::vector<PgTask2> pg_make_tasks_circ(int n, int rows) {
    std::vector<PgTask2> out;
    const int ncols = std::min(n, rows + n / 2);           // columns row0 .. row0 + rows - 1 + floor(n/2)
    for (int r0 = 0; r0 < n; r0 += rows)
        for (int k = 0; k < ncols; k += 64) {
            PgTask2 t;
            t.row0 = r0;
            t.col0 = (r0 + k) % n;
            t.nsub = std::min(64, ncols - k);
            t.lower = 2;
            out.push_back(t);
        }
    return out;
}

// ---- samples ----------------------------------------------------------------------------------------
extern "C" int pg_set_samples(pg_ctx *c, int n_hap, const int32_t *hap_pop, const int32_t *hap_sample, int n_pops) {
    if (!c || !hap_pop || !hap_sample) return pg_fail(PG_ERR_ARG, "pg_set_samples: null argument");
    if (n_hap < 1 || n_hap > PG_MAX_HAP) return pg_fail(PG_ERR_ARG, "n_hap %d out of range [1,%d]", n_hap, PG_MAX_HAP);
    if (n_pops < 0) return pg_fail(PG_ERR_ARG, "n_pops < 0");
    HIPCHK(hipSetDevice(c->device));
    // populations: contiguous, increasing, -1 last
    std::vector<int32_t> pstart(n_pops + 1, 0);
    int prev = (n_pops > 0 ? 0 : -1), h = 0;
    {
        std::vector<int32_t> count(n_pops, 0);
        int last = -2;
        bool seen_none = false;
        for (h = 0; h < n_hap; ++h) {
            int p = hap_pop[h];
            if (p < -1 || p >= n_pops) return pg_fail(PG_ERR_ARG, "hap_pop[%d]=%d out of range", h, p);
            if (p == -1) { seen_none = true; continue; }
            if (seen_none) return pg_fail(PG_ERR_ARG, "haplotype slots without a population must come last (slot %d)", h);
            if (last != -2 && p < last) return pg_fail(PG_ERR_ARG, "populations must be contiguous and increasing in slot order (slot %d)", h);
            last = p;
            count[p]++;
        }
        int acc = 0;
        for (int p = 0; p < n_pops; ++p) { pstart[p] = acc; acc += count[p]; }
        pstart[n_pops] = acc;
        (void)prev;
    }
    // individuals: contiguous runs
    std::vector<int32_t> sstart;
    {
        int last = -1;
        std::vector<char> used;
        for (h = 0; h < n_hap; ++h) {
            int s = hap_sample[h];
            if (s < 0) return pg_fail(PG_ERR_ARG, "hap_sample[%d] < 0", h);
            if (s != last) {
                if ((size_t)s >= used.size()) used.resize(s + 1, 0);
                if (used[s]) return pg_fail(PG_ERR_ARG, "haplotype slots of individual %d are not contiguous", s);
                if (s != (int)sstart.size()) return pg_fail(PG_ERR_ARG, "individuals must be numbered 0.. in slot order (slot %d has %d)", h, s);
                used[s] = 1;
                sstart.push_back(h);
                last = s;
            }
        }
        sstart.push_back(n_hap);
    }
    c->n_hap = n_hap;
    c->n_pops = n_pops;
    c->n_samp = (int)sstart.size() - 1;
    c->S = (n_hap + 15) / 16 * 16;
    // plane stride: 32 haplotypes (one tile of the matrix-core pair kernels); the popcount kernels work on 64-lane column chunks
    c->NP = getenv("PG_PAIR_VALU") ? (n_hap + 63) / 64 * 64 : (n_hap + 31) / 32 * 32;
    c->h_pop_start = pstart;
    c->h_samp_start = sstart;
    {   // the reference's row order / population rank: identity until pg_set_reference_order
        std::vector<int32_t> ident((size_t)std::max(n_hap, n_pops) + 1);
        for (size_t k = 0; k < ident.size(); ++k) ident[k] = (int32_t)k;
        int r2;
        if ((r2 = c->ref_row.upload(ident.data(), (size_t)n_hap, c->stream)) != PG_OK) return r2;
        if ((r2 = c->pop_rank.upload(ident.data(), (size_t)std::max(n_pops, 1), c->stream)) != PG_OK) return r2;
        if ((r2 = c->samp_rank.upload(ident.data(), (size_t)n_hap, c->stream)) != PG_OK) return r2;        // (n_samp <= n_hap)
        HIPCHK(hipStreamSynchronize(c->stream));
        c->np_state = 0;
    }
    std::vector<PgTask2> tasks2 = pg_make_tasks_circ(n_hap, 16);       // k_pairD: 16-row circulant tasks
    c->n_tasks2 = (int)tasks2.size();
    c->all_diploid = (n_hap % 2 == 0);
    for (size_t k = 0; k + 1 < sstart.size() && c->all_diploid; ++k)
        if (sstart[k + 1] - sstart[k] != 2) c->all_diploid = false;
    // the fast forms of k_popdist_fin (<1>, <2>) walk a population individual by individual (8-byte loads at its even slots): they
    // need every population to begin and end on an individual boundary.  A caller of the C-ABI may put the two haplotypes of an
    // individual into different populations (the Python layout never does): the general form then
    c->pops_on_individuals = true;
    for (int p = 0; p <= n_pops; ++p)
        if (pstart[p] % 2) c->pops_on_individuals = false;
    std::vector<PgTask2> tasksC;
    if (c->all_diploid) tasksC = pg_make_tasks_circ(n_hap / 2, 8);         // k_pairC works on 8-row circulant tasks
    c->n_tasksC = (int)tasksC.size();
    std::vector<PgTask2> tasksCh = pg_make_tasks_circ(n_hap, 8);
    c->n_tasksCh = (int)tasksCh.size();
    int rc;
    if ((rc = c->hap_pop.upload(hap_pop, n_hap, c->stream)) != PG_OK) return rc;
    if ((rc = c->pop_start.upload(pstart.data(), pstart.size(), c->stream)) != PG_OK) return rc;
    if ((rc = c->samp_start.upload(sstart.data(), sstart.size(), c->stream)) != PG_OK) return rc;
    if (!tasks2.empty() && (rc = c->tasks2.upload(tasks2.data(), tasks2.size(), c->stream)) != PG_OK) return rc;
    if (!tasksC.empty() && (rc = c->tasksC.upload(tasksC.data(), tasksC.size(), c->stream)) != PG_OK) return rc;
    if (!tasksCh.empty() && (rc = c->tasksCh.upload(tasksCh.data(), tasksCh.size(), c->stream)) != PG_OK) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    // the resident buffer layout depends on S: drop it
    c->gt.release();
    c->cap_sites = 0;
    return PG_OK;
}

// ---- resident site buffer ---------------------------------------------------------------------------
extern "C" int pg_reserve_sites(pg_ctx *c, int64_t n_sites) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (n_sites < 0) return pg_fail(PG_ERR_ARG, "n_sites < 0");
    HIPCHK(hipSetDevice(c->device));
    if (n_sites <= c->cap_sites) return PG_OK;
    HIPCHK(hipStreamSynchronize(c->stream_up));
    c->up_pending = false;
    c->gt.release();
    int rc = c->gt.alloc((size_t)(n_sites + 32) * c->S);     // +32 rows so a word tile never reads past the end
    if (rc != PG_OK) return rc;
    HIPCHK(hipMemsetAsync(c->gt.p, 0, (size_t)(n_sites + 32) * c->S, c->stream));
    // the rows are filled through other streams too (asynchronous uploads, the device tokenizer on the copy stream): nothing may
    // be queued there while the clearing is still on its way -- it would wipe rows that were already written
    HIPCHK(hipStreamSynchronize(c->stream));
    c->cap_sites = n_sites;
    return PG_OK;
}

extern "C" int pg_upload_sites(pg_ctx *c, int64_t off, const int8_t *gt, int64_t n) {
    if (!c || (!gt && n > 0)) return pg_fail(PG_ERR_ARG, "pg_upload_sites: null argument");
    if (off < 0 || n < 0 || off + n > c->cap_sites) return pg_fail(PG_ERR_ARG, "sites [%lld,%lld) exceed reserved %lld", (long long)off, (long long)(off + n), (long long)c->cap_sites);
    if (n == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy2DAsync(c->gt.p + off * c->S, c->S, gt, c->n_hap, c->n_hap, (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

extern "C" int pg_download_sites(pg_ctx *c, int64_t off, int8_t *gt_out, int64_t n) {
    if (!c || (!gt_out && n > 0)) return pg_fail(PG_ERR_ARG, "pg_download_sites: null argument");
    if (off < 0 || n < 0 || off + n > c->cap_sites) return pg_fail(PG_ERR_ARG, "sites out of range");
    if (n == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy2DAsync(gt_out, c->n_hap, c->gt.p + off * c->S, c->S, c->n_hap, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

extern "C" int pg_move_rows(pg_ctx *c, int64_t src_row, int64_t dst_row, int64_t n) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (n < 0 || src_row < 0 || dst_row < 0 || src_row + n > c->cap_sites || dst_row + n > c->cap_sites)
        return pg_fail(PG_ERR_ARG, "pg_move_rows: rows out of the reserved range");
    if (n == 0 || src_row == dst_row) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = (size_t)n * c->S;
    const bool overlap = src_row < dst_row + n && dst_row < src_row + n;
    // (on the copy stream: the ingestion thread moves the rows it carries into the half of the resident buffer the next block is
    // tokenised into while the compute stream works on the windows of the current block in the other half)
    hipStream_t st = c->stream_up;
    if (!overlap) {
        HIPCHK(hipMemcpyAsync(c->gt.p + dst_row * c->S, c->gt.p + src_row * c->S, bytes, hipMemcpyDeviceToDevice, st));
    } else {                                                       // through the (idle) staging buffer of the packed uploads
        if (bytes > c->cells_stage.cap) {
            int rc = c->cells_stage.alloc(bytes);
            if (rc != PG_OK) return rc;
        }
        HIPCHK(hipMemcpyAsync(c->cells_stage.p, c->gt.p + src_row * c->S, bytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(c->gt.p + dst_row * c->S, c->cells_stage.p, bytes, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return PG_OK;
}

extern "C" int pg_row_pitch(pg_ctx *c, int *pitch_out) {
    if (!c || !pitch_out) return pg_fail(PG_ERR_ARG, "pg_row_pitch: null argument");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    *pitch_out = c->S;
    return PG_OK;
}

// Asynchronous upload on the context's copy stream.  The host rows must stay valid (and should be page-locked: pg_host_alloc)
// until pg_upload_wait returns.  Rows whose pitch equals pg_row_pitch (pad bytes zero) go down as one linear copy.
extern "C" int pg_upload_sites_async(pg_ctx *c, int64_t off, const int8_t *gt, int64_t n, int64_t row_pitch) {
    if (!c || (!gt && n > 0)) return pg_fail(PG_ERR_ARG, "pg_upload_sites_async: null argument");
    if (off < 0 || n < 0 || off + n > c->cap_sites) return pg_fail(PG_ERR_ARG, "sites [%lld,%lld) exceed reserved %lld", (long long)off, (long long)(off + n), (long long)c->cap_sites);
    if (row_pitch < c->n_hap) return pg_fail(PG_ERR_ARG, "row pitch %lld is smaller than the %d haplotypes of a row", (long long)row_pitch, c->n_hap);
    if (n == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    if (row_pitch == c->S)
        HIPCHK(hipMemcpyAsync(c->gt.p + off * c->S, gt, (size_t)n * c->S, hipMemcpyHostToDevice, c->stream_up));
    else
        HIPCHK(hipMemcpy2DAsync(c->gt.p + off * c->S, c->S, gt, (size_t)row_pitch, c->n_hap, (size_t)n, hipMemcpyHostToDevice, c->stream_up));
    HIPCHK(hipEventRecord(c->up_ev, c->stream_up));
    c->up_pending = true;
    return PG_OK;
}

// Packed cells (one byte per genotype cell: first allele's one-hot code | second << 4, the `.pgeno` payload) are copied as they
// are -- half the PCIe bytes of a diploid data set -- and expanded into resident rows by k_unpack on the copy stream.
extern "C" int pg_upload_packed_async(pg_ctx *c, int64_t off, const uint8_t *cells, int64_t n, int n_cols, const int32_t *slot_src) {
    if (!c || ((!cells || !slot_src) && n > 0)) return pg_fail(PG_ERR_ARG, "pg_upload_packed_async: null argument");
    if (off < 0 || n < 0 || off + n > c->cap_sites) return pg_fail(PG_ERR_ARG, "sites [%lld,%lld) exceed reserved %lld", (long long)off, (long long)(off + n), (long long)c->cap_sites);
    if (n_cols < 1) return pg_fail(PG_ERR_ARG, "n_cols < 1");
    if (n == 0) return PG_OK;
    for (int h = 0; h < c->n_hap; ++h)
        if (slot_src[h] < -1 || slot_src[h] >= 2 * n_cols) return pg_fail(PG_ERR_ARG, "slot_src[%d] = %d out of range", h, slot_src[h]);
    HIPCHK(hipSetDevice(c->device));
    // the staging buffer and the table are reused by every upload: earlier users on the copy stream are done in stream order,
    // a reallocation waits for them explicitly
    if ((size_t)n * n_cols > c->cells_stage.cap) {
        HIPCHK(hipStreamSynchronize(c->stream_up));
        int rc = c->cells_stage.alloc((size_t)n * n_cols);
        if (rc != PG_OK) return rc;
    }
    int rc = c->slot_src.upload(slot_src, (size_t)c->n_hap, c->stream_up);
    if (rc != PG_OK) return rc;
    HIPCHK(hipMemcpyAsync(c->cells_stage.p, cells, (size_t)n * n_cols, hipMemcpyHostToDevice, c->stream_up));
    pg_launch_unpack(c->stream_up, c->cells_stage.p, n_cols, n, c->slot_src.p, c->n_hap, c->gt.p + off * c->S, c->S);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->up_ev, c->stream_up));
    c->up_pending = true;
    return PG_OK;
}

extern "C" int pg_upload_wait(pg_ctx *c) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (!c->up_pending) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->up_ev));
    c->up_pending = false;
    return PG_OK;
}

// ---- placement experiments (tools/pack_variance.py): where the big buffers sit, and a way to shift them ----------------------
// which: 0 resident rows, 1 called plane (slot 0), 2 XV planes (slot 0).  pg_debug_place releases the buffer and makes its next
// allocation start lead_bytes behind what hipMalloc returns (the resident rows are lost: fill them again).
extern "C" int pg_debug_address(pg_ctx *c, int which, uint64_t *addr_out, uint64_t *bytes_out) {
    if (!c || !addr_out || !bytes_out) return pg_fail(PG_ERR_ARG, "pg_debug_address: null argument");
    switch (which) {
        case 0: *addr_out = (uint64_t)c->gt.p; *bytes_out = c->gt.cap; break;
        case 1: *addr_out = (uint64_t)c->slot[0].Vp.p; *bytes_out = c->slot[0].Vp.cap * 4; break;
        case 2: *addr_out = (uint64_t)c->slot[0].XV.p; *bytes_out = c->slot[0].XV.cap * 4; break;
        default: return pg_fail(PG_ERR_ARG, "pg_debug_address: which = %d", which);
    }
    return PG_OK;
}

extern "C" int pg_debug_place(pg_ctx *c, int which, uint64_t lead_bytes) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    switch (which) {
        case 0: c->gt.release(); c->gt.lead = lead_bytes; c->cap_sites = 0; break;
        case 1: c->slot[0].Vp.release(); c->slot[0].Vp.lead = lead_bytes / 4; break;
        case 2: c->slot[0].XV.release(); c->slot[0].XV.lead = lead_bytes / 4; break;
        default: return pg_fail(PG_ERR_ARG, "pg_debug_place: which = %d", which);
    }
    return PG_OK;
}

// CU partition experiment (tools/cu_split_sweep.py; VERDICT round 5 #4a): ctx->stream (pair kernels, finalisers) is recreated on
// `pair_cus_per_xcd` compute units of every XCD, ctx->stream2 (the pack kernel of the two-stream pipeline, PG_OVERLAP=1) on the
// others; 0 = plain streams again.  (A queue's CU mask is dealt out to the XCDs bit by bit: bit i = CU i / 8 of XCD i % 8.)
extern "C" int pg_debug_cu_split(pg_ctx *c, int pair_cus_per_xcd) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (pair_cus_per_xcd < 0 || pair_cus_per_xcd > 31) return pg_fail(PG_ERR_ARG, "pg_debug_cu_split: 0 ... 31 compute units per XCD");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    hipStream_t a = nullptr, b = nullptr;
    if (pair_cus_per_xcd == 0) {
        HIPCHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    } else {
        uint32_t ma[8], mb[8];
        for (int w = 0; w < 8; ++w) { ma[w] = 0u; mb[w] = 0u; }
        for (int i = 0; i < 256; ++i) ((i / 8 < pair_cus_per_xcd) ? ma : mb)[i >> 5] |= 1u << (i & 31);
        HIPCHK(hipExtStreamCreateWithCUMask(&a, 8, ma));
        HIPCHK(hipExtStreamCreateWithCUMask(&b, 8, mb));
    }
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->stream2);
    c->stream = a;
    c->stream2 = b;
    return PG_OK;
}

// ---- kernel timing ----------------------------------------------------------------------------------
// timing events are pooled: creating a pair per launch costs more host time than a small kernel
static int event_get(pg_ctx *c, hipEvent_t *e) {
    if (!c->event_pool.empty()) {
        *e = c->event_pool.back();
        c->event_pool.pop_back();
        return PG_OK;
    }
    HIPCHK(hipEventCreate(e));
    return PG_OK;
}

int pg_time_begin(pg_ctx *c, int k, hipEvent_t *e0, hipEvent_t *e1) {
    int rc;
    *e0 = *e1 = nullptr;
    if (!((c->time_mask >> k) & 1u)) return PG_OK;
    if ((rc = event_get(c, e0)) != PG_OK) return rc;
    if ((rc = event_get(c, e1)) != PG_OK) return rc;
    HIPCHK(hipEventRecord(*e0, c->stream));
    (void)k;
    return PG_OK;
}

int pg_time_end(pg_ctx *c, int k, hipEvent_t e0, hipEvent_t e1, int launches) {
    if (!e0) return PG_OK;
    HIPCHK(hipEventRecord(e1, c->stream));
    c->events[k].push_back(std::make_pair(e0, e1));
    c->acc_launches[k] += launches;
    return PG_OK;
}

static int fold_events(pg_ctx *c);

static int fold_events(pg_ctx *c) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->stream2));
    if (c->res_stream) HIPCHK(hipStreamSynchronize(c->res_stream));
    for (int k = 0; k < PG_K_COUNT_; ++k) {
        for (auto &pr : c->events[k]) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
            c->acc_ms[k] += ms;
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        c->events[k].clear();
    }
    return PG_OK;
}

extern "C" int pg_kernel_time_select(pg_ctx *c, uint32_t mask) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    c->time_mask = mask;
    return PG_OK;
}

extern "C" int pg_kernel_time(pg_ctx *c, int kernel_id, double *ms_out, int64_t *launches_out) {
    if (!c || kernel_id < 0 || kernel_id >= PG_K_COUNT_) return pg_fail(PG_ERR_ARG, "bad kernel id");
    HIPCHK(hipSetDevice(c->device));
    int rc = fold_events(c);
    if (rc != PG_OK) return rc;
    if (ms_out) *ms_out = c->acc_ms[kernel_id];
    if (launches_out) *launches_out = c->acc_launches[kernel_id];
    return PG_OK;
}

extern "C" int pg_kernel_time_reset(pg_ctx *c) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    int rc = fold_events(c);
    if (rc != PG_OK) return rc;
    for (int k = 0; k < PG_K_COUNT_; ++k) { c->acc_ms[k] = 0.0; c->acc_launches[k] = 0; }
    return PG_OK;
}

// ---- synthetic fill ---------------------------------------------------------------------------------
extern "C" int pg_synth_fill(pg_ctx *c, int64_t off, int64_t n, int64_t first_site_index, uint64_t seed,
                             int64_t scaf_len, int32_t n_dip, int32_t n_pops_gen, const int32_t *slot_gen_hap,
                             int32_t var_thr, int32_t miss_thr) {
    if (!c || !slot_gen_hap) return pg_fail(PG_ERR_ARG, "pg_synth_fill: null argument");
    if (off < 0 || n < 0 || off + n > c->cap_sites) return pg_fail(PG_ERR_ARG, "sites out of reserved range");
    if (scaf_len < 1 || n_dip < 1 || n_pops_gen < 1) return pg_fail(PG_ERR_ARG, "bad generator parameters");
    for (int h = 0; h < c->n_hap; ++h)
        if (slot_gen_hap[h] < 0 || slot_gen_hap[h] >= 2 * n_dip) return pg_fail(PG_ERR_ARG, "slot_gen_hap[%d] out of range", h);
    HIPCHK(hipSetDevice(c->device));
    int rc = c->slot_gen.upload(slot_gen_hap, c->n_hap, c->stream);
    if (rc != PG_OK) return rc;
    PgSynthParams p;
    p.seed = seed; p.first_site_index = first_site_index; p.scaf_len = scaf_len;
    p.n_dip = n_dip; p.n_pops = n_pops_gen; p.var_thr = var_thr; p.miss_thr = miss_thr;
    hipEvent_t e0, e1;
    if ((rc = pg_time_begin(c, PG_K_SYNTH, &e0, &e1)) != PG_OK) return rc;
    pg_launch_synth(c->stream, c->gt.p, c->S, c->n_hap, off, n, c->slot_gen.p, p);
    if ((rc = pg_time_end(c, PG_K_SYNTH, e0, e1, 1)) != PG_OK) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

// ---- window validation + batching -------------------------------------------------------------------
static int check_windows(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (n_win < 0 || (n_win > 0 && (!lo || !hi))) return pg_fail(PG_ERR_ARG, "bad window arrays");
    for (int w = 0; w < n_win; ++w) {
        if (lo[w] < 0 || hi[w] < lo[w] || hi[w] > c->cap_sites)
            return pg_fail(PG_ERR_ARG, "window %d = [%lld,%lld) outside resident sites [0,%lld)", w, (long long)lo[w], (long long)hi[w], (long long)c->cap_sites);
        if (hi[w] - lo[w] > 0x7FFFFFFFll) return pg_fail(PG_ERR_ARG, "window %d longer than 2^31-1 sites", w);
    }
    return PG_OK;
}

// Upload lo/hi (+ word offsets) of windows [w0,w1) into ctx->win = [lo | hi | woff(n+1)].
static int stage_windows(pg_ctx *c, const int64_t *lo, const int64_t *hi, int w0, int w1, int64_t *total_words,
                         int *max_words, int64_t *max_len) {
    int n = w1 - w0;
    // pinned staging: the copy is asynchronous and needs no synchronisation here (every entry point synchronises
    // ctx->stream before it returns or before it stages again, so the buffer is never rewritten while in flight)
    if (c->win_ev) HIPCHK(hipEventSynchronize(c->win_ev));          // previous copy out of the staging buffer is done
    else HIPCHK(hipEventCreateWithFlags(&c->win_ev, hipEventDisableTiming));
    int rc0 = c->win_pin.ensure(3 * (size_t)n + 1);
    if (rc0 != PG_OK) return rc0;
    int64_t *h = c->win_pin.p;
    int64_t acc = 0;
    int mw = 0;
    int64_t ml = 0;
    for (int k = 0; k < n; ++k) {
        h[k] = lo[w0 + k];
        h[n + k] = hi[w0 + k];
        h[2 * (size_t)n + k] = acc;
        int64_t len = hi[w0 + k] - lo[w0 + k];
        int words = (int)((len + 31) / 32);
        acc += words;
        mw = std::max(mw, words);
        ml = std::max(ml, len);
    }
    h[3 * (size_t)n] = acc;
    if (total_words) *total_words = acc;
    if (max_words) *max_words = mw;
    if (max_len) *max_len = ml;
    rc0 = c->win.upload(h, 3 * (size_t)n + 1, c->stream);
    if (rc0 != PG_OK) return rc0;
    HIPCHK(hipEventRecord(c->win_ev, c->stream));
    return PG_OK;
}

// Windows are cut into sub-batches; k_pack2 of sub-batch k+1 runs on ctx->stream2 while k_pairC / k_pairD and
// `consume` of sub-batch k run on ctx->stream (two slots of planes).  `consume(batch_w0, batch_n)` is called with the
// batch's matrices queued on ctx->stream: D in ctx->Dmat ([N][N] per window, upper triangle) and the called counts in
// ctx->Cmat ([cN][cN] per window, upper triangle, entry of haplotypes (i,j) at (i>>cshift, j>>cshift)).
template <class F>
static int pairwise_batches(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, bool dip, F consume) {
    const int N = c->n_hap, NP = c->NP;
    const int n_units = dip ? N / 2 : N;
    const int NPv = dip ? (NP % 64 ? (n_units + 31) / 32 * 32 : (n_units + 63) / 64 * 64) : NP;
    c->cN = n_units;
    c->cshift = dip ? 1 : 0;
    const int64_t mat_bytes = 4ll * N * N + 4ll * n_units * n_units;
    // compaction group = words per pack block: 128 when that still oversubscribes the chip's wave slots several times (see
    // pg_internal.h), else 64, and 32 when 64 would leave fewer than two waves per SIMD slot (C2: 4883 one-wave blocks; measured
    // 0.44 - 0.47 ms with 32 against 0.465 - 0.484 with 64); PG_GROUP_WORDS overrides (A/B runs)
    int grp = PG_GROUP;
    {
        int64_t blocks64 = 0;
        for (int w = 0; w < n_win; ++w) blocks64 += ((hi[w] - lo[w] + 31) / 32 + PG_GROUP - 1) / PG_GROUP;
        const int waves_per_block = (NP / 4 + 63) / 64;
        if (blocks64 * waves_per_block >= 32768) grp = PG_GROUP_MAX;
        else if (blocks64 * waves_per_block < 8192) grp = PG_GROUP / 2;
        if (const char *g = getenv("PG_GROUP_WORDS")) grp = std::min(PG_GROUP_MAX, std::max(8, atoi(g) / 4 * 4));
    }
    // scratch bytes per 32-site input word of one slot: called plane + reserved virtual-site planes (capg words per group)
    const int capg = c->xv_worst ? PG_XV_CAP(grp) : PG_XV_CAP_DEFAULT(grp);
    const int64_t word_bytes = ((int64_t)NP * 4 * PG_XV_PLANES * capg + grp - 1) / grp + (int64_t)NPv * 4;
    // sub-batch size: at most half the scratch budget per slot (and, for the two-stream pipeline, at least ~8 sub-batches per
    // call, but none so small that it cannot fill the GPU)
    int64_t total_words_all = 0;
    for (int w = 0; w < n_win; ++w) total_words_all += ((hi[w] - lo[w] + 31) / 32 + grp - 1) / grp * grp;
    // A job that fits one batch runs as one batch on one stream: splitting it only to overlap the pack kernel with the pair
    // kernels is slower (measured: C2 1.44 vs 0.99 ms).  A job that needs several batches is cut at the scratch limit and its
    // sub-batches follow each other on the one stream: since the pair kernels run on the matrix cores, the pack kernel beside
    // them on a second stream costs more CU time than it hides (north-star shape 11.2 - 12.1 against 9.8 - 11.0 ms; one rank's
    // share of config 5, 150 GB: 40.1 ms pipelined).  PG_OVERLAP=1 brings the two-stream pipeline back (at least 8 sub-batches,
    // all but the first pack kernel beside the pair kernels of the sub-batch before).
    // (one batch uses one slot: it may take the whole scratch budget; sub-batches alternate between the two slots)
    const bool multi = total_words_all * word_bytes + (int64_t)n_win * mat_bytes > c->scratch_limit;
    const bool two_streams = getenv("PG_OVERLAP") != nullptr;
    const int n_sub = two_streams && atoi(getenv("PG_OVERLAP")) >= 2 ? atoi(getenv("PG_OVERLAP")) : 8;      // (PG_OVERLAP=n: n sub-batches)
    const bool split = multi || two_streams;
    int64_t target_words = two_streams ? std::max<int64_t>(total_words_all / n_sub, 32768) : total_words_all;
    for (int k = 0; k < 2; ++k) {
        if (!c->slot[k].packed) HIPCHK(hipEventCreateWithFlags(&c->slot[k].packed, hipEventDisableTiming));
        if (!c->slot[k].consumed) HIPCHK(hipEventCreateWithFlags(&c->slot[k].consumed, hipEventDisableTiming));
        c->slot[k].used = false;
    }
    int w0 = 0, bi = 0;
    while (w0 < n_win) {
        int64_t words = 0;
        int w1 = w0;
        while (w1 < n_win) {
            int64_t wlen = ((hi[w1] - lo[w1] + 31) / 32 + grp - 1) / grp * grp;
            int64_t nbytes = (words + wlen) * word_bytes + (int64_t)(w1 - w0 + 1) * mat_bytes;
            if (w1 > w0 && (nbytes > (split ? c->scratch_limit / 2 : c->scratch_limit) || words + wlen > target_words)) break;
            words += wlen;
            ++w1;
            if (w1 - w0 >= 65535) break;                      // gridDim.y limit
        }
        const int nb = w1 - w0;
        pg_ctx::Slot &sl = c->slot[bi & 1];
        int rc;
        // one batch, or sub-batches one after the other: the pack kernel goes on the same stream as its consumers
        // (no cross-stream event hand-over)
        const bool single = (w0 == 0 && w1 == n_win) || !two_streams;
        hipStream_t ps = single ? c->stream : c->stream2;
        // the slot's previous occupant (sub-batch bi-2) must be fully consumed before its planes are overwritten, and
        // its staging vector must have been copied before it is rebuilt
        if (sl.used) {
            HIPCHK(hipStreamWaitEvent(ps, sl.consumed, 0));
            HIPCHK(hipEventSynchronize(sl.packed));
        }
        // stage [lo | hi | goff(n+1) | vgoff(n+1) | nw]: nw = int32 word counters of k_pack2, zero per window (and, in the
        // presence-pre-pass mode, one slot per group for k_word_scan) -- they ride in the same copy instead of a memset
        int64_t ga_pre = 0;
        for (int k = 0; k < nb; ++k) ga_pre += ((hi[w0 + k] - lo[w0 + k] + 31) / 32 + grp - 1) / grp;
        const size_t n_nw = (size_t)nb + (pg_pack_needs_presence(NP) ? (size_t)ga_pre : 0);
        const size_t h_len = 4 * (size_t)nb + 2 + (n_nw + 1) / 2;
        if ((rc = sl.host.ensure(h_len)) != PG_OK) return rc;
        int64_t *h = sl.host.p;
        int64_t ga = 0, va = 0;
        int max_groups = 0;
        for (int k = 0; k < nb; ++k) {
            h[k] = lo[w0 + k];
            h[nb + k] = hi[w0 + k];
            const int64_t wds = (hi[w0 + k] - lo[w0 + k] + 31) / 32;
            h[2 * (size_t)nb + k] = ga;
            h[3 * (size_t)nb + 1 + k] = va;
            const int64_t groups = (wds + grp - 1) / grp;
            ga += groups;
            va += (wds + 3) / 4;
            max_groups = (int)std::max<int64_t>(max_groups, groups);
        }
        h[3 * (size_t)nb] = ga;
        h[4 * (size_t)nb + 1] = va;
        memset(h + 4 * (size_t)nb + 2, 0, ((n_nw + 1) / 2) * 8);
        if ((rc = sl.win.upload(h, h_len, ps)) != PG_OK) return rc;
        const int64_t *d_lo = sl.win.p, *d_hi = sl.win.p + nb, *d_goff = sl.win.p + 2 * (size_t)nb,
                      *d_vgoff = sl.win.p + 3 * (size_t)nb + 1;
        int32_t *d_nw = reinterpret_cast<int32_t *>(sl.win.p + 4 * (size_t)nb + 2);
        // + 4 word groups / words: the last stage (look-ahead load) of the pair kernels reads up to three past a part's range
        if ((rc = sl.Vp.ensure((size_t)(std::max<int64_t>(va, 1) + 4) * NPv * 4)) != PG_OK) return rc;
        if ((rc = sl.XV.ensure(((size_t)std::max<int64_t>(ga, 1) * capg + 4) * PG_XV_PLANES * NP)) != PG_OK) return rc;
        if ((rc = c->Cmat.ensure((size_t)nb * n_units * n_units)) != PG_OK) return rc;
        if ((rc = c->Dmat.ensure((size_t)nb * N * N)) != PG_OK) return rc;
        hipEvent_t e0, e1;
        // ---- stream2: pack ----
        const bool time_pack = (c->time_mask >> PG_K_PACK) & 1u;
        if (time_pack) {
            if ((rc = event_get(c, &e0)) != PG_OK) return rc;
            if ((rc = event_get(c, &e1)) != PG_OK) return rc;
            HIPCHK(hipEventRecord(e0, ps));
        }
        if (pg_pack_needs_presence(NP) && (rc = sl.pres.ensure((size_t)std::max<int64_t>(ga, 1) * grp * 4)) != PG_OK) return rc;
        pg_launch_pack2(ps, c->gt.p, c->S, d_lo, d_hi, d_goff, d_vgoff, nb, max_groups, ga, sl.Vp.p, NPv, sl.XV.p, NP,
                        d_nw, dip ? 1 : 0, c->flag.p, sl.pres.p, capg, grp);
        if (time_pack) {
            HIPCHK(hipEventRecord(e1, ps));
            c->events[PG_K_PACK].push_back(std::make_pair(e0, e1));
            c->acc_launches[PG_K_PACK] += 1;
        }
        HIPCHK(hipEventRecord(sl.packed, ps));
        // ---- stream: pair kernels + consume ----
        if (!single) HIPCHK(hipStreamWaitEvent(c->stream, sl.packed, 0));
        if ((rc = pg_time_begin(c, PG_K_PAIRWISE, &e0, &e1)) != PG_OK) return rc;
        // the pair counts run on the matrix cores (exact products of the bit planes, pg_pair_mfma.hip: MX fp4, or int8 with
        // PG_PAIR_I8=1); PG_PAIR_VALU=1 keeps the popcount kernels (A/B runs, tests)
        const bool valu_pairs = getenv("PG_PAIR_VALU") != nullptr;
        if (valu_pairs && NP % 64) return pg_fail(PG_ERR_STATE, "PG_PAIR_VALU must be set before pg_set_samples (plane stride %d)", NP);
        if (!valu_pairs && pg_pair_big_fits(NPv, n_units)) {
            pg_launch_pairC_big(c->stream, sl.Vp.p, d_vgoff, nb, NPv, n_units, dip ? 1 : 0, va / nb, (int64_t)max_groups * grp * 32, c->Cmat.p);
        } else if (!valu_pairs && pg_pair_tile_fits(NPv)) {
            if (pg_launch_pairC_tile(c->stream, sl.Vp.p, d_vgoff, nb, NPv, n_units, dip ? 1 : 0, va / nb, (int64_t)max_groups * grp * 32, c->Cmat.p))
                return pg_fail(PG_ERR_HIP, "pair-kernel program upload failed");
        } else if (!valu_pairs) pg_launch_pairC_mfma(c->stream, sl.Vp.p, d_vgoff, nb, NPv, n_units, dip ? 1 : 0, va / nb, (int64_t)max_groups * grp * 32, c->Cmat.p);
        else if (dip) pg_launch_pairC(c->stream, sl.Vp.p, d_vgoff, nb, c->tasksC.p, c->n_tasksC, NPv, n_units, 1, va / nb, c->Cmat.p);
        else pg_launch_pairC(c->stream, sl.Vp.p, d_vgoff, nb, c->tasksCh.p, c->n_tasksCh, NPv, n_units, 0, va / nb, c->Cmat.p);
        if ((rc = pg_time_end(c, PG_K_PAIRWISE, e0, e1, 1)) != PG_OK) return rc;
        // (running k_pairD beside k_pairC on a third stream was measured: +3 % throughput, but overlapping kernels make the
        // per-kernel timings ambiguous; kept sequential)
        if ((rc = pg_time_begin(c, PG_K_PAIRD, &e0, &e1)) != PG_OK) return rc;
        if (!valu_pairs) pg_launch_pairD_mfma(c->stream, sl.XV.p, d_nw, d_goff, nb, NP, N, ga / nb * grp / 10, (int64_t)max_groups * capg * 32, c->Dmat.p, capg);
        else pg_launch_pairD(c->stream, sl.XV.p, d_nw, d_goff, nb, c->tasks2.p, c->n_tasks2, NP, N, ga / nb, c->Dmat.p, capg);
        if ((rc = pg_time_end(c, PG_K_PAIRD, e0, e1, 1)) != PG_OK) return rc;
        HIPCHK(hipGetLastError());
        c->cur_win_lo = d_lo;
        c->cur_win_hi = d_hi;
        if ((rc = consume(w0, nb)) != PG_OK) return rc;
        HIPCHK(hipEventRecord(sl.consumed, c->stream));
        sl.used = true;
        w0 = w1;
        ++bi;
    }
    // leave both streams quiescent with respect to each other: later work on ctx->stream must see stream2's writes done
    for (int k = 0; k < 2; ++k)
        if (c->slot[k].used) HIPCHK(hipEventSynchronize(c->slot[k].packed));
    return PG_OK;
}

// End-of-call wait on ctx->stream.  The result table is a few kilobytes and the caller's next call follows immediately, so the
// wake-up latency of a blocking wait (tens of microseconds) is a visible fraction of a millisecond-scale pass: poll the stream
// for a short while first (PG_SPIN_US microseconds, default 2000; 0 = block at once).
static int stream_wait(pg_ctx *c) {
    static const long spin_us = getenv("PG_SPIN_US") ? atol(getenv("PG_SPIN_US")) : 2000;
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) return PG_OK;
            if (e != hipErrorNotReady) return pg_fail(PG_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

// The diploid-shortcut flag (raised by k_pack2) is zero whenever no call is in flight: zeroed when allocated and again, on
// ctx->stream, right after each read; every entry point synchronises ctx->stream before it returns.
static int flag_ready(pg_ctx *c) {
    if (c->flag.p) return PG_OK;
    int rc = c->flag.ensure(1);
    if (rc != PG_OK) return rc;
    HIPCHK(hipMemsetAsync(c->flag.p, 0, 4, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

// Diploid fast path first (called counts per individual); if any window turns out to hold an individual whose two
// haplotypes differ in calledness (e.g. phased `A|N`), everything is recomputed with per-haplotype called counts.  Likewise a
// window with more virtual sites than the default reservation (PG_XV_CAP_DEFAULT words per group) makes the call start over
// with the worst-case reservation.
static int note_flags(pg_ctx *c, int flag, bool *dip, bool *again) {
    *again = false;
    if (flag & PG_FLAG_XV_OVERFLOW) {
        if (c->xv_worst) return pg_fail(PG_ERR_STATE, "XV overflow with the worst-case reservation");
        c->xv_worst = true;
        for (int k = 0; k < 2; ++k) c->slot[k].XV.release();
        *again = true;
    }
    if (*dip && (flag & PG_FLAG_MISMATCH)) {
        *dip = false;
        *again = true;
    }
    return PG_OK;
}

template <class F>
static int pairwise_run(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, F consume) {
    bool dip = c->all_diploid && getenv("PG_NO_DIP") == nullptr;
    int rc;
    if ((rc = flag_ready(c)) != PG_OK) return rc;
    for (;;) {
        if ((rc = pairwise_batches(c, lo, hi, n_win, dip, consume)) != PG_OK) return rc;
        int32_t flag = 0;
        HIPCHK(hipMemcpyAsync(&flag, c->flag.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemsetAsync(c->flag.p, 0, 4, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        bool again;
        if ((rc = note_flags(c, flag, &dip, &again)) != PG_OK) return rc;
        if (!again) return PG_OK;
    }
}

// ---- placement of the resident rows ---------------------------------------------------------------------------------------------
// The pack kernel streams the resident rows once per pass and its time moves by +-6 % with the PHYSICAL pages behind them: the same
// virtual address, released and allocated again, gives 7.7 ... 8.7 ms on the north-star shape, stable for the life of the
// allocation (tools/pack_variance.py, tools/placement_probe.py; a read-only stream over the same rows moves by 2 %).  The kernel's
// time on the freshly zeroed buffer predicts its time with the data in place (probe 7.4-7.5 -> 7.8-7.9 ms, probe 8.0-8.4 -> 8.3-8.6),
// so a large reservation is tried a few times -- the candidates are held together, so they are different pages -- and the one on
// which the probe pass (the regular pack + pair path over 50 000-site windows of the empty rows) is fastest is kept.
extern "C" int pg_reserve_sites_tuned(pg_ctx *c, int64_t n_sites, int max_trials, double *probe_ms_out, int *n_trials_out,
                                      int *chosen_out) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (n_trials_out) *n_trials_out = 0;
    if (chosen_out) *chosen_out = -1;
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (n_sites < 0) return pg_fail(PG_ERR_ARG, "n_sites < 0");
    HIPCHK(hipSetDevice(c->device));
    if (n_sites <= c->cap_sites) return PG_OK;
    const size_t bytes = (size_t)(n_sites + 32) * c->S;
    if (max_trials > 8) max_trials = 8;
    int trials = 1;
    if (c->n_hap <= 4096 && bytes >= ((size_t)4 << 30) && max_trials > 1) {
        HIPCHK(hipStreamSynchronize(c->stream_up));
        c->up_pending = false;
        c->gt.release();                                   // (as in pg_reserve_sites: growing drops the rows)
        c->cap_sites = 0;
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t keep = (size_t)c->scratch_limit + ((size_t)8 << 30);      // room for the probe's (and the job's) scratch
        const size_t fit = free_b > keep ? (free_b - keep) / bytes : 0;
        trials = (int)std::min<size_t>((size_t)max_trials, fit);
    }
    if (trials < 2) return pg_reserve_sites(c, n_sites);
    const int64_t wind = 50000;
    const int n_win = (int)std::max<int64_t>(1, n_sites / wind);
    std::vector<int64_t> lo((size_t)n_win), hi((size_t)n_win);
    for (int w = 0; w < n_win; ++w) { lo[(size_t)w] = w * wind; hi[(size_t)w] = std::min<int64_t>(n_sites, (w + 1) * wind); }
    std::vector<DevBuf<int8_t>> cand((size_t)trials);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PG_OK, best = -1, tried = 0;
    double best_ms = 0.0;
    auto cleanup = [&](int keep_idx) {
        for (int t = 0; t < trials; ++t)
            if (t != keep_idx) cand[(size_t)t].release();
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cleanup(-1); return pg_fail(PG_ERR_HIP, "hipEventCreate"); }
    const uint32_t saved_mask = c->time_mask;
    c->time_mask = 0;                                      // the probe's launches are not the caller's statistics
    for (int t = 0; t < trials && rc == PG_OK; ++t) {
        if (cand[(size_t)t].alloc(bytes) != PG_OK) break;  // out of memory: make do with the candidates so far
        ++tried;
        if (hipMemsetAsync(cand[(size_t)t].p, 0, bytes, c->stream) != hipSuccess) { rc = pg_fail(PG_ERR_HIP, "hipMemsetAsync"); break; }
        c->gt = cand[(size_t)t];                           // (plain pointers: ownership stays with cand[] until the choice is made)
        c->cap_sites = n_sites;
        float ms = 0.0f;
        for (int pass = 0; pass < 2 && rc == PG_OK; ++pass) {                 // the first pass also allocates the scratch
            if (hipEventRecord(e0, c->stream) != hipSuccess) { rc = pg_fail(PG_ERR_HIP, "hipEventRecord"); break; }
            rc = pairwise_run(c, lo.data(), hi.data(), n_win, [](int, int) -> int { return PG_OK; });
            if (rc != PG_OK) break;
            if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = pg_fail(PG_ERR_HIP, "probe timing"); break; }
        }
        if (rc != PG_OK) break;
        if (probe_ms_out) probe_ms_out[t] = ms;
        if (best < 0 || ms < best_ms) { best = t; best_ms = ms; }
    }
    c->time_mask = saved_mask;
    c->gt = DevBuf<int8_t>();
    c->cap_sites = 0;
    if (rc != PG_OK || best < 0) {
        cleanup(-1);
        return rc != PG_OK ? rc : pg_reserve_sites(c, n_sites);
    }
    cleanup(best);
    c->gt = cand[(size_t)best];
    c->cap_sites = n_sites;
    if (n_trials_out) *n_trials_out = tried;
    if (chosen_out) *chosen_out = best;
    return PG_OK;
}

// ---- placement of the planes ------------------------------------------------------------------------------------------------------
// The other half of the pair: with the rows where they are, the pack kernel's time still moves by up to 4 % with the pages behind the
// planes it writes (tools/pack_placement2.py: 7.81 -> 7.54, 7.55 -> 7.87 ms when only the planes were allocated again).  Planes are
// cheap to try -- a few GB, nothing to fill: up to max_trials (<= 8) sets of them are held together, the regular pack + pair path over
// windows of window_sites (<= 0: 50 000) of the resident rows (empty or not: the rows are only read) is timed on each, the fastest set
// stays in the slots.  The choice made on empty rows carries over to the filled ones only in part (tools/plane_placement.py: 1 % against
// 1.5 - 3 % for a choice made on the rows as they are), so a caller with a long-lived data set calls this once the rows are loaded.  Candidate 0 = the planes the context holds when called.  Later passes of that shape or a smaller one keep the set (buffers only grow).
extern "C" int pg_tune_planes(pg_ctx *c, int64_t n_sites, int64_t window_sites, int max_trials, double *probe_ms_out, int *n_trials_out,
                              int *chosen_out) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (n_trials_out) *n_trials_out = 0;
    if (chosen_out) *chosen_out = -1;
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (n_sites < 1 || n_sites > c->cap_sites) return pg_fail(PG_ERR_ARG, "pg_tune_planes: n_sites must lie inside the reservation");
    HIPCHK(hipSetDevice(c->device));
    const int trials = std::min(max_trials, 8);
    if (trials < 2) return PG_OK;
    const int64_t wind = window_sites > 0 ? window_sites : 50000;
    if ((n_sites + wind - 1) / wind > (1 << 24)) return pg_fail(PG_ERR_ARG, "pg_tune_planes: more than 2^24 windows");
    const int n_win = (int)((n_sites + wind - 1) / wind);
    std::vector<int64_t> lo((size_t)n_win), hi((size_t)n_win);
    for (int w = 0; w < n_win; ++w) { lo[(size_t)w] = w * wind; hi[(size_t)w] = std::min<int64_t>(n_sites, (w + 1) * wind); }
    struct Set { DevBuf<uint32_t> Vp[2], XV[2]; };
    std::vector<Set> cand((size_t)trials);
    auto take = [&](Set &s) {                                 // the slots' planes -> s (plain pointers; the slots are left without)
        for (int k = 0; k < 2; ++k) {
            s.Vp[k] = c->slot[k].Vp;
            s.XV[k] = c->slot[k].XV;
            c->slot[k].Vp = DevBuf<uint32_t>();
            c->slot[k].XV = DevBuf<uint32_t>();
        }
    };
    auto give = [&](Set &s) {
        for (int k = 0; k < 2; ++k) {
            c->slot[k].Vp = s.Vp[k];
            c->slot[k].XV = s.XV[k];
            s.Vp[k] = DevBuf<uint32_t>();
            s.XV[k] = DevBuf<uint32_t>();
        }
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return pg_fail(PG_ERR_HIP, "hipEventCreate");
    }
    const uint32_t saved_mask = c->time_mask;
    c->time_mask = 0;                                         // the probe's launches are not the caller's statistics
    int rc = PG_OK, best = -1, tried = 0;
    double best_ms = 0.0;
    size_t set_bytes = 0;
    for (int t = 0; t < trials && rc == PG_OK; ++t) {
        // (t > 0: the slots hold no planes, pairwise_run's ensure() allocates a fresh set beside the ones held in cand[])
        if (t > 0) {                                          // room for one more set, twice over, and 8 GiB to spare -- or stop here
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * set_bytes + ((size_t)8 << 30)) break;
        }
        float ms = 0.0f, fastest = 0.0f;
        for (int pass = 0; pass < 3 && rc == PG_OK; ++pass) {                 // the first pass allocates; the faster of the next two counts
            if (hipEventRecord(e0, c->stream) != hipSuccess) { rc = pg_fail(PG_ERR_HIP, "hipEventRecord"); break; }
            rc = pairwise_run(c, lo.data(), hi.data(), n_win, [](int, int) -> int { return PG_OK; });
            if (rc != PG_OK) break;
            if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = pg_fail(PG_ERR_HIP, "probe timing"); break; }
            if (pass == 1 || (pass == 2 && ms < fastest)) fastest = ms;
        }
        if (rc != PG_OK) {
            if (t > 0 && rc == PG_ERR_HIP) {               // out of memory: make do with the candidates so far
                rc = PG_OK;
                (void)hipGetLastError();
                Set failed;
                take(failed);
                for (int k = 0; k < 2; ++k) { failed.Vp[k].release(); failed.XV[k].release(); }
            }
            break;
        }
        take(cand[(size_t)t]);
        for (int k = 0; t == 0 && k < 2; ++k) set_bytes += (cand[0].Vp[k].cap + cand[0].XV[k].cap) * sizeof(uint32_t);
        ++tried;
        if (probe_ms_out) probe_ms_out[t] = fastest;
        if (best < 0 || fastest < best_ms) { best = t; best_ms = fastest; }
    }
    c->time_mask = saved_mask;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream2);
    if (best >= 0) give(cand[(size_t)best]);
    for (int t = 0; t < trials; ++t)
        for (int k = 0; k < 2; ++k) { cand[(size_t)t].Vp[k].release(); cand[(size_t)t].XV[k].release(); }
    for (int k = 0; k < 2; ++k) c->slot[k].used = false;      // (everything was waited for above)
    if (rc != PG_OK) return rc;
    if (n_trials_out) *n_trials_out = tried;
    if (chosen_out) *chosen_out = best;
    return PG_OK;
}

extern "C" int pg_pairwise(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int32_t *D_out, int32_t *C_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    if (n_win > 0 && (!D_out || !C_out)) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    const size_t NN = (size_t)c->n_hap * c->n_hap;
    rc = pairwise_run(c, lo, hi, n_win, [&](int w0, int nb) -> int {
        int r;
        if ((r = c->Cfull.ensure((size_t)nb * NN)) != PG_OK) return r;
        if ((r = c->Dfull.ensure((size_t)nb * NN)) != PG_OK) return r;
        pg_launch_expand(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->cN, c->cshift, nb, c->Cfull.p, c->Dfull.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(C_out + (size_t)w0 * NN, c->Cfull.p, nb * NN * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(D_out + (size_t)w0 * NN, c->Dfull.p, nb * NN * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return PG_OK;
    });
    return rc;
}

extern "C" int pg_popdist(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int min_pair_sites,
                          double *sum_out, int64_t *cnt_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    if (c->n_pops < 1) return pg_fail(PG_ERR_STATE, "pg_popdist needs at least one population");
    if (n_win > 0 && (!sum_out || !cnt_out)) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    const int npairs = c->n_pops * (c->n_pops + 1) / 2;
    if ((rc = c->res_f64.ensure((size_t)std::max(n_win, 1) * npairs)) != PG_OK) return rc;
    if ((rc = c->res_i64.ensure((size_t)std::max(n_win, 1) * npairs)) != PG_OK) return rc;
    rc = pairwise_run(c, lo, hi, n_win, [&](int w0, int nb) -> int {
        hipEvent_t e0, e1;
        int r = pg_time_begin(c, PG_K_POPDIST_FIN, &e0, &e1);
        if (r != PG_OK) return r;
        pg_launch_popdist_fin(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->cN, c->cshift, nb, c->pop_start.p, c->n_pops, min_pair_sites,
                              c->res_f64.p + (size_t)w0 * npairs, c->res_i64.p + (size_t)w0 * npairs, c->all_diploid && c->pops_on_individuals ? 1 : 0);
        if ((r = pg_time_end(c, PG_K_POPDIST_FIN, e0, e1, 1)) != PG_OK) return r;
        HIPCHK(hipGetLastError());
        return PG_OK;
    });
    if (rc != PG_OK) return rc;
    if (n_win > 0) {
        HIPCHK(hipMemcpyAsync(sum_out, c->res_f64.p, (size_t)n_win * npairs * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(cnt_out, c->res_i64.p, (size_t)n_win * npairs * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

extern "C" int pg_set_reference_order(pg_ctx *c, const int32_t *pop_row_order, const int32_t *pop_name_rank) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    HIPCHK(hipSetDevice(c->device));
    const int n_in = c->n_pops > 0 ? c->h_pop_start[c->n_pops] : 0;
    int rc;
    if (pop_row_order && n_in > 0) {
        for (int p = 0; p < c->n_pops; ++p) {                        // a permutation of the population's own slots
            std::vector<char> seen(c->h_pop_start[p + 1] - c->h_pop_start[p], 0);
            for (int k = c->h_pop_start[p]; k < c->h_pop_start[p + 1]; ++k) {
                const int sl = pop_row_order[k] - c->h_pop_start[p];
                if (sl < 0 || sl >= (int)seen.size() || seen[sl]) return pg_fail(PG_ERR_ARG, "pop_row_order is not a permutation of the slots of population %d", p);
                seen[sl] = 1;
            }
        }
        if ((rc = c->ref_row.upload(pop_row_order, (size_t)n_in, c->stream)) != PG_OK) return rc;
    }
    if (pop_name_rank && c->n_pops > 0) {
        std::vector<char> seen(c->n_pops, 0);
        for (int p = 0; p < c->n_pops; ++p) {
            if (pop_name_rank[p] < 0 || pop_name_rank[p] >= c->n_pops || seen[pop_name_rank[p]]) return pg_fail(PG_ERR_ARG, "pop_name_rank is not a permutation");
            seen[pop_name_rank[p]] = 1;
        }
        if ((rc = c->pop_rank.upload(pop_name_rank, (size_t)c->n_pops, c->stream)) != PG_OK) return rc;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

extern "C" int pg_set_sum_order(pg_ctx *c, int mode) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (mode < 0 || mode > 2) return pg_fail(PG_ERR_ARG, "pg_set_sum_order: mode %d (0 = by window length, 1 = NumPy's order, 2 = fixed trees)", mode);
    c->sum_order = mode;
    return PG_OK;
}

// windows of up to this many sites take NumPy's summation order (env: an A/B switch of the tests; else the context's mode)
static long long np_sites_limit(pg_ctx *c, const char *env_name) {
    const char *force = getenv(env_name);
    const int mode = force ? (atoi(force) != 0 ? 1 : 2) : c->sum_order;
    return mode == 1 ? (long long)INT64_MAX : mode == 2 ? -1 : PG_NP_MAX_SITES;
}

extern "C" int pg_set_sample_rank(pg_ctx *c, const int32_t *rank) {
    if (!c || !rank) return pg_fail(PG_ERR_ARG, "pg_set_sample_rank: null argument");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = c->samp_rank.upload(rank, (size_t)c->n_samp, c->stream)) != PG_OK) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

// NumPy's pairwise summation of n values as a tree (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum): fewer than 8 values or at
// most 128: a run (k_popdist_np adds it up as NumPy's unrolled loop does); more: the first n/2 rounded down to a multiple of 8, then
// the rest.  blob = [L, n_inner, n_levels, leaf_off[L + 1], node_l[n_inner], node_r[n_inner], level_start[n_levels + 1]]; the inner
// nodes ordered by height, slots: runs 0 .. L - 1, inner node k at L + k.
namespace {
struct NpNode { int l, r, h; };
int np_build(int off, int n, std::vector<int32_t> &leaf_off, std::vector<NpNode> &inner, int *height) {
    if (n <= 128) {
        leaf_off.push_back(off);
        *height = 0;
        return (int)leaf_off.size() - 1;                            // a run: its index
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    int hl, hr;
    const int l = np_build(off, n2, leaf_off, inner, &hl), r = np_build(off + n2, n - n2, leaf_off, inner, &hr);
    inner.push_back(NpNode{l, r, std::max(hl, hr) + 1});
    *height = inner.back().h;
    return -(int)inner.size();                                      // an inner node: -(index + 1)
}
std::vector<int32_t> np_tree(int n) {
    std::vector<int32_t> leaf_off;
    std::vector<NpNode> inner;
    int h = 0;
    // np.add.reduce hands the flattened block to the pairwise sum in pieces of the ufunc buffer size (np.getbufsize(): 8192
    // values) and adds the pieces' sums one after the other
    int node = np_build(0, std::min(n, 8192), leaf_off, inner, &h);
    for (int at = 8192; at < n; at += 8192) {
        int hc;
        const int piece = np_build(at, std::min(8192, n - at), leaf_off, inner, &hc);
        h = std::max(h, hc) + 1;
        inner.push_back(NpNode{node, piece, h});
        node = -(int)inner.size();
    }
    leaf_off.push_back(n);
    const int L = (int)leaf_off.size() - 1, ni = (int)inner.size();
    std::vector<int> order(ni), where(ni);
    for (int k = 0; k < ni; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return inner[a].h < inner[b].h; });
    for (int k = 0; k < ni; ++k) where[order[k]] = k;
    auto slot = [&](int id) { return id >= 0 ? id : L + where[-id - 1]; };
    std::vector<int32_t> blob;
    blob.push_back(L); blob.push_back(ni); blob.push_back(h);
    blob.insert(blob.end(), leaf_off.begin(), leaf_off.end());
    for (int k = 0; k < ni; ++k) blob.push_back(slot(inner[order[k]].l));
    for (int k = 0; k < ni; ++k) blob.push_back(slot(inner[order[k]].r));
    int at = 0;
    for (int lv = 1; lv <= h; ++lv) {
        blob.push_back(at);
        while (at < ni && inner[order[at]].h == lv) ++at;
    }
    blob.push_back(at);
    return blob;
}
}  // namespace

// the tree as k_popdist_np walks it, for callers without a GPU (tests/test_host.py adds values up along it and compares with NumPy)
extern "C" int pg_np_tree(int n, int32_t *out, int64_t cap, int64_t *len) {
    if (n < 0 || !len) return pg_fail(PG_ERR_ARG, "pg_np_tree: bad argument");
    const std::vector<int32_t> b = np_tree(n);
    *len = (int64_t)b.size();
    if (out && cap >= (int64_t)b.size()) memcpy(out, b.data(), b.size() * sizeof(int32_t));
    return PG_OK;
}

#define PG_NP_MAX_LEAVES 1024      // LDS of a block: 2 * 1024 + 8 * 128 doubles + the tree's tables (3 * 1024 ints) + row maps: 40 KB

// the trees of this context's blocks: (x, x), and (x, y) / (x + y, x + y) for every pair: their lengths do not depend on the orientation
static int np_prepare(pg_ctx *c) {
    if (c->np_state != 0) return PG_OK;
    const int P = c->n_pops;
    std::vector<int> lens;
    int side = 2;
    for (int p = 0; p < P; ++p) {
        const int n = c->h_pop_start[p + 1] - c->h_pop_start[p];
        if (n > 4096) { c->np_state = -1; return PG_OK; }
        lens.push_back(n * n);
        side = std::max(side, 2 * n);
    }
    for (int x = 0; x < P; ++x)
        for (int y = x + 1; y < P; ++y) {
            const int64_t nx = c->h_pop_start[x + 1] - c->h_pop_start[x], ny = c->h_pop_start[y + 1] - c->h_pop_start[y];
            if ((nx + ny) * (nx + ny) > (int64_t)PG_NP_MAX_LEAVES * 128) { c->np_state = -1; return PG_OK; }
            lens.push_back((int)(nx * ny));
            lens.push_back((int)((nx + ny) * (nx + ny)));
            side = std::max(side, (int)(2 * (nx + ny)));
        }
    for (int n : lens) if ((int64_t)n > (int64_t)PG_NP_MAX_LEAVES * 128) { c->np_state = -1; return PG_OK; }
    std::vector<int32_t> all, task_tree;
    std::vector<std::pair<int, int>> known;                            // (n, offset)
    int max_leaves = 1;
    for (int n : lens) {
        int off = -1;
        for (auto &kv : known) if (kv.first == n) off = kv.second;
        if (off < 0) {
            std::vector<int32_t> b = np_tree(n);
            if (b[0] > PG_NP_MAX_LEAVES) { c->np_state = -1; return PG_OK; }
            max_leaves = std::max(max_leaves, (int)b[0]);
            off = (int)all.size();
            all.insert(all.end(), b.begin(), b.end());
            known.push_back(std::make_pair(n, off));
        }
        task_tree.push_back(off);
    }
    int rc;
    if ((rc = c->np_trees.upload(all.data(), all.size(), c->stream)) != PG_OK) return rc;
    if ((rc = c->np_task_tree.upload(task_tree.data(), task_tree.size(), c->stream)) != PG_OK) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->np_max_leaves = max_leaves;
    c->np_max_side = side;
    c->np_state = 1;
    return PG_OK;
}

extern "C" int pg_popdist_stats(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int min_pair_sites, double min_data,
                                int do_pairs, double *stats_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    if (c->n_pops < 1) return pg_fail(PG_ERR_STATE, "pg_popdist_stats needs at least one population");
    if (n_win == 0) return PG_OK;
    if (!stats_out) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    // the sums in NumPy's order (k_popdist_np) wherever the blocks' trees fit a thread block's LDS: populations of up to a few
    // hundred haplotypes; PG_POPDIST_TREE=0: the older finisher (upper triangles, a fixed reduction tree: equal within 1e-15)
    if ((rc = np_prepare(c)) != PG_OK) return rc;
    // NumPy's order where the last bit can show: windows of up to PG_NP_MAX_SITES sites (quotients of small integers sit on rounding
    // ties of the printed digit; Fst of equal populations is +-0.0).  Longer windows: the older finisher (upper triangles, a fixed
    // tree; 20 times less work: every quotient is formed once instead of 2 .. 6 times), equal within 1e-15 -- a printed difference
    // would need a mean within 1e-16 of a tie.  The choice is made window by window (a window's numbers do not depend on what it is
    // batched with).  PG_POPDIST_TREE=1 / 0 forces one or the other.
    const long long np_upto = c->np_state != 1 ? -1 : np_sites_limit(c, "PG_POPDIST_TREE");
    const int P = c->n_pops, npairs = P * (P + 1) / 2, ncols = P + (do_pairs ? P * (P - 1) : 0);
    if ((rc = c->res_f64.ensure((size_t)n_win * P * P)) != PG_OK) return rc;           // k_popdist_np: P^2 sums a window
    if ((rc = c->res_i64.ensure((size_t)n_win * P * P)) != PG_OK) return rc;
    if ((rc = c->part_f64.ensure((size_t)n_win * npairs)) != PG_OK) return rc;         // k_popdist_fin: P (P + 1) / 2
    if ((rc = c->part_i64.ensure((size_t)n_win * npairs)) != PG_OK) return rc;
    if ((rc = c->stats.ensure((size_t)n_win * ncols + 1)) != PG_OK) return rc;
    if ((rc = flag_ready(c)) != PG_OK) return rc;
    if (c->events[PG_K_PACK].size() > 4096 && (rc = fold_events(c)) != PG_OK) return rc;
    auto consume = [&](int w0, int nb) -> int {
        bool any_short = false, any_long = false;
        for (int w = w0; w < w0 + nb; ++w) (hi[w] - lo[w] <= np_upto ? any_short : any_long) = true;
        hipEvent_t e0, e1;
        int r = pg_time_begin(c, PG_K_POPDIST_FIN, &e0, &e1);
        if (r != PG_OK) return r;
        if (any_short)
            pg_launch_popdist_np(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->cN, c->cshift, nb, c->pop_start.p, P, c->ref_row.p,
                                 c->pop_rank.p, c->np_task_tree.p, c->np_trees.p, c->np_max_leaves, c->np_max_side, min_pair_sites, min_data, do_pairs,
                                 c->res_f64.p + (size_t)w0 * P * P, c->res_i64.p + (size_t)w0 * P * P, c->stats.p + (size_t)w0 * ncols,
                                 c->cur_win_lo, c->cur_win_hi, np_upto);
        if (any_long) {
            pg_launch_popdist_fin(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->cN, c->cshift, nb, c->pop_start.p, P, min_pair_sites,
                                  c->part_f64.p + (size_t)w0 * npairs, c->part_i64.p + (size_t)w0 * npairs,
                                  c->all_diploid && c->pops_on_individuals ? 1 : 0, c->cur_win_lo, c->cur_win_hi, np_upto);
            pg_launch_popstats(c->stream, c->part_f64.p + (size_t)w0 * npairs, c->part_i64.p + (size_t)w0 * npairs, nb, c->pop_start.p, P,
                               min_data, do_pairs, c->stats.p + (size_t)w0 * ncols, c->cur_win_lo, c->cur_win_hi, np_upto);
        }
        if ((r = pg_time_end(c, PG_K_POPDIST_FIN, e0, e1, 1)) != PG_OK) return r;
        HIPCHK(hipGetLastError());
        return PG_OK;
    };
    bool dip = c->all_diploid && getenv("PG_NO_DIP") == nullptr;
    const size_t n_out = (size_t)n_win * ncols;
    if ((rc = c->out_pin.ensure(n_out + 1)) != PG_OK) return rc;
    // one device-to-host copy (into pinned memory) and one synchronisation per pass: the flag word of the pack kernels (diploid
    // shortcut verdict, XV overflow) travels in the slot after the table (k_flag_export also re-arms the flag)
    for (;;) {
        if ((rc = pairwise_batches(c, lo, hi, n_win, dip, consume)) != PG_OK) return rc;
        pg_launch_flag_export(c->stream, c->flag.p, c->stats.p + n_out);
        HIPCHK(hipMemcpyAsync(c->out_pin.p, c->stats.p, (n_out + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        if ((rc = stream_wait(c)) != PG_OK) return rc;
        bool again;
        if ((rc = note_flags(c, (int)c->out_pin.p[n_out], &dip, &again)) != PG_OK) return rc;
        if (again) continue;
        memcpy(stats_out, c->out_pin.p, n_out * 8);
        return PG_OK;
    }
}

static int indpair_run(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int min_pair_sites, int mean_mode,
                       double *sum_out, int64_t *cnt_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    if (n_win > 0 && (!sum_out || (!cnt_out && mean_mode == 0))) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    const size_t npairs = (size_t)c->n_samp * (c->n_samp + 1) / 2;
    // results are copied back per batch (they can be large for distMat-sized inputs)
    // deferred: the table's copy into the caller's page-locked memory runs on a stream of its own while the NEXT call's kernels
    // run; the caller reads the table after pg_results_wait / pg_sync
    bool defer = false;
    if (c->defer_results && mean_mode != 0 && n_win > 0) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, sum_out) == hipSuccess) defer = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    rc = pairwise_run(c, lo, hi, n_win, [&](int w0, int nb) -> int {
        int r;
        if (defer) {
            const int f = c->res_flip;
            c->res_flip ^= 1;
            if (!c->res_stream) HIPCHK(hipStreamCreateWithFlags(&c->res_stream, hipStreamNonBlocking));
            if (!c->res_fin[f]) HIPCHK(hipEventCreateWithFlags(&c->res_fin[f], hipEventDisableTiming));
            const bool had_copy = c->res_copied[f] != nullptr;
            if (!had_copy) HIPCHK(hipEventCreateWithFlags(&c->res_copied[f], hipEventDisableTiming));
            if (c->res_alt[f].cap < (size_t)nb * npairs) {
                if (had_copy) HIPCHK(hipEventSynchronize(c->res_copied[f]));       // (the buffer is replaced: its last copy must be out)
                if ((r = c->res_alt[f].ensure((size_t)nb * npairs)) != PG_OK) return r;
            } else if (had_copy) {
                HIPCHK(hipStreamWaitEvent(c->stream, c->res_copied[f], 0));      // the finaliser overwrites what that copy reads
            }
            hipEvent_t e0, e1;
            if ((r = pg_time_begin(c, PG_K_INDPAIR_FIN, &e0, &e1)) != PG_OK) return r;
            pg_launch_indpair_fin(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->cN, c->cshift, nb, c->samp_start.p, c->samp_rank.p, c->n_samp,
                                  min_pair_sites, c->res_alt[f].p, nullptr, mean_mode);
            if ((r = pg_time_end(c, PG_K_INDPAIR_FIN, e0, e1, 1)) != PG_OK) return r;
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->res_fin[f], c->stream));
            HIPCHK(hipStreamWaitEvent(c->res_stream, c->res_fin[f], 0));
            hipEvent_t t0 = nullptr, t1 = nullptr;
            if ((c->time_mask >> PG_K_RESULT_D2H) & 1u) {
                if ((r = event_get(c, &t0)) != PG_OK || (r = event_get(c, &t1)) != PG_OK) return r;
                HIPCHK(hipEventRecord(t0, c->res_stream));
            }
            HIPCHK(hipMemcpyAsync(sum_out + (size_t)w0 * npairs, c->res_alt[f].p, (size_t)nb * npairs * 8, hipMemcpyDeviceToHost, c->res_stream));
            if (t0) {
                HIPCHK(hipEventRecord(t1, c->res_stream));
                c->events[PG_K_RESULT_D2H].push_back(std::make_pair(t0, t1));
                c->acc_launches[PG_K_RESULT_D2H] += 1;
            }
            HIPCHK(hipEventRecord(c->res_copied[f], c->res_stream));
            HIPCHK(hipStreamSynchronize(c->stream));                             // the kernels are done (the staging buffers are free); the copy goes on
            return PG_OK;
        }
        if ((r = c->res_f64.ensure((size_t)nb * npairs)) != PG_OK) return r;
        if (mean_mode == 0 && (r = c->res_i64.ensure((size_t)nb * npairs)) != PG_OK) return r;
        hipEvent_t e0, e1;
        if ((r = pg_time_begin(c, PG_K_INDPAIR_FIN, &e0, &e1)) != PG_OK) return r;
        pg_launch_indpair_fin(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->cN, c->cshift, nb, c->samp_start.p, c->samp_rank.p, c->n_samp, min_pair_sites,
                              c->res_f64.p, c->res_i64.p, mean_mode);
        if ((r = pg_time_end(c, PG_K_INDPAIR_FIN, e0, e1, 1)) != PG_OK) return r;
        HIPCHK(hipGetLastError());
        // (the table can be tens of megabytes -- 40 MB per distMat pass over 1000 diploids --: PCIe time, timed as its own family)
        if ((r = pg_time_begin(c, PG_K_RESULT_D2H, &e0, &e1)) != PG_OK) return r;
        HIPCHK(hipMemcpyAsync(sum_out + (size_t)w0 * npairs, c->res_f64.p, (size_t)nb * npairs * 8, hipMemcpyDeviceToHost, c->stream));
        if (mean_mode == 0)
            HIPCHK(hipMemcpyAsync(cnt_out + (size_t)w0 * npairs, c->res_i64.p, (size_t)nb * npairs * 8, hipMemcpyDeviceToHost, c->stream));
        if ((r = pg_time_end(c, PG_K_RESULT_D2H, e0, e1, 1)) != PG_OK) return r;
        HIPCHK(hipStreamSynchronize(c->stream));
        return PG_OK;
    });
    return rc;
}

extern "C" int pg_indpairdist(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int min_pair_sites,
                              double *sum_out, int64_t *cnt_out) {
    return indpair_run(c, lo, hi, n_win, min_pair_sites, 0, sum_out, cnt_out);
}

extern "C" int pg_indpairdist_mean(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int min_pair_sites,
                                   int diag_counts_zeros, double *d_out) {
    return indpair_run(c, lo, hi, n_win, min_pair_sites, diag_counts_zeros ? 2 : 1, d_out, nullptr);
}

extern "C" int pg_indpairdist_mean_from_counts(pg_ctx *c, const int32_t *D, const int32_t *C, int n_win, int min_pair_sites,
                                               int diag_counts_zeros, double *d_out) {
    if (!c) return pg_fail(PG_ERR_ARG, "null context");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples has not been called");
    if (n_win < 0) return pg_fail(PG_ERR_ARG, "negative window count");
    if (n_win == 0) return PG_OK;
    if (!D || !C || !d_out) return pg_fail(PG_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    const size_t N = (size_t)c->n_hap, NN = N * N;
    const size_t npairs = (size_t)c->n_samp * (c->n_samp + 1) / 2;
    // haplotype-level counts on both sides (cN = n_hap, no unit shift), a bounded number of windows per trip
    const int per = (int)std::max<size_t>(1, ((size_t)256 << 20) / (NN * 4));
    int rc;
    for (int w0 = 0; w0 < n_win; w0 += per) {
        const int nb = std::min(per, n_win - w0);
        if ((rc = c->Dmat.ensure((size_t)nb * NN)) != PG_OK) return rc;
        if ((rc = c->Cmat.ensure((size_t)nb * NN)) != PG_OK) return rc;
        if ((rc = c->res_f64.ensure((size_t)nb * npairs)) != PG_OK) return rc;
        HIPCHK(hipMemcpyAsync(c->Dmat.p, D + (size_t)w0 * NN, (size_t)nb * NN * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->Cmat.p, C + (size_t)w0 * NN, (size_t)nb * NN * 4, hipMemcpyHostToDevice, c->stream));
        pg_launch_indpair_fin(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->n_hap, 0, nb, c->samp_start.p, c->samp_rank.p, c->n_samp, min_pair_sites,
                              c->res_f64.p, nullptr, diag_counts_zeros ? 2 : 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(d_out + (size_t)w0 * npairs, c->res_f64.p, (size_t)nb * npairs * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return PG_OK;
}

// ---- indHet / hapStats: device finalisers of the pairwise matrices ------------------------------------------
extern "C" int pg_sample_het(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int min_pair_sites, double *het_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    if (n_win == 0) return PG_OK;
    if (!het_out) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    const size_t ns = (size_t)c->n_samp;
    return pairwise_run(c, lo, hi, n_win, [&](int w0, int nb) -> int {
        int r;
        if ((r = c->res_f64.ensure((size_t)nb * ns)) != PG_OK) return r;
        pg_launch_sample_het(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->cN, c->cshift, nb, c->samp_start.p, c->n_samp,
                             min_pair_sites, c->res_f64.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(het_out + (size_t)w0 * ns, c->res_f64.p, (size_t)nb * ns * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return PG_OK;
    });
}

extern "C" int pg_hapstats(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int min_pair_sites, int diag_nan,
                           double max_dist, const int32_t *pop_row_order, double *h_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    if (c->n_pops < 1) return pg_fail(PG_ERR_STATE, "pg_hapstats needs at least one population");
    if (n_win == 0) return PG_OK;
    if (!h_out || !pop_row_order) return pg_fail(PG_ERR_ARG, "null argument");
    const int P = c->n_pops, n_in = c->h_pop_start[P];
    // every population's range must be a permutation of its own slots
    {
        std::vector<char> seen((size_t)std::max(n_in, 1), 0);
        for (int p = 0; p < P; ++p)
            for (int k = c->h_pop_start[p]; k < c->h_pop_start[p + 1]; ++k) {
                const int s = pop_row_order[k];
                if (s < c->h_pop_start[p] || s >= c->h_pop_start[p + 1] || seen[s])
                    return pg_fail(PG_ERR_ARG, "pop_row_order[%d] = %d is not a permutation of population %d's slots", k, s, p);
                seen[s] = 1;
            }
    }
    HIPCHK(hipSetDevice(c->device));
    if ((rc = c->hap_order.upload(pop_row_order, (size_t)std::max(n_in, 1), c->stream)) != PG_OK) return rc;
    size_t per_win = 0;
    int max_pop = 0;
    for (int p = 0; p < P; ++p) {
        const size_t n = c->h_pop_start[p + 1] - c->h_pop_start[p];
        per_win += n * ((n + 31) / 32);
        max_pop = std::max<int>(max_pop, (int)n);
    }
    return pairwise_run(c, lo, hi, n_win, [&](int w0, int nb) -> int {
        int r;
        if ((r = c->hapbits.ensure((size_t)nb * std::max<size_t>(per_win, 1))) != PG_OK) return r;
        if ((r = c->res_f64.ensure((size_t)nb * P * 3)) != PG_OK) return r;
        pg_launch_hapstats(c->stream, c->Cmat.p, c->Dmat.p, c->n_hap, c->cN, c->cshift, nb, c->pop_start.p, P, max_pop,
                           c->hap_order.p, min_pair_sites, diag_nan, max_dist, c->hapbits.p, per_win, c->res_f64.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_out + (size_t)w0 * P * 3, c->res_f64.p, (size_t)nb * P * 3 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return PG_OK;
    });
}

// ---- site statistics --------------------------------------------------------------------------------
static int quartet_stats(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int p1, int p2, int p3, int p4,
                         double min_data, int sel, int nsum, double *sums_out, int64_t *used_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    const int ps[4] = {p1, p2, p3, p4};
    for (int k = 0; k < 4; ++k)
        if (ps[k] < 0 || ps[k] >= c->n_pops) return pg_fail(PG_ERR_ARG, "population id %d out of range [0,%d)", ps[k], c->n_pops);
    if (sel != PG_SEL_MINOR && sel != PG_SEL_POLARIZE && sel != PG_SEL_FIXED) return pg_fail(PG_ERR_ARG, "unknown allele_sel %d", sel);
    if (n_win == 0) return PG_OK;
    if (!sums_out || !used_out) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    int w0 = 0;
    while (w0 < n_win) {
        int w1 = std::min(n_win, w0 + 65535);
        int nb = w1 - w0;
        int64_t max_len = 0;
        if ((rc = stage_windows(c, lo, hi, w0, w1, nullptr, nullptr, &max_len)) != PG_OK) return rc;
        int max_chunks = (int)((max_len + PG_ABBA_SITES_PER_BLOCK - 1) / PG_ABBA_SITES_PER_BLOCK);
        if ((rc = c->part_f64.ensure((size_t)nb * std::max(max_chunks, 1) * nsum)) != PG_OK) return rc;
        if ((rc = c->part_i64.ensure((size_t)nb * std::max(max_chunks, 1))) != PG_OK) return rc;
        if ((rc = c->res_f64.ensure((size_t)nb * nsum)) != PG_OK) return rc;
        if ((rc = c->res_i64.ensure((size_t)nb)) != PG_OK) return rc;
        // the sums in NumPy's order where the last bit can show (windows of up to PG_NP_MAX_SITES sites; see pg_popdist_stats):
        // k_abba_q raises a flag bit per used site, k_quartet_np adds the sites' terms up again
        uint32_t *flags = nullptr;
        int64_t base = 0;
        const long long np_upto = np_sites_limit(c, "PG_QUARTET_TREE");                                          // window by window
        bool any_short = false;
        for (int w = w0; w < w1; ++w) any_short = any_short || hi[w] - lo[w] <= np_upto;
        if (any_short) {
            int64_t top = 0;
            base = INT64_MAX;
            for (int w = w0; w < w1; ++w)
                if (hi[w] > lo[w]) { base = std::min(base, lo[w]); top = std::max(top, hi[w]); }
            if (base == INT64_MAX) base = top = 0;
            base &= ~(int64_t)127;
            const size_t words = (size_t)((top - base + 31) / 32) + 64;
            if ((rc = c->site_flags.ensure(words)) != PG_OK) return rc;
            HIPCHK(hipMemsetAsync(c->site_flags.p, 0, words * 4, c->stream));
            flags = c->site_flags.p;
        }
        hipEvent_t e0, e1;
        if ((rc = pg_time_begin(c, PG_K_SITESTATS, &e0, &e1)) != PG_OK) return rc;
        pg_launch_abba(c->stream, c->gt.p, c->S, c->win.p, c->win.p + nb, nb, max_chunks, c->pop_start.p, p1, p2, p3, p4,
                       min_data, sel, nsum, c->part_f64.p, c->part_i64.p, c->res_f64.p, c->res_i64.p, flags, base, np_upto);
        if ((rc = pg_time_end(c, PG_K_SITESTATS, e0, e1, 1)) != PG_OK) return rc;
        HIPCHK(hipGetLastError());
        if ((rc = c->out_pin.ensure((size_t)nb * (nsum + 1))) != PG_OK) return rc;
        HIPCHK(hipMemcpyAsync(c->out_pin.p, c->res_f64.p, (size_t)nb * nsum * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->out_pin.p + (size_t)nb * nsum, c->res_i64.p, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
        if ((rc = stream_wait(c)) != PG_OK) return rc;
        memcpy(sums_out + (size_t)w0 * nsum, c->out_pin.p, (size_t)nb * nsum * 8);
        memcpy(used_out + w0, c->out_pin.p + (size_t)nb * nsum, (size_t)nb * 8);
        w0 = w1;
    }
    return PG_OK;
}

extern "C" int pg_abbababa(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int p1, int p2, int p3, int p4,
                           double min_data, double *sums_out, int64_t *used_out) {
    return quartet_stats(c, lo, hi, n_win, p1, p2, p3, p4, min_data, PG_SEL_POLARIZE, PG_ABBA_NSUM, sums_out, used_out);
}

extern "C" int pg_fourpop(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int p1, int p2, int p3, int p4,
                          double min_data, int allele_sel, double *sums_out, int64_t *used_out) {
    return quartet_stats(c, lo, hi, n_win, p1, p2, p3, p4, min_data, allele_sel, PG_FOURPOP_NSUM, sums_out, used_out);
}

extern "C" int pg_popfreq(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int64_t *l_out, int64_t *S_out,
                          int64_t *pairsum_out, double *theta_pi_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    if (c->n_pops < 1 || c->n_pops > PG_MAX_POPS) return pg_fail(PG_ERR_ARG, "pg_popfreq supports 1..%d populations (got %d)", PG_MAX_POPS, c->n_pops);
    if (n_win == 0) return PG_OK;
    if (!l_out || !S_out || !pairsum_out) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    const int P = c->n_pops;
    int w0 = 0;
    while (w0 < n_win) {
        int w1 = std::min(n_win, w0 + 65535);
        int nb = w1 - w0;
        int64_t max_len = 0;
        if ((rc = stage_windows(c, lo, hi, w0, w1, nullptr, nullptr, &max_len)) != PG_OK) return rc;
        int max_chunks = (int)((max_len + PG_SITES_PER_BLOCK - 1) / PG_SITES_PER_BLOCK);
        size_t nres = (size_t)nb * (1 + 2 * P), nall = nres + (theta_pi_out ? (size_t)nb * P : 0);
        if ((rc = c->res_i64.ensure(nall)) != PG_OK) return rc;
        HIPCHK(hipMemsetAsync(c->res_i64.p, 0, nall * 8, c->stream));
        unsigned long long *dl = reinterpret_cast<unsigned long long *>(c->res_i64.p);
        unsigned long long *dS = dl + nb, *dP = dS + (size_t)nb * P;
        double *dT = theta_pi_out ? reinterpret_cast<double *>(dP + (size_t)nb * P) : nullptr;
        // theta_pi_out: one flag bit per site of the span of these windows, raised by the counting kernel, walked in site order by
        // k_popfreq_ordered (the reference's sequential sum)
        uint32_t *flags = nullptr;
        int64_t base = 0;
        if (theta_pi_out) {
            int64_t top = 0;
            base = INT64_MAX;
            for (int w = w0; w < w1; ++w)
                if (hi[w] > lo[w]) { base = std::min(base, lo[w]); top = std::max(top, hi[w]); }
            if (base == INT64_MAX) base = top = 0;
            base &= ~(int64_t)127;
            const size_t words = (size_t)((top - base + 31) / 32) + 64;
            if ((rc = c->site_flags.ensure(words)) != PG_OK) return rc;
            HIPCHK(hipMemsetAsync(c->site_flags.p, 0, words * 4, c->stream));
            flags = c->site_flags.p;
        }
        hipEvent_t e0, e1;
        if ((rc = pg_time_begin(c, PG_K_SITESTATS, &e0, &e1)) != PG_OK) return rc;
        pg_launch_popfreq(c->stream, c->gt.p, c->S, c->n_hap, c->win.p, c->win.p + nb, nb, max_chunks, c->pop_start.p, P, dl, dS, dP,
                          flags, base);
        if ((rc = pg_time_end(c, PG_K_SITESTATS, e0, e1, 1)) != PG_OK) return rc;
        HIPCHK(hipGetLastError());
        if (flags) {
            if ((rc = pg_time_begin(c, PG_K_ORDERED, &e0, &e1)) != PG_OK) return rc;
            pg_launch_popfreq_ordered(c->stream, c->gt.p, c->S, c->win.p, c->win.p + nb, nb, c->pop_start.p, P, flags, base, dT);
            if ((rc = pg_time_end(c, PG_K_ORDERED, e0, e1, 1)) != PG_OK) return rc;
            HIPCHK(hipGetLastError());
        }
        if ((rc = c->out_pin.ensure(nall)) != PG_OK) return rc;
        HIPCHK(hipMemcpyAsync(c->out_pin.p, dl, nall * 8, hipMemcpyDeviceToHost, c->stream));
        if ((rc = stream_wait(c)) != PG_OK) return rc;
        const int64_t *hp = reinterpret_cast<const int64_t *>(c->out_pin.p);
        memcpy(l_out + w0, hp, (size_t)nb * 8);
        memcpy(S_out + (size_t)w0 * P, hp + nb, (size_t)nb * P * 8);
        memcpy(pairsum_out + (size_t)w0 * P, hp + nb + (size_t)nb * P, (size_t)nb * P * 8);
        if (theta_pi_out) memcpy(theta_pi_out + (size_t)w0 * P, hp + nres, (size_t)nb * P * 8);
        w0 = w1;
    }
    return PG_OK;
}

extern "C" int pg_site_counts(pg_ctx *c, int64_t site_lo, int64_t site_hi, int32_t *cnt_out) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (site_lo < 0 || site_hi < site_lo || site_hi > c->cap_sites) return pg_fail(PG_ERR_ARG, "site range out of bounds");
    if (c->n_pops < 1) return pg_fail(PG_ERR_STATE, "no populations set");
    int64_t n = site_hi - site_lo;
    if (n == 0) return PG_OK;
    if (!cnt_out) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    const int64_t chunk = 1 << 22;
    DevBuf<int32_t> &tmp = c->site_tmp;          // kept in the context: freq.py calls this once per block of sites
    int rc = tmp.ensure((size_t)std::min(n, chunk) * c->n_pops * 4);
    if (rc != PG_OK) return rc;
    for (int64_t s = site_lo; s < site_hi; s += chunk) {
        int64_t e = std::min(site_hi, s + chunk);
        pg_launch_site_counts(c->stream, c->gt.p, c->S, s, e, c->pop_start.p, c->n_pops, tmp.p);
        hipError_t err = hipGetLastError();
        if (err == hipSuccess)
            err = hipMemcpyAsync(cnt_out + (size_t)(s - site_lo) * c->n_pops * 4, tmp.p, (size_t)(e - s) * c->n_pops * 16, hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
        if (err != hipSuccess) return pg_fail(PG_ERR_HIP, "pg_site_counts: %s", hipGetErrorString(err));
    }
    return PG_OK;
}

extern "C" int pg_site_target(pg_ctx *c, int64_t site_lo, int64_t site_hi, int target, double min_data, int as_counts, int has_threshold,
                              double threshold, void *values_out, uint8_t *keep_out) {
    if (!c) return pg_fail(PG_ERR_ARG, "null ctx");
    if (c->n_hap <= 0) return pg_fail(PG_ERR_STATE, "pg_set_samples must be called first");
    if (site_lo < 0 || site_hi < site_lo || site_hi > c->cap_sites) return pg_fail(PG_ERR_ARG, "site range out of bounds");
    if (c->n_pops < 1) return pg_fail(PG_ERR_STATE, "no populations set");
    if (target != 1 && target != 2) return pg_fail(PG_ERR_ARG, "pg_site_target: target must be 1 (derived) or 2 (minor)");
    if (target == 1 && c->n_pops < 2) return pg_fail(PG_ERR_ARG, "pg_site_target: the derived allele needs an outgroup population");
    const int64_t n = site_hi - site_lo;
    if (n == 0) return PG_OK;
    if (!values_out || !keep_out) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    const int64_t chunk = 1 << 22;
    const int P = c->n_pops;
    int rc;
    if ((rc = c->site_tmp.ensure((size_t)std::min(n, chunk) * P * 4)) != PG_OK) return rc;
    if ((rc = c->site_val.ensure((size_t)std::min(n, chunk) * P)) != PG_OK) return rc;
    if ((rc = c->site_keep.ensure((size_t)std::min(n, chunk))) != PG_OK) return rc;
    for (int64_t s = site_lo; s < site_hi; s += chunk) {
        const int64_t e = std::min(site_hi, s + chunk);
        pg_launch_site_counts(c->stream, c->gt.p, c->S, s, e, c->pop_start.p, P, c->site_tmp.p);
        pg_launch_site_target(c->stream, c->site_tmp.p, e - s, P, target, min_data, as_counts ? 1 : 0, has_threshold ? 1 : 0, threshold,
                              c->site_val.p, reinterpret_cast<long long *>(c->site_val.p), c->site_keep.p);
        hipError_t err = hipGetLastError();
        if (err == hipSuccess)
            err = hipMemcpyAsync(static_cast<char *>(values_out) + (size_t)(s - site_lo) * P * 8, c->site_val.p, (size_t)(e - s) * P * 8,
                                 hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipMemcpyAsync(keep_out + (s - site_lo), c->site_keep.p, (size_t)(e - s), hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
        if (err != hipSuccess) return pg_fail(PG_ERR_HIP, "pg_site_target: %s", hipGetErrorString(err));
    }
    return PG_OK;
}

extern "C" int pg_hap_called(pg_ctx *c, const int64_t *lo, const int64_t *hi, int n_win, int64_t *called_out) {
    int rc = check_windows(c, lo, hi, n_win);
    if (rc != PG_OK) return rc;
    if (n_win == 0) return PG_OK;
    if (!called_out) return pg_fail(PG_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(c->device));
    int w0 = 0;
    while (w0 < n_win) {
        int w1 = std::min(n_win, w0 + 65535);
        int nb = w1 - w0;
        int64_t max_len = 0;
        if ((rc = stage_windows(c, lo, hi, w0, w1, nullptr, nullptr, &max_len)) != PG_OK) return rc;
        int max_chunks = (int)((max_len + PG_SITES_PER_BLOCK - 1) / PG_SITES_PER_BLOCK);
        size_t nres = (size_t)nb * c->n_hap;
        if ((rc = c->res_i64.ensure(nres)) != PG_OK) return rc;
        HIPCHK(hipMemsetAsync(c->res_i64.p, 0, nres * 8, c->stream));
        pg_launch_hap_called(c->stream, c->gt.p, c->S, c->n_hap, c->win.p, c->win.p + nb, nb, max_chunks,
                             reinterpret_cast<unsigned long long *>(c->res_i64.p));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(called_out + (size_t)w0 * c->n_hap, c->res_i64.p, nres * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        w0 = w1;
    }
    return PG_OK;
}
