// K0: native multi-threaded `.geno` tokenizer (host only; no GPU call).
// Replaces the reference's per-line Python parsing and per-window string->array conversion:
//   GenoFileReader.nextSite / parseGenoLine  genomics.py:1940-1945, 1884-1904
//   splitSeq / forceHomo / seqArrayToNumArray genomics.py:390-396, 407-408, 74-77
// Output is the engine's one-hot int8 code (A=1 C=2 G=4 T=8, everything else 0 = missing) written straight
// into device-slot order, plus int32 positions and the location of each row's scaffold token.
#include "pg_ctx.h"

#include <atomic>
#include <cmath>
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>

#include <zlib.h>

namespace {

struct CodeTables {
    uint8_t base[256];        // allele char -> one-hot
    uint8_t dip[256][2];      // IUPAC diploid char -> two one-hot codes (genomics.py:14-15)
    CodeTables() {
        memset(base, 0, sizeof(base));
        base[(int)'A'] = 1; base[(int)'C'] = 2; base[(int)'G'] = 4; base[(int)'T'] = 8;
        memset(dip, 0, sizeof(dip));
        const char *d = "ACGKMNSRTWY";
        const char *pr[] = {"AA", "CC", "GG", "GT", "AC", "NN", "CG", "AG", "TT", "AT", "CT"};
        for (int k = 0; d[k]; ++k) {
            dip[(int)d[k]][0] = base[(int)pr[k][0]];
            dip[(int)d[k]][1] = base[(int)pr[k][1]];
        }
    }
};
const CodeTables T;

inline bool is_ws(char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f'; }

struct Shared {
    const char *buf;
    size_t len;
    int fmt, n_cols, max_ploidy, n_hap;
    bool narrow_ok;              // PG_FMT_NARROW_OK: a cell may hold fewer alleles than its column's slots (the rest stay missing)
    const int32_t *col_slot, *col_ploidy;
    int8_t *gt;
    int64_t *pos;
    int64_t *scaf_off;
    int32_t *scaf_len;
    int64_t cap;
    std::atomic<int> err;
    char msg[256];
};

// Is the line [b,e) a data row?  ('#' comment lines and whitespace-only lines are not.)
inline bool data_line(const char *b, const char *e) {
    if (b >= e || *b == '#') return false;
    for (const char *p = b; p < e; ++p)
        if (!is_ws(*p)) return true;
    return false;
}

int64_t count_rows(const char *b, const char *e) {
    int64_t n = 0;
    while (b < e) {
        const char *nl = static_cast<const char *>(memchr(b, '\n', e - b));
        const char *le = nl ? nl : e;
        if (data_line(b, le)) ++n;
        b = le + 1;
    }
    return n;
}

void set_err(Shared &sh, const char *what, int64_t row, int col) {
    int expected = 0;
    if (sh.err.compare_exchange_strong(expected, 1))
        snprintf(sh.msg, sizeof(sh.msg), "%s (data row %lld, genotype column %d)", what, (long long)row, col);
}

void parse_range(Shared &sh, const char *b, const char *e, int64_t row) {
    const int H = sh.n_hap;
    while (b < e && !sh.err.load(std::memory_order_relaxed)) {
        const char *nl = static_cast<const char *>(memchr(b, '\n', e - b));
        const char *le = nl ? nl : e;
        if (data_line(b, le)) {
            if (row >= sh.cap) { set_err(sh, "more rows than output capacity", row, -1); return; }
            const char *p = b;
            while (p < le && is_ws(*p)) ++p;
            const char *s0 = p;
            while (p < le && !is_ws(*p)) ++p;
            sh.scaf_off[row] = s0 - sh.buf;
            sh.scaf_len[row] = (int32_t)(p - s0);
            while (p < le && is_ws(*p)) ++p;
            // position: optional sign + digits (Python int())
            bool neg = false;
            if (p < le && (*p == '+' || *p == '-')) { neg = (*p == '-'); ++p; }
            if (p >= le || *p < '0' || *p > '9') { set_err(sh, "position is not an integer", row, -1); return; }
            // (the reference parses Python integers, genomics.py:1884-1904: positions beyond 2^31 -- chromosomes of more than
            // 2.1 Gb exist -- are carried as int64; eighteen digits is where this parser stops)
            // (leading zeros do not count -- int("0000000000000000000012") is 12 there --, and the nineteenth significant digit is
            // refused before it is multiplied in: ADVICE round 5)
            int64_t v = 0;
            int nd = 0;
            while (p < le && *p == '0') ++p;
            while (p < le && *p >= '0' && *p <= '9' && nd < 18) { v = v * 10 + (*p - '0'); ++p; ++nd; }
            if (p < le && !is_ws(*p)) { set_err(sh, "position is not an integer of at most 18 digits", row, -1); return; }
            sh.pos[row] = neg ? -v : v;
            int8_t *out = sh.gt + (size_t)row * H;
            memset(out, 0, H);
            for (int c = 0; c < sh.n_cols; ++c) {
                while (p < le && is_ws(*p)) ++p;
                const char *c0 = p;
                while (p < le && !is_ws(*p)) ++p;
                const int w = (int)(p - c0);
                if (w == 0) { set_err(sh, "row has fewer genotype columns than the header", row, c); return; }
                const int pl = sh.col_ploidy[c];
                if (pl <= 0) continue;
                const int32_t *slots = sh.col_slot + (size_t)c * sh.max_ploidy;
                switch (sh.fmt) {
                    case PG_FMT_PHASED: {
                        const int have = (w + 1) / 2;                        // splitSeq: every other character (genomics.py:390-396)
                        if (w != 2 * pl - 1 && !(sh.narrow_ok && have < pl)) { set_err(sh, "Sample ploidy doesn't match number of sequences (cell width)", row, c); return; }
                        for (int k = 0; k < (have < pl ? have : pl); ++k) out[slots[k]] = (int8_t)T.base[(uint8_t)c0[2 * k]];
                        break;
                    }
                    case PG_FMT_PAIRS:
                    case PG_FMT_HAPLO:
                        if (w != pl && !(sh.narrow_ok && w < pl)) { set_err(sh, "Sample ploidy doesn't match number of sequences (cell width)", row, c); return; }
                        for (int k = 0; k < (w < pl ? w : pl); ++k) out[slots[k]] = (int8_t)T.base[(uint8_t)c0[k]];
                        break;
                    default:  // PG_FMT_DIPLO
                        if (w != 1 || pl != 2) { set_err(sh, "Sample ploidy doesn't match number of sequences (diplo cell)", row, c); return; }
                        out[slots[0]] = (int8_t)T.dip[(uint8_t)c0[0]][0];
                        out[slots[1]] = (int8_t)T.dip[(uint8_t)c0[0]][1];
                        break;
                }
            }
            ++row;
        }
        b = le + 1;
    }
}

}  // namespace

// line-start cuts of a buffer for `nt` threads
static std::vector<size_t> line_cuts(const char *buf, size_t len, int nt) {
    std::vector<size_t> cut(nt + 1, len);
    cut[0] = 0;
    for (int t = 1; t < nt; ++t) {
        size_t guess = len / nt * t;
        if (guess < cut[t - 1]) guess = cut[t - 1];
        const char *nl = static_cast<const char *>(memchr(buf + guess, '\n', len - guess));
        cut[t] = nl ? (size_t)(nl - buf) + 1 : len;
    }
    return cut;
}

extern "C" int pg_count_lines(const char *buf, size_t len, int64_t *n_rows_out) {
    if ((!buf && len) || !n_rows_out) return pg_fail(PG_ERR_ARG, "pg_count_lines: null argument");
    // all host threads: on a gigabyte block a single-threaded count costs as much as the parallel parse that follows it
    int nt = pg_host_threads();
    if (nt < 1) nt = 1;
    if ((size_t)nt > len / (1 << 20) + 1) nt = (int)(len / (1 << 20) + 1);
    if (nt == 1) {
        *n_rows_out = count_rows(buf, buf + len);
        return PG_OK;
    }
    const std::vector<size_t> cut = line_cuts(buf, len, nt);
    std::vector<int64_t> cnt(nt, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { cnt[t] = count_rows(buf + cut[t], buf + cut[t + 1]); });
    for (auto &x : th) x.join();
    int64_t n = 0;
    for (int t = 0; t < nt; ++t) n += cnt[t];
    *n_rows_out = n;
    return PG_OK;
}

// Scaffold runs of raw text: the byte offset of the first data line of every run of data lines that share their first field.
extern "C" int pg_text_runs(const char *buf, size_t len, int64_t *starts_out, int64_t cap, int64_t *n_out) {
    if ((!buf && len) || !n_out || (cap > 0 && !starts_out)) return pg_fail(PG_ERR_ARG, "pg_text_runs: null argument");
    const char *p = buf, *end = buf + len;
    const char *prev = nullptr;                      // first field of the previous data line
    size_t prev_len = 0;
    int64_t n = 0;
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        const char *le = nl ? nl : end;
        if (le > p && *p != '#' && !(le - p == 1 && *p == '\r')) {
            const char *q = p;
            while (q < le && *q != '\t' && *q != ' ') ++q;
            const size_t fl = (size_t)(q - p);
            if (!prev || fl != prev_len || memcmp(p, prev, fl) != 0) {
                if (n < cap) starts_out[n] = (int64_t)(p - buf);
                ++n;
                prev = p;
                prev_len = fl;
            }
        }
        p = nl ? nl + 1 : end;
    }
    *n_out = n;
    return PG_OK;
}

// Walk the data lines of raw text that belong to one scaffold run until a position is reached (the window-range cuts of the
// multi-GPU input plan: a coordinate window is a function of (scaffold, position), genomics.py:1988-2017).
extern "C" int pg_text_seek_pos(const char *buf, size_t len, int whole, const char *scaf, size_t scaf_len, int64_t pos_min,
                                int64_t *off_out, int32_t *state_out, int64_t *pos_out, int64_t *rows_out) {
    if ((!buf && len) || (!scaf && scaf_len) || !off_out || !state_out || !pos_out || !rows_out)
        return pg_fail(PG_ERR_ARG, "pg_text_seek_pos: null argument");
    const char *p = buf, *end = buf + len;
    int64_t rows = 0;
    *state_out = -1;
    *pos_out = 0;
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        if (!nl && !whole) break;                     // an incomplete last line: the caller comes back with more text
        const char *le = nl ? nl : end;
        if (data_line(p, le)) {
            const char *s0 = p;
            while (s0 < le && is_ws(*s0)) ++s0;
            const char *q = s0;
            while (q < le && !is_ws(*q)) ++q;
            if ((size_t)(q - s0) != scaf_len || memcmp(s0, scaf, scaf_len) != 0) {
                *state_out = 0;
                break;
            }
            while (q < le && is_ws(*q)) ++q;
            int64_t v = 0;
            const char *d = q;
            while (d < le && *d >= '0' && *d <= '9') v = v * 10 + (*d++ - '0');
            if (d == q) return pg_fail(PG_ERR_PARSE, "pg_text_seek_pos: a data line without a position");
            if (v >= pos_min) {
                *state_out = 1;
                *pos_out = v;
                break;
            }
            ++rows;
        }
        p = nl ? nl + 1 : end;
    }
    *off_out = (int64_t)(p - buf);
    *rows_out = rows;
    return PG_OK;
}

// Offset of the data line that follows `n_rows` data lines (the row-index cuts of sites windows, genomics.py:2032-2108).
extern "C" int pg_text_skip_rows(const char *buf, size_t len, int64_t n_rows, int64_t *off_out, int64_t *rows_out) {
    if ((!buf && len) || !off_out || !rows_out) return pg_fail(PG_ERR_ARG, "pg_text_skip_rows: null argument");
    const char *p = buf, *end = buf + len;
    int64_t rows = 0;
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        const char *le = nl ? nl : end;
        if (data_line(p, le)) {
            if (rows == n_rows) break;
            ++rows;
        }
        p = nl ? nl + 1 : end;
    }
    *off_out = (int64_t)(p - buf);
    *rows_out = rows;
    return PG_OK;
}

extern "C" int pg_encode_text(const char *buf, size_t len, int fmt, int n_cols, int max_ploidy, const int32_t *col_slot,
                              const int32_t *col_ploidy, int n_hap, int8_t *gt_out, int64_t *pos_out, int64_t *scaf_off,
                              int32_t *scaf_len, int64_t cap_sites, int64_t *n_sites_out, int n_threads) {
    if ((!buf && len) || !col_slot || !col_ploidy || !n_sites_out) return pg_fail(PG_ERR_ARG, "pg_encode_text: null argument");
    if (cap_sites > 0 && (!gt_out || !pos_out || !scaf_off || !scaf_len)) return pg_fail(PG_ERR_ARG, "pg_encode_text: null output");
    const bool narrow_ok = (fmt & PG_FMT_NARROW_OK) != 0;
    fmt &= ~PG_FMT_NARROW_OK;
    if (fmt < PG_FMT_PHASED || fmt > PG_FMT_DIPLO) return pg_fail(PG_ERR_ARG, "unknown genotype format %d", fmt);
    if (n_cols < 0 || max_ploidy < 1 || n_hap < 1) return pg_fail(PG_ERR_ARG, "bad column description");
    for (int c = 0; c < n_cols; ++c) {
        if (col_ploidy[c] < 0 || col_ploidy[c] > max_ploidy) return pg_fail(PG_ERR_ARG, "col_ploidy[%d] out of range", c);
        for (int k = 0; k < col_ploidy[c]; ++k) {
            int s = col_slot[(size_t)c * max_ploidy + k];
            if (s < 0 || s >= n_hap) return pg_fail(PG_ERR_ARG, "col_slot[%d][%d]=%d out of range", c, k, s);
        }
        if (fmt == PG_FMT_DIPLO && col_ploidy[c] != 0 && col_ploidy[c] != 2)
            return pg_fail(PG_ERR_PARSE, "Sample ploidy (%d) doesn't match number of sequences (2): diplo format is diploid", col_ploidy[c]);
    }
    *n_sites_out = 0;
    if (len == 0) return PG_OK;
    int nt = n_threads > 0 ? n_threads : pg_host_threads();
    if (nt < 1) nt = 1;
    if ((size_t)nt > len / (1 << 16) + 1) nt = (int)(len / (1 << 16) + 1);
    // chunk boundaries at line starts
    std::vector<size_t> cut(nt + 1, len);
    cut[0] = 0;
    for (int t = 1; t < nt; ++t) {
        size_t guess = len / nt * t;
        if (guess < cut[t - 1]) guess = cut[t - 1];
        const char *nl = static_cast<const char *>(memchr(buf + guess, '\n', len - guess));
        cut[t] = nl ? (size_t)(nl - buf) + 1 : len;
    }
    std::vector<int64_t> cnt(nt, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { cnt[t] = count_rows(buf + cut[t], buf + cut[t + 1]); });
        for (auto &x : th) x.join();
    }
    std::vector<int64_t> base(nt + 1, 0);
    for (int t = 0; t < nt; ++t) base[t + 1] = base[t] + cnt[t];
    if (base[nt] > cap_sites) return pg_fail(PG_ERR_ARG, "text holds %lld rows but output capacity is %lld", (long long)base[nt], (long long)cap_sites);
    Shared sh;
    sh.narrow_ok = narrow_ok;
    sh.buf = buf; sh.len = len; sh.fmt = fmt; sh.n_cols = n_cols; sh.max_ploidy = max_ploidy; sh.n_hap = n_hap;
    sh.col_slot = col_slot; sh.col_ploidy = col_ploidy; sh.gt = gt_out; sh.pos = pos_out; sh.scaf_off = scaf_off;
    sh.scaf_len = scaf_len; sh.cap = cap_sites; sh.err = 0; sh.msg[0] = 0;
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { parse_range(sh, buf + cut[t], buf + cut[t + 1], base[t]); });
        for (auto &x : th) x.join();
    }
    if (sh.err.load()) return pg_fail(PG_ERR_PARSE, "%s", sh.msg);
    *n_sites_out = base[nt];
    return PG_OK;
}

// --inferPloidy (genomics.py:1108-1111: the reference takes a sample's ploidy in a window from the cells the window holds): the
// widths of the watched columns' cells, row by row, reported as the rows at which they change.  state[n_cols] carries the widths
// of the last data row from one buffer of whole lines to the next (-1 in every watched column: no row yet); the first data row
// after such a state is always reported.  A row with fewer cells than columns keeps the missing columns' widths (the tokenizer
// names that row).  change_width_out[k][n_cols] holds the widths from row change_row_out[k] (index among the buffer's data rows)
// on; unwatched columns 0.  *n_changes_out is the true count also when it exceeds cap (call again with more room and the same
// state as before: state is only advanced when everything fitted).
extern "C" int pg_text_cell_widths(const char *buf, size_t len, int n_cols, const int32_t *col_watch, int32_t *state,
                                   int64_t *change_row_out, int32_t *change_width_out, int64_t cap, int64_t *n_changes_out,
                                   int64_t *n_rows_out) {
    if ((!buf && len) || !col_watch || !state || !n_changes_out || !n_rows_out || n_cols < 0 || (cap > 0 && (!change_row_out || !change_width_out)))
        return pg_fail(PG_ERR_ARG, "pg_text_cell_widths: bad argument");
    int nt = pg_host_threads();
    if (nt < 1) nt = 1;
    if ((size_t)nt > len / (1 << 20) + 1) nt = (int)(len / (1 << 20) + 1);
    const std::vector<size_t> cut = line_cuts(buf, len, nt);
    struct Part { int64_t rows = 0; std::vector<int64_t> at; std::vector<int32_t> w; std::vector<int32_t> last; };
    std::vector<Part> part((size_t)nt);
    auto work = [&](int t) {
        Part &P = part[(size_t)t];
        std::vector<int32_t> cur((size_t)n_cols, -1), line((size_t)n_cols);
        const char *b = buf + cut[(size_t)t], *e = buf + cut[(size_t)t + 1];
        while (b < e) {
            const char *nl = static_cast<const char *>(memchr(b, '\n', (size_t)(e - b)));
            const char *le = nl ? nl : e;
            if (data_line(b, le)) {
                const char *p = b;
                for (int f = 0; f < 2; ++f) {                       // scaffold, position
                    while (p < le && is_ws(*p)) ++p;
                    while (p < le && !is_ws(*p)) ++p;
                }
                bool differs = false;
                for (int c = 0; c < n_cols; ++c) {
                    while (p < le && is_ws(*p)) ++p;
                    const char *c0 = p;
                    while (p < le && !is_ws(*p)) ++p;
                    const int32_t w = col_watch[c] ? (p > c0 ? (int32_t)(p - c0) : cur[(size_t)c]) : 0;
                    line[(size_t)c] = w;
                    differs |= w != cur[(size_t)c];
                }
                if (differs) {
                    P.at.push_back(P.rows);
                    P.w.insert(P.w.end(), line.begin(), line.end());
                    cur = line;
                }
                ++P.rows;
            }
            b = le + 1;
        }
        P.last = cur;
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    // the parts one after the other: a part's first report is dropped when its widths are those of the row in front of it
    std::vector<int32_t> cur(state, state + n_cols);
    for (int c = 0; c < n_cols; ++c)
        if (!col_watch[c]) cur[(size_t)c] = 0;
    int64_t n = 0, rows = 0;
    for (int t = 0; t < nt; ++t) {
        const Part &P = part[(size_t)t];
        for (size_t k = 0; k < P.at.size(); ++k) {
            const int32_t *w = P.w.data() + k * (size_t)n_cols;
            // (a part starts from "unknown": a column it never saw a cell of still reads -1 there -- keep what is known)
            std::vector<int32_t> full(w, w + n_cols);
            for (int c = 0; c < n_cols; ++c)
                if (full[(size_t)c] < 0) full[(size_t)c] = cur[(size_t)c];
            if (full == cur) continue;
            if (n < cap) {
                change_row_out[n] = rows + P.at[k];
                memcpy(change_width_out + (size_t)n * n_cols, full.data(), (size_t)n_cols * sizeof(int32_t));
            }
            ++n;
            cur = full;
        }
        rows += P.rows;
    }
    *n_changes_out = n;
    *n_rows_out = rows;
    if (n <= cap) memcpy(state, cur.data(), (size_t)n_cols * sizeof(int32_t));
    return PG_OK;
}

extern "C" int pg_scaffold_runs(const char *buf, const int64_t *scaf_off, const int32_t *scaf_len, int64_t n_sites,
                                int64_t *run_start_out, int64_t max_runs, int64_t *n_runs_out) {
    if (!n_runs_out || (n_sites > 0 && (!buf || !scaf_off || !scaf_len))) return pg_fail(PG_ERR_ARG, "pg_scaffold_runs: null argument");
    // row i starts a run when its scaffold token differs from row i-1's: independent per row, so the rows are cut into ranges for
    // the host threads (every row's token sits in a different cache line of the text) and the ranges' run starts concatenated
    int nt = pg_host_threads();
    if (nt < 1) nt = 1;
    if ((int64_t)nt > n_sites / 65536 + 1) nt = (int)(n_sites / 65536 + 1);
    std::vector<std::vector<int64_t>> found(nt);
    auto work = [&](int t) {
        const int64_t a = n_sites * t / nt, b = n_sites * (t + 1) / nt;
        for (int64_t i = a; i < b; ++i) {
            const bool same = i > 0 && scaf_len[i] == scaf_len[i - 1] &&
                              memcmp(buf + scaf_off[i], buf + scaf_off[i - 1], (size_t)scaf_len[i]) == 0;
            if (!same) found[t].push_back(i);
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    int64_t n = 0;
    for (int t = 0; t < nt; ++t)
        for (int64_t i : found[t]) {
            if (n < max_runs && run_start_out) run_start_out[n] = i;
            ++n;
        }
    *n_runs_out = n;
    if (n > max_runs) return pg_fail(PG_ERR_ARG, "%lld scaffold runs exceed capacity %lld", (long long)n, (long long)max_runs);
    return PG_OK;
}

// ---- packed `.pgeno` blocks -> engine codes --------------------------------------------------------------------------
// cells[n_rows][n_cols]: one byte per genotype cell, first allele in the low nibble, second in the high nibble, both as the
// engine's one-hot code (written by genomics_general_amd/genoio.PackedWriter from the output of pg_encode_text).
extern "C" int pg_decode_packed(const uint8_t *cells, int64_t n_rows, int n_cols, int max_ploidy, const int32_t *col_slot,
                                const int32_t *col_ploidy, int n_hap, int8_t *gt_out, int n_threads) {
    if (n_rows < 0 || n_cols < 0 || max_ploidy < 1 || max_ploidy > 2 || n_hap < 1)
        return pg_fail(PG_ERR_ARG, "pg_decode_packed: bad shape");
    if (n_rows == 0) return PG_OK;
    if (!cells || !col_slot || !col_ploidy || !gt_out) return pg_fail(PG_ERR_ARG, "pg_decode_packed: null argument");
    int nt = n_threads > 0 ? n_threads : pg_host_threads();
    if (nt < 1) nt = 1;
    if ((int64_t)nt > n_rows) nt = (int)n_rows;
    auto work = [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            const uint8_t *c = cells + r * n_cols;
            int8_t *o = gt_out + r * n_hap;
            memset(o, 0, (size_t)n_hap);
            for (int k = 0; k < n_cols; ++k) {
                const int pl = col_ploidy[k];
                if (pl <= 0) continue;
                const int s0 = col_slot[k * max_ploidy];
                if (s0 >= 0) o[s0] = (int8_t)(c[k] & 15);
                if (pl > 1) {
                    const int s1 = col_slot[k * max_ploidy + 1];
                    if (s1 >= 0) o[s1] = (int8_t)(c[k] >> 4);
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i) th.emplace_back(work, n_rows * i / nt, n_rows * (i + 1) / nt);
    for (auto &t : th) t.join();
    return PG_OK;
}

// ---- deflated `.pgeno` chunks -> their destinations, on all host threads ---------------------------------------------------
// A block's payload is pos || cells, deflated in independent chunks (genoio.PackedWriter).  The logical output of the chunks,
// concatenated, is written to two destinations: the first len_a bytes to dst_a (the block's positions), the rest to dst_b (its
// cells, e.g. rows of the page-locked array an upload will read) -- no intermediate copy of the gigabyte.
extern "C" int pg_inflate_chunks(const uint8_t *src, const int64_t *src_off, const int64_t *src_len, const int64_t *raw_len,
                                 int n_chunks, uint8_t *dst_a, int64_t len_a, uint8_t *dst_b, int64_t len_b, int n_threads) {
    if (n_chunks < 0 || (n_chunks > 0 && (!src || !src_off || !src_len || !raw_len)) || len_a < 0 || len_b < 0)
        return pg_fail(PG_ERR_ARG, "pg_inflate_chunks: bad argument");
    if ((len_a > 0 && !dst_a) || (len_b > 0 && !dst_b)) return pg_fail(PG_ERR_ARG, "pg_inflate_chunks: null destination");
    std::vector<int64_t> at(n_chunks + 1, 0);
    for (int i = 0; i < n_chunks; ++i) {
        if (src_len[i] < 0 || raw_len[i] < 0) return pg_fail(PG_ERR_ARG, "pg_inflate_chunks: negative size");
        at[i + 1] = at[i] + raw_len[i];
    }
    if (at[n_chunks] != len_a + len_b) return pg_fail(PG_ERR_ARG, "pg_inflate_chunks: the chunks hold %lld bytes, the destinations %lld", (long long)at[n_chunks], (long long)(len_a + len_b));
    int nt = n_threads > 0 ? n_threads : pg_host_threads();
    if (nt < 1) nt = 1;
    if (nt > n_chunks) nt = n_chunks > 0 ? n_chunks : 1;
    std::atomic<int> next(0), bad(0);
    auto work = [&]() {
        std::vector<uint8_t> tmp;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_chunks || bad.load()) return;
            const int64_t a = at[i], b = at[i + 1];
            uint8_t *out;
            const bool straddles = a < len_a && b > len_a;
            if (straddles) { tmp.resize((size_t)(b - a)); out = tmp.data(); }
            else out = a >= len_a ? dst_b + (a - len_a) : dst_a + a;
            uLongf got = (uLongf)(b - a);
            const int rc = uncompress(out, &got, src + src_off[i], (uLong)src_len[i]);
            if (rc != Z_OK || (int64_t)got != b - a) { bad.store(1); return; }
            if (straddles) {
                memcpy(dst_a + a, tmp.data(), (size_t)(len_a - a));
                memcpy(dst_b, tmp.data() + (len_a - a), (size_t)(b - len_a));
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work);
    for (auto &x : th) x.join();
    if (bad.load()) return pg_fail(PG_ERR_PARSE, "pg_inflate_chunks: a chunk does not inflate to its recorded size (damaged .pgeno block)");
    return PG_OK;
}

// ---- freq.py rows: numbers -> text on all host threads ---------------------------------------------------------------------
// One output row per kept site: scaffold name, position, then one cell per population, tab separated (freq.py:98-113).
//   mode 0  cells = the four base counts "a,c,g,t" (values[n][n_pops][4], --asCounts without --target)
//   mode 1  cells = one integer per population (values[n][n_pops], --target with --asCounts)
//   mode 2  cells = one float per population as NumPy prints a double that has been rounded to four decimals (`nan`, `0.0`,
//           `0.3333`): the shortest repr of the double nearest to k / 10000 is k / 10000 written out
// run_of_row[i] indexes the scaffold names (names = concatenated bytes, name_off[r] .. name_off[r+1]); keep[i] == 0 drops row i.
namespace {
inline void put_int(std::string &o, long long v) {
    char buf[24];
    int n = 0;
    bool neg = v < 0;
    unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) o.push_back('-');
    while (n) o.push_back(buf[--n]);
}
inline void put_round4(std::string &o, double v) {
    if (std::isnan(v)) { o += "nan"; return; }
    if (std::isinf(v)) { o += v < 0 ? "-inf" : "inf"; return; }
    if (std::signbit(v)) o.push_back('-');
    const double a = std::fabs(v);
    if (a >= 1e15) { char b[40]; snprintf(b, sizeof(b), "%.17g", a); o += b; return; }       // (never a frequency)
    const unsigned long long k = (unsigned long long)std::llround(a * 1e4);
    put_int(o, (long long)(k / 10000));
    o.push_back('.');
    unsigned f = (unsigned)(k % 10000);
    char d[4] = {(char)('0' + f / 1000), (char)('0' + f / 100 % 10), (char)('0' + f / 10 % 10), (char)('0' + f % 10)};
    int nd = 4;
    while (nd > 1 && d[nd - 1] == '0') --nd;
    o.append(d, (size_t)nd);
}
}  // namespace

extern "C" int pg_format_freq_rows(int mode, int64_t n_rows, int n_pops, const void *values, const int64_t *pos,
                                   const int32_t *run_of_row, const char *names, const int64_t *name_off, const uint8_t *keep,
                                   char *out, int64_t out_cap, int64_t *out_len, int n_threads) {
    if (mode < 0 || mode > 2 || n_rows < 0 || n_pops < 1 || !out_len) return pg_fail(PG_ERR_ARG, "pg_format_freq_rows: bad argument");
    *out_len = 0;
    if (n_rows == 0) return PG_OK;
    if (!values || !pos || !run_of_row || !names || !name_off || !out) return pg_fail(PG_ERR_ARG, "pg_format_freq_rows: null argument");
    int nt = n_threads > 0 ? n_threads : pg_host_threads();
    if (nt < 1) nt = 1;
    if ((int64_t)nt > n_rows / 4096 + 1) nt = (int)(n_rows / 4096 + 1);
    std::vector<std::string> part(nt);
    auto work = [&](int t) {
        std::string &o = part[t];
        const int64_t a = n_rows * t / nt, b = n_rows * (t + 1) / nt;
        o.reserve((size_t)(b - a) * (size_t)(24 + n_pops * (mode == 0 ? 16 : 8)));
        for (int64_t i = a; i < b; ++i) {
            if (keep && !keep[i]) continue;
            const int r = run_of_row[i];
            o.append(names + name_off[r], (size_t)(name_off[r + 1] - name_off[r]));
            o.push_back('\t');
            put_int(o, pos[i]);
            for (int q = 0; q < n_pops; ++q) {
                o.push_back('\t');
                if (mode == 0) {
                    const int32_t *c = static_cast<const int32_t *>(values) + ((size_t)i * n_pops + q) * 4;
                    for (int k = 0; k < 4; ++k) {
                        if (k) o.push_back(',');
                        put_int(o, c[k]);
                    }
                } else if (mode == 1) {
                    put_int(o, static_cast<const int64_t *>(values)[(size_t)i * n_pops + q]);
                } else {
                    put_round4(o, static_cast<const double *>(values)[(size_t)i * n_pops + q]);
                }
            }
            o.push_back('\n');
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    int64_t total = 0;
    for (auto &x : part) total += (int64_t)x.size();
    *out_len = total;
    if (total > out_cap) return pg_fail(PG_ERR_ARG, "pg_format_freq_rows: %lld bytes of rows, capacity %lld", (long long)total, (long long)out_cap);
    int64_t at = 0;
    for (auto &x : part) {
        memcpy(out + at, x.data(), x.size());
        at += (int64_t)x.size();
    }
    return PG_OK;
}


// ---- rows of float64 values the way Python prints them ------------------------------------------------------------------------------
// repr(float) (what `np.round(M, r).astype(str)` / `str(x)` give for a float64, genomics.py:2288-2306 makeDistMatString & co.): the
// shortest digits that read back as the same double (std::to_chars, the same digits as David Gay's mode 0 in CPython), fixed
// notation while the decimal point lies within (-4, 16] of the first digit -- ".0" behind an integer --, else d[.ddd]e+XX with at
// least two exponent digits; "nan", "inf", "-inf"; "-0.0".
namespace {

void put_py_repr(std::string &o, double v) {
    if (v != v) { o += "nan"; return; }
    if (v == HUGE_VAL) { o += "inf"; return; }
    if (v == -HUGE_VAL) { o += "-inf"; return; }
    char b[40];
    const auto r = std::to_chars(b, b + sizeof(b), v, std::chars_format::scientific);      // [-]d[.ddd]e[+-]XX[X]
    const char *p = b, *e = r.ptr;
    if (*p == '-') { o.push_back('-'); ++p; }
    const char *ex = p;
    while (ex < e && *ex != 'e') ++ex;
    char dig[24];
    int nd = 0;
    for (const char *q = p; q < ex; ++q)
        if (*q != '.') dig[nd++] = *q;
    int e10 = 0;
    {
        const char *q = ex + 1;
        const bool neg = *q == '-';
        if (*q == '-' || *q == '+') ++q;
        for (; q < e; ++q) e10 = e10 * 10 + (*q - '0');
        if (neg) e10 = -e10;
    }
    if (nd == 1 && dig[0] == '0') { o += "0.0"; return; }
    const int decpt = e10 + 1;                                // the value is 0.d1d2... x 10^decpt
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            o += "0.";
            o.append((size_t)(-decpt), '0');
            o.append(dig, (size_t)nd);
        } else if (decpt >= nd) {
            o.append(dig, (size_t)nd);
            o.append((size_t)(decpt - nd), '0');
            o += ".0";
        } else {
            o.append(dig, (size_t)decpt);
            o.push_back('.');
            o.append(dig + decpt, (size_t)(nd - decpt));
        }
        return;
    }
    o.push_back(dig[0]);
    if (nd > 1) {
        o.push_back('.');
        o.append(dig + 1, (size_t)(nd - 1));
    }
    o.push_back('e');
    int x = e10;
    if (x < 0) { o.push_back('-'); x = -x; } else o.push_back('+');
    char t[8];
    int k = 0;
    do { t[k++] = (char)('0' + x % 10); x /= 10; } while (x);
    if (k < 2) t[k++] = '0';
    while (k) o.push_back(t[--k]);
}

}  // namespace

// n_rows rows of row_len values: value j of row i is v[i * row_len + j], rounded as np.round(x, round_to) rounds when round_to >= 0
// (rint(x * 10^r) / 10^r; round_to < 0: as it is), printed as repr(float) does, the values of a row separated by `sep`, the row
// ended by a line feed; prefix (may be NULL): bytes put in front of row i, prefix[prefix_off[i] .. prefix_off[i + 1]).
extern "C" int pg_format_float_rows(const double *v, int64_t n_rows, int64_t row_len, int round_to, char sep, const char *prefix,
                                    const int64_t *prefix_off, char *out, int64_t out_cap, int64_t *out_len, int n_threads) {
    if (n_rows < 0 || row_len < 0 || !out_len || round_to > 22) return pg_fail(PG_ERR_ARG, "pg_format_float_rows: bad argument");
    *out_len = 0;
    if (n_rows == 0) return PG_OK;
    if ((!v && row_len) || (prefix && !prefix_off)) return pg_fail(PG_ERR_ARG, "pg_format_float_rows: null argument");
    double f = 1.0;
    for (int k = 0; k < round_to; ++k) f *= 10.0;             // exact up to 10^22
    int nt = n_threads > 0 ? n_threads : pg_host_threads();
    if (nt < 1) nt = 1;
    if ((int64_t)nt > n_rows * row_len / 2048 + 1) nt = (int)(n_rows * row_len / 2048 + 1);
    if ((int64_t)nt > n_rows) nt = (int)n_rows;
    std::vector<std::string> part(nt);
    auto work = [&](int t) {
        std::string &o = part[t];
        const int64_t a = n_rows * t / nt, b = n_rows * (t + 1) / nt;
        o.reserve((size_t)(b - a) * (size_t)(row_len * 8 + 16));
        for (int64_t i = a; i < b; ++i) {
            if (prefix) o.append(prefix + prefix_off[i], (size_t)(prefix_off[i + 1] - prefix_off[i]));
            const double *row = v + (size_t)i * (size_t)row_len;
            for (int64_t j = 0; j < row_len; ++j) {
                if (j) o.push_back(sep);
                double x = row[j];
                if (round_to >= 0 && x == x && x != HUGE_VAL && x != -HUGE_VAL) x = std::nearbyint(x * f) / f;
                put_py_repr(o, x);
            }
            o.push_back('\n');
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    int64_t total = 0;
    for (auto &x : part) total += (int64_t)x.size();
    *out_len = total;
    if (!out) return PG_OK;                                  // sizing call
    if (total > out_cap) return pg_fail(PG_ERR_ARG, "pg_format_float_rows: %lld bytes of rows, capacity %lld", (long long)total, (long long)out_cap);
    int64_t at = 0;
    for (auto &x : part) {
        memcpy(out + at, x.data(), x.size());
        at += (int64_t)x.size();
    }
    return PG_OK;
}
