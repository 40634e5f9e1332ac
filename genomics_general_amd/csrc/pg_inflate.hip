// `.geno.gz` on the device: BGZF members -> text, a wavefront per member (k_inflate), their CRC-32 checked (k_crc32), the text
// handed to the device tokenizer (pg_tokenize.hip) where it lies.  Replaces gzip.open() in GenoFileReader.__init__ (genomics.py:1917)
// for files written by bgzip (VCF_processing/README.md:33) -- the reference's normal input (popgenWindows.py:313).
//   pg_bgzf_walk             host: the member table of a run of compressed bytes (gzip headers parsed, sizes from the BC field)
//   pg_inflate_members       host: the same members inflated by a pool of threads (zlib) -- the route of blocks the device
//                            tokenizer does not take and of engines without a device
//   pg_inflate_device        members -> text on the device, result copied back (tests, tools/inflate_bench.py)
//   pg_tokenize_submit_bgzf  the submit step of the device tokenizer for a block that is still deflated
#include "pg_ctx.h"
#include "pg_inflate_core.h"
#include "pg_fast_inflate.h"
#include "pg_par_gunzip.h"
#include "pg_fast_deflate.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <deque>
#include <condition_variable>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

namespace {

#ifndef PGI_WAVES
#define PGI_WAVES 6
#endif
// nl_list (may be null): nl_cap offsets per member -- the member's line feeds in front of byte text_limit of the output, found in
// the registers of the flush (pg_inflate_core.h); nl_cnt[m] = how many there are (more than nl_cap: the caller takes the passes over
// the text instead, k_nl_count / k_nl_write)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PGI_WAVES, PGI_WAVES))) void k_inflate(const uint32_t *__restrict__ comp, uint32_t n_dw, const PgiMember *__restrict__ mem,
                                                int n_members, uint8_t *__restrict__ out, uint8_t *__restrict__ sink,
                                                int32_t *__restrict__ status, uint16_t *__restrict__ nl_list, uint32_t nl_cap,
                                                int32_t *__restrict__ nl_cnt, uint64_t text_limit, const uint32_t *__restrict__ crc_fold) {
    __shared__ PgiShared sh;
    const int lane = (int)threadIdx.x;
    const int m = (int)blockIdx.x;
    if (m >= n_members) return;
    const uint32_t in_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)mem[m].in_off);
    const uint32_t in_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)mem[m].in_len);
    const uint32_t out_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)mem[m].out_len);
    const uint64_t out_off = mem[m].out_off;
    const uint64_t lim64 = text_limit > out_off ? text_limit - out_off : 0ull;
    const uint32_t nl_lim = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lim64 > 0x10000ull ? 0x10000u : (uint32_t)lim64));
    uint32_t nl_n = 0;
    const int rc = pgi_member(comp, n_dw, in_off, in_len, out + out_off, out_len, sink + (size_t)m * 128, &sh,   // sink: where lanes without a byte store
                              nl_list ? nl_list + (size_t)m * nl_cap : nullptr, nl_cap, nl_lim, &nl_n,
                              crc_fold, (uint32_t)__builtin_amdgcn_readfirstlane((int)mem[m].crc), lane);    // crc_fold: the member's CRC-32 checked in the flush
    if (nl_cnt && lane == 0) nl_cnt[m] = rc ? 0 : (int32_t)nl_n;
    if (rc && lane == 0) {
        atomicOr(status, rc);
        atomicMin(status + 1, m);
    }
}

// the members' line-feed lists -> the block's list: one block scans the counts (mem_base[m] = line feeds in front of member m, the
// total, and whether a member had more than its list holds), then a wavefront per member moves its entries
__global__ __launch_bounds__(256) void k_member_scan(const int32_t *__restrict__ nl_cnt, int n_members, uint32_t nl_cap, int64_t *__restrict__ mem_base,
                                                     int64_t *__restrict__ total, int32_t *__restrict__ overflow) {
    __shared__ long long sh[256];
    __shared__ int over;
    if (threadIdx.x == 0) over = 0;
    __syncthreads();
    const int per = (n_members + 255) / 256;
    const int t0 = (int)threadIdx.x * per, t1 = t0 + per < n_members ? t0 + per : n_members;
    long long mine = 0;
    for (int t = t0; t < t1; ++t) {
        mine += nl_cnt[t];
        if ((uint32_t)nl_cnt[t] > nl_cap) over = 1;
    }
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const long long x = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += x;
        __syncthreads();
    }
    long long run = sh[threadIdx.x] - mine;
    for (int t = t0; t < t1; ++t) {
        mem_base[t] = run;
        run += nl_cnt[t];
    }
    if (threadIdx.x == 255) {
        *total = sh[255];
        *overflow = over;
    }
}

__global__ __launch_bounds__(64) void k_nl_gather(const uint16_t *__restrict__ nl_list, uint32_t nl_cap, const int32_t *__restrict__ nl_cnt,
                                                  const int64_t *__restrict__ mem_base, const PgiMember *__restrict__ mem, int64_t text_base,
                                                  int64_t *__restrict__ nl_pos) {
    const int m = (int)blockIdx.x;
    const int n = nl_cnt[m];
    const int64_t at = text_base + (int64_t)mem[m].out_off, base = mem_base[m];
    const uint16_t *list = nl_list + (size_t)m * nl_cap;
    for (int i = (int)threadIdx.x; i < n; i += 64) nl_pos[base + i] = at + list[i];
}

// ---- CRC-32 of the inflated members (gzip trailer, RFC 1952 2.3.1): four members per block, a wavefront each ----
// Lane i takes the aligned dwords i, i + 64, ... of the member's text: its register goes 256 bytes forward per dword (four table
// lookups: tab[256 ..] = multiplication by x^2048), the last one only as far as the end of the aligned part (a multiplication by
// x^(32 m), tab[1280 + m]); the lanes' registers are XORed (the CRC is linear), head and tail bytes go through the byte table.
constexpr uint32_t CRC_POLY = 0xEDB88320u;
constexpr int CRC_TAB = 256 + 1024 + 65;

__device__ __forceinline__ uint32_t crc_mul(uint32_t a, uint32_t b) {      // a * b mod P, reflected (zlib's multmodp)
    uint32_t p = 0;
#pragma unroll 4
    for (int k = 31; k >= 0; --k) {
        if ((a >> k) & 1u) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return p;
}

// the CRC-32 of p[0 .. n) by one wavefront (tab: crc_tables() in LDS); the result is lane 0's
__device__ inline uint32_t crc_wave(const uint8_t *__restrict__ p, uint32_t n, const uint32_t *tab, int lane) {
    uint32_t head = (uint32_t)((4u - ((uint32_t)(uintptr_t)p & 3u)) & 3u);
    if (head > n) head = n;
    const uint32_t N = (n - head) >> 2, t = n - head - 4u * N;
    uint32_t s = 0u;
    if (lane == 0) {
        s = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < head; ++k) s = tab[(s ^ p[k]) & 255u] ^ (s >> 8);
    }
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p + head);
    if ((uint32_t)lane < N) {
        uint32_t j = (uint32_t)lane;
        for (; j + 64u < N; j += 64u) {
            const uint32_t v = s ^ w[j];
            s = tab[256 + (v & 255u)] ^ tab[512 + ((v >> 8) & 255u)] ^ tab[768 + ((v >> 16) & 255u)] ^ tab[1024 + (v >> 24)];
        }
        s = crc_mul(tab[1280 + (N - j)], s ^ w[j]);                      // N - j in 1 .. 64 dwords to the end of the aligned part
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s ^= (uint32_t)__shfl_xor((int)s, d, 64);
    if (lane == 0) {
        const uint8_t *q = p + head + 4u * N;
        for (uint32_t k = 0; k < t; ++k) s = tab[(s ^ q[k]) & 255u] ^ (s >> 8);
    }
    return s ^ 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void k_crc32(const uint8_t *__restrict__ text, const PgiMember *__restrict__ mem, int n_members,
                                               const uint32_t *__restrict__ tab_g, int32_t *__restrict__ status) {
    __shared__ uint32_t tab[CRC_TAB];
    for (int k = (int)threadIdx.x; k < CRC_TAB; k += 256) tab[k] = tab_g[k];
    __syncthreads();
    const int lane = (int)threadIdx.x & 63;
    const int m = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (m >= n_members) return;
    const uint32_t crc = crc_wave(text + mem[m].out_off, mem[m].out_len, tab, lane);
    if (lane == 0 && crc != mem[m].crc) {
        atomicOr(status, PGI_ERR_CRC);
        atomicMin(status + 1, m);
    }
}

// the CRC-32 of the pieces text[k * piece ..] (piece bytes each, the last one what is left of *total_p) -> crc_out[k]: the trailers of
// members k_deflate (pg_deflate.hip) writes
__global__ __launch_bounds__(256) void k_crc32_pieces(const uint8_t *__restrict__ text, const long long *__restrict__ total_p, uint32_t piece,
                                                      const uint32_t *__restrict__ tab_g, uint32_t *__restrict__ crc_out) {
    __shared__ uint32_t tab[CRC_TAB];
    for (int k = (int)threadIdx.x; k < CRC_TAB; k += 256) tab[k] = tab_g[k];
    __syncthreads();
    const int lane = (int)threadIdx.x & 63;
    const long long total = *total_p;
    const long long n_pieces = (total + piece - 1) / piece;
    for (long long m = (long long)blockIdx.x * 4 + ((int)threadIdx.x >> 6); m < n_pieces; m += (long long)gridDim.x * 4) {
        const long long left = total - m * piece;
        const uint32_t crc = crc_wave(text + m * piece, (uint32_t)(left < piece ? left : piece), tab, lane);
        if (lane == 0) crc_out[m] = crc;
    }
}

// bytes [off[k], off[k] + len[k]) of text -> out[dst[k] ..] (the scaffold names of a block's runs)
__global__ __launch_bounds__(64) void k_gather_bytes(const uint8_t *__restrict__ text, const int64_t *__restrict__ off,
                                                     const int32_t *__restrict__ len, const int64_t *__restrict__ dst,
                                                     uint8_t *__restrict__ out) {
    const int k = (int)blockIdx.x;
    for (int i = (int)threadIdx.x; i < len[k]; i += 64) out[dst[k] + i] = text[off[k] + i];
}

uint32_t crc_mul_h(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int k = 31; k >= 0; --k) {
        if ((a >> k) & 1u) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return p;
}

const std::vector<uint32_t> &crc_tables() {
    static const std::vector<uint32_t> tab = [] {
        std::vector<uint32_t> t(CRC_TAB, 0u);
        for (uint32_t b = 0; b < 256; ++b) {
            uint32_t c = b;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
            t[b] = c;
        }
        uint32_t x32 = 0x80000000u;                                     // x^0
        for (int k = 0; k < 32; ++k) x32 = (x32 >> 1) ^ ((x32 & 1u) ? CRC_POLY : 0u);
        t[1280] = 0x80000000u;
        for (int m = 1; m <= 64; ++m) t[1280 + m] = crc_mul_h(t[1280 + m - 1], x32);      // x^(32 m)
        const uint32_t x2048 = t[1280 + 64];
        for (int q = 0; q < 4; ++q)
            for (uint32_t b = 0; b < 256; ++b) t[256 + 256 * q + b] = crc_mul_h(x2048, b << (8 * q));
        return t;
    }();
    return tab;
}

}  // namespace

// ---- host: member table ----------------------------------------------------------------------------------------------------------
// Walks the gzip members of buf[0 .. len) (RFC 1952); every member must carry the BGZF extra subfield 'B' 'C' (its total size - 1).
// Stops in front of a member that is not complete in buf, after max_members members, or once the members walked hold at least
// max_text bytes of text (max_text <= 0: no limit).  Per member: where its deflate stream lies (in_off, in_len), ISIZE and CRC-32
// of its trailer.  *consumed_out = bytes of buf the walked members occupy.  PG_ERR_PARSE: not a BGZF member at the walk's position.
extern "C" int pg_bgzf_walk(const uint8_t *buf, int64_t len, int64_t max_members, int64_t max_text, uint32_t *in_off, uint32_t *in_len,
                            uint32_t *out_len, uint32_t *crc, int64_t *n_members_out, int64_t *consumed_out, int64_t *text_out) {
    if (!n_members_out || !consumed_out || !text_out || len < 0 || (len > 0 && !buf) || max_members < 0 ||
        (max_members > 0 && (!in_off || !in_len || !out_len || !crc)))
        return pg_fail(PG_ERR_ARG, "pg_bgzf_walk: bad argument");
    if (len >= (1ll << 32)) return pg_fail(PG_ERR_ARG, "pg_bgzf_walk: at most 4 GiB of compressed bytes at a time");
    int64_t o = 0, n = 0, text = 0;
    while (n < max_members && o + 18 <= len && (max_text <= 0 || text < max_text)) {
        const uint8_t *h = buf + o;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
            return pg_fail(PG_ERR_PARSE, "the input stops being BGZF at compressed offset +%lld (bad member header)", (long long)o);
        const int flg = h[3];
        const int64_t xlen = h[10] | (h[11] << 8);
        if (o + 12 + xlen > len) break;
        int64_t bsize = -1;
        for (int64_t x = 12; x + 4 <= 12 + xlen;) {
            const int64_t slen = h[x + 2] | (h[x + 3] << 8);
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = (h[x + 4] | (h[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 0) return pg_fail(PG_ERR_PARSE, "a gzip member without the BGZF size field at compressed offset +%lld", (long long)o);
        if (o + bsize > len) break;
        int64_t hl = 12 + xlen;
        if (flg & 8) {                                              // FNAME
            while (hl < bsize && h[hl]) ++hl;
            ++hl;
        }
        if (flg & 16) {                                             // FCOMMENT
            while (hl < bsize && h[hl]) ++hl;
            ++hl;
        }
        if (flg & 2) hl += 2;                                       // FHCRC
        if (hl + 8 > bsize) return pg_fail(PG_ERR_PARSE, "damaged BGZF member at compressed offset +%lld", (long long)o);
        in_off[n] = (uint32_t)(o + hl);
        in_len[n] = (uint32_t)(bsize - hl - 8);
        const uint8_t *tr = h + bsize - 8;
        crc[n] = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        out_len[n] = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        text += out_len[n];
        o += bsize;
        ++n;
    }
    *n_members_out = n;
    *consumed_out = o;
    *text_out = text;
    return PG_OK;
}

// ---- host: a pool of threads inflates members (zlib) ------------------------------------------------------------------------------
// member k: raw deflate stream comp[in_off[k] .. + in_len[k]) -> dst[out_off[k] .. + out_len[k]); crc (may be null): checked.
extern "C" int pg_inflate_members(const uint8_t *comp, const uint32_t *in_off, const uint32_t *in_len, const int64_t *out_off,
                                  const uint32_t *out_len, const uint32_t *crc, int64_t n_members, uint8_t *dst, int n_threads) {
    if (n_members < 0 || (n_members > 0 && (!comp || !in_off || !in_len || !out_off || !out_len || !dst)))
        return pg_fail(PG_ERR_ARG, "pg_inflate_members: bad argument");
    int nt = n_threads > 0 ? n_threads : pg_host_threads();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, (n_members + 15) / 16));
    std::atomic<int64_t> next(0), bad(-1);
    // (the library's own decoder and checksum, pg_fast_inflate.h / pg_crc32_fast.h: three times zlib's rate per thread; PG_GZIP_FAST=0
    // keeps zlib's inflate)
    static const bool use_zlib = getenv("PG_GZIP_FAST") && atoi(getenv("PG_GZIP_FAST")) == 0;
    auto work = [&]() {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (use_zlib && inflateInit2(&zs, -15) != Z_OK) { bad.store(0); return; }
        pgfi::State *st = use_zlib ? nullptr : new pgfi::State();
        for (;;) {
            const int64_t k0 = next.fetch_add(16);
            if (k0 >= n_members || bad.load() >= 0) break;
            for (int64_t k = k0; k < std::min(n_members, k0 + 16); ++k) {
                bool ok;
                if (use_zlib) {
                    inflateReset(&zs);
                    zs.next_in = const_cast<Bytef *>(comp + in_off[k]);
                    zs.avail_in = in_len[k];
                    zs.next_out = dst + out_off[k];
                    zs.avail_out = out_len[k];
                    const int rc = out_len[k] || in_len[k] ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
                    ok = rc == Z_STREAM_END && zs.avail_out == 0;
                } else if (!out_len[k] && !in_len[k]) {
                    ok = true;
                } else {
                    pgfi::start_at(*st, comp + in_off[k], in_len[k], 0);
                    st->win_len = 0;
                    uint64_t n = 0;
                    // (one element of room behind the member's size: a stream that holds more must not pass as complete)
                    const int rc = pgfi::inflate(*st, dst + out_off[k], out_len[k], &n);
                    ok = (rc == pgfi::STREAM_END && n == out_len[k]) ||
                         (rc == pgfi::NEED_OUTPUT && n == out_len[k] && st->pend_len == 0 && [&] {          // the end-of-block code may still be unread
                              uint8_t extra;
                              uint64_t m = 0;
                              return pgfi::inflate(*st, &extra, 1, &m) == pgfi::STREAM_END && m == 0;
                          }());
                }
                ok = ok && (!crc || pg_crc32(0u, dst + out_off[k], out_len[k]) == crc[k]);
                if (!ok) { bad.store(k); break; }
            }
        }
        if (use_zlib) inflateEnd(&zs);
        delete st;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    if (bad.load() >= 0) return pg_fail(PG_ERR_PARSE, "damaged BGZF member (member %lld of the block does not inflate to its recorded size and checksum)", (long long)bad.load());
    return PG_OK;
}

// ---- host: ONE gzip stream (what `gzip` writes: `.geno.gz`, README.md:106 of the reference, read there by gzip.open) ---------------
// A single deflate stream has no independent pieces: it is inflated serially, here by zlib's inflate() straight into the caller's
// block buffer from a thread of the reader (the gzip module of rounds 3 - 5 went through 128 KiB Python byte strings: 0.4 GB/s of
// text; this: profiles/r06/gzip_stream_reader.txt).  Concatenated members (`cat a.gz b.gz`) are followed; BGZF files take the member-
// parallel routes above.
struct pg_gz {
    int fd = -1;
    z_stream zs;
    bool zs_live = false, eof = false, member_done = false;
    std::vector<uint8_t> in, pending;          // compressed bytes read ahead; inflated bytes behind the last line feed handed out
    size_t pending_at = 0;
    int64_t total_in = 0, total_out = 0;
    // the route of pg_fast_inflate.h: the file mapped, this library's own decoder (PG_GZIP_FAST=0: zlib's inflate(), above)
    const uint8_t *map = nullptr;
    size_t map_len = 0, map_at = 0;
    pgfi::State *fi = nullptr;
    bool in_member = false;
    uint32_t crc = 0, isize = 0;
    // ... and of pg_par_gunzip.h: the stream decoded in chunks side by side, a batch of text at a time
    int par_threads = 0;                       // 0: serial
    uint64_t par_chunk = 2u << 20;
    pgpar::Batch *pb = nullptr;                // its spill buffer: text of the last batch that did not fit the caller's buffer
    size_t ptext_at = 0;
    uint64_t pbit = 0;
    uint8_t pwin[32768];
    uint32_t pwl = 0;
    int64_t par_batches = 0, par_fallbacks = 0;
};

// n bytes from src to dst on a few threads (the batch's text into the caller's block buffer)
static void copy_parallel(uint8_t *dst, const uint8_t *src, size_t n, int nt) {
    if (n < (16u << 20) || nt < 2) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + (size_t)nt - 1) / (size_t)nt;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=]() {
            const size_t a = std::min(n, per * (size_t)t), b = std::min(n, a + per);
            memcpy(dst + a, src + a, b - a);
        });
    for (auto &x : th) x.join();
}

static uint32_t crc32_threads(uint32_t crc, const uint8_t *p, size_t n, int nt) {
    if (n < (8u << 20) || nt < 2) return pg_crc32(crc, p, n);
    std::vector<uint32_t> part((size_t)nt);
    std::vector<std::thread> th;
    const size_t per = (n + (size_t)nt - 1) / (size_t)nt;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            const size_t a = std::min(n, per * (size_t)t), b = std::min(n, a + per);
            part[(size_t)t] = pg_crc32(0u, p + a, b - a);
        });
    for (auto &x : th) x.join();
    for (int t = 0; t < nt; ++t) {
        const size_t a = std::min(n, per * (size_t)t), b = std::min(n, a + per);
        crc = (uint32_t)crc32_combine(crc, part[(size_t)t], (z_off_t)(b - a));
    }
    return crc;
}

// CRC-32 beside the decoder: zlib's crc32 runs at about 1 GB/s a thread, the decoder at more than 2 -- so the pieces the decoder
// finishes (8 MiB each, still in the cache) are checksummed by a few helper threads while it goes on, and the pieces' values are
// combined in order (crc32_combine) where the member ends or the call returns.
struct CrcPipe {
    struct Job { const uint8_t *p; size_t n; uint32_t crc; };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> jobs;
    size_t next = 0;
    bool closing = false;
    std::vector<std::thread> th;
    void start(int n_threads) {
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([this]() {
                for (;;) {
                    Job *j = nullptr;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv.wait(lk, [&] { return next < jobs.size() || closing; });
                        if (next >= jobs.size()) return;
                        j = &jobs[next++];
                    }
                    j->crc = pg_crc32(0u, j->p, j->n);
                }
            });
    }
    void push(const uint8_t *p, size_t n) {
        {
            std::lock_guard<std::mutex> lk(m);
            jobs.push_back(Job{p, n, 0u});
        }
        cv.notify_one();
    }
    // waits for the helpers and folds the pieces pushed so far into crc
    uint32_t drain(uint32_t crc) {
        {
            std::lock_guard<std::mutex> lk(m);
            closing = true;
        }
        cv.notify_all();
        for (auto &x : th) x.join();
        th.clear();
        for (const Job &j : jobs) crc = (uint32_t)crc32_combine(crc, j.crc, (z_off_t)j.n);
        jobs.clear();
        next = 0;
        closing = false;
        return crc;
    }
};

// the gzip member header at g->map[g->map_at ..] (RFC 1952); 0: parsed (map_at behind it), 1: the end of the file (only zero padding
// left), < 0: not a gzip member
static int gz_member_header(pg_gz *g) {
    const uint8_t *p = g->map + g->map_at, *e = g->map + g->map_len;
    while (p < e && *p == 0) ++p;                                       // (padding between / behind members, as gzip skips it)
    if (p == e) { g->map_at = g->map_len; return 1; }
    if (e - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return -1;
    const int flg = p[3];
    p += 10;
    if (flg & 4) {
        if (e - p < 2) return -1;
        const size_t xlen = (size_t)p[0] | ((size_t)p[1] << 8);
        p += 2;
        if ((size_t)(e - p) < xlen) return -1;
        p += xlen;
    }
    for (int f = 8; f <= 16; f <<= 1)                                   // FNAME, FCOMMENT: zero-terminated
        if (flg & f) {
            const void *z = memchr(p, 0, (size_t)(e - p));
            if (!z) return -1;
            p = static_cast<const uint8_t *>(z) + 1;
        }
    if (flg & 2) {
        if (e - p < 2) return -1;
        p += 2;
    }
    g->map_at = (size_t)(p - g->map);
    return 0;
}

static int64_t gz_fill_fast(pg_gz *g, uint8_t *dst, int64_t room) {
    int64_t got = 0;
    static const bool skip_crc = getenv("PG_GZIP_NO_CRC") != nullptr;                        // (timing experiments only)
    // (where the CPU multiplies carry-less, pg_crc32 keeps up with the decoder on this thread: pieces of 2 MiB, checksummed while they
    // are in the cache; elsewhere zlib's crc32 on helper threads)
    bool fast_crc = false;
#if defined(__x86_64__)
    fast_crc = pgcrc::have_clmul();
#endif
    const bool piped = room >= (32 << 20) && !skip_crc && !fast_crc;
    CrcPipe pipe;
    const int helpers = std::max(1, std::min(4, pg_host_threads() - 1));
    auto fold = [&]() {                                                                      // the helpers' pieces into g->crc
        if (!pipe.th.empty() || !pipe.jobs.empty()) {
            g->crc = pipe.drain(g->crc);
        }
    };
    int64_t rc_out = 0;
    while (got < room && !g->eof) {
        if (!g->in_member) {
            const int h = gz_member_header(g);
            if (h == 1) {
                if (g->total_out == 0 && !g->member_done && g->map_len > 0) { rc_out = -4; break; }   // (nothing but zeros)
                g->eof = true;
                break;
            }
            if (h < 0) { rc_out = g->member_done ? -5 : -4; break; }
            pgfi::State &st = *g->fi;
            st.in = g->map + g->map_at;
            st.in_end = g->map + g->map_len;
            st.bitbuf = 0; st.bitcnt = 0; st.over = 0; st.phase = 0; st.last = false; st.pend_len = 0; st.win_len = 0; st.tables_fixed = false;
            g->in_member = true;
            g->member_done = false;
            g->crc = 0;
            g->isize = 0;
        }
        uint64_t n = 0;
        const uint64_t piece = piped ? std::min<uint64_t>((uint64_t)(room - got), 8u << 20) : fast_crc ? std::min<uint64_t>((uint64_t)(room - got), 2u << 20) : (uint64_t)(room - got);
        const int rc = pgfi::inflate(*g->fi, dst + got, piece, &n);
        if (n) {
            if (piped) {
                if (pipe.th.empty()) pipe.start(helpers);
                pipe.push(dst + got, (size_t)n);
            } else if (!skip_crc) g->crc = pg_crc32(g->crc, dst + got, (size_t)n);
            g->isize += (uint32_t)n;
            got += (int64_t)n;
        }
        if (rc == pgfi::NEED_OUTPUT) continue;                                               // (the piece is full; the loop ends when the room is)
        if (rc == pgfi::ERR_INPUT) { rc_out = -2; break; }
        if (rc != pgfi::STREAM_END) { rc_out = -4; break; }
        const uint8_t *t = nullptr;
        if (!pgfi::stream_tail(*g->fi, &t) || (g->map + g->map_len) - t < 8) { rc_out = -2; break; }
        const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t want_len = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        fold();
        if ((want_crc != g->crc && !skip_crc) || want_len != g->isize) { rc_out = -4; break; }
        g->map_at = (size_t)(t + 8 - g->map);
        g->in_member = false;
        g->member_done = true;
    }
    fold();
    if (rc_out < 0) return rc_out;
    g->total_out += got;
    return got;
}

extern "C" int pg_gzip_open(const char *path, pg_gz **out) {
    if (!path || !out) return pg_fail(PG_ERR_ARG, "pg_gzip_open: null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return pg_fail(PG_ERR_ARG, "pg_gzip_open: cannot open %s", path);
    pg_gz *g = new pg_gz();
    g->fd = fd;
    memset(&g->zs, 0, sizeof(g->zs));
    if (inflateInit2(&g->zs, 15 + 32) != Z_OK) {             // gzip or zlib header, detected
        close(fd);
        delete g;
        return pg_fail(PG_ERR_HIP, "pg_gzip_open: inflateInit2 failed");
    }
    g->zs_live = true;
    g->in.resize(1 << 20);
    g->zs.avail_in = 0;
    if (!(getenv("PG_GZIP_FAST") && atoi(getenv("PG_GZIP_FAST")) == 0)) {
        struct stat sb;
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
            g->map_len = (size_t)sb.st_size;
            if (g->map_len) {
                void *m = mmap(nullptr, g->map_len, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    g->map = static_cast<const uint8_t *>(m);
                    (void)madvise(m, g->map_len, MADV_SEQUENTIAL);
                }
            }
            if (g->map || g->map_len == 0) g->fi = new pgfi::State();
            // chunks side by side from 8 MiB of compressed bytes and three threads on (PG_GZIP_THREADS=1: the serial decoder)
            int nt = std::min(16, pg_host_threads());
            if (const char *e = getenv("PG_GZIP_THREADS")) nt = atoi(e);
            if (const char *e = getenv("PG_GZIP_CHUNK")) g->par_chunk = (uint64_t)std::max(4096, atoi(e));
            if (g->fi && nt >= 3 && g->map_len >= (getenv("PG_GZIP_CHUNK") ? 0u : (8u << 20))) {
                g->par_threads = nt;
                g->pb = new pgpar::Batch();
            }
        }
    }
    *out = g;
    return PG_OK;
}

// [0] threads of the chunk-parallel decoder (0: serial), [1] batches it decoded, [2] times a member went on serially, [3] bytes of text so far
extern "C" int pg_gzip_stats(pg_gz *g, int64_t *out4) {
    if (!g || !out4) return pg_fail(PG_ERR_ARG, "pg_gzip_stats: null argument");
    out4[0] = g->fi ? (g->pb ? (g->par_threads > 0 ? g->par_threads : -1) : 0) : -2;      // -1: fell back to the serial decoder, -2: zlib
    out4[1] = g->par_batches;
    out4[2] = g->par_fallbacks;
    out4[3] = g->total_out;
    return PG_OK;
}

extern "C" int pg_gzip_close(pg_gz *g) {
    if (!g) return PG_OK;
    if (g->zs_live) inflateEnd(&g->zs);
    if (g->map) munmap(const_cast<uint8_t *>(g->map), g->map_len);
    delete g->fi;
    delete g->pb;
    if (g->fd >= 0) close(g->fd);
    delete g;
    return PG_OK;
}

// inflate up to `room` bytes to dst; returns the number produced (0 at the end of the input), < 0 on a damaged stream
static int gz_member_header(pg_gz *g);
static int64_t gz_fill_fast(pg_gz *g, uint8_t *dst, int64_t room);

// the stream in chunks side by side (pg_par_gunzip.h): a batch of text at a time into g->ptext, handed out from there
static int64_t gz_fill_par(pg_gz *g, uint8_t *dst, int64_t room) {
    int64_t got = 0;
    pgpar::Batch &B = *g->pb;
    while (got < room && !g->eof) {
        if (g->ptext_at < B.spill_len) {
            const size_t take = std::min<size_t>((size_t)B.spill_len - g->ptext_at, (size_t)(room - got));
            copy_parallel(dst + got, B.spill + g->ptext_at, take, g->par_threads);
            g->ptext_at += take;
            got += (int64_t)take;
            continue;
        }
        B.spill_len = 0;
        g->ptext_at = 0;
        if (g->par_threads <= 0) break;                                              // (fell back: the serial decoder goes on)
        if (!g->in_member) {
            const int h = gz_member_header(g);
            if (h == 1) {
                if (g->total_out == 0 && !g->member_done && g->map_len > 0) return -4;
                g->eof = true;
                break;
            }
            if (h < 0) return g->member_done ? -5 : -4;
            g->in_member = true;
            g->member_done = false;
            g->crc = 0;
            g->isize = 0;
            g->pbit = (uint64_t)g->map_at * 8;
            g->pwl = 0;
        }
        const int rc = B.decode(g->map, g->map_len, g->pbit, g->pwin, g->pwl, g->par_threads, g->par_chunk);
        if (rc < 0) {
            // no chain of chunks (no dynamic block start found where one was needed, a piece that does not decode): this member goes on
            // serially from the batch's first bit, with the window in front of it
            ++g->par_fallbacks;
            pgfi::State &st = *g->fi;
            pgfi::start_at(st, g->map, g->map_len, g->pbit);
            memcpy(st.window, g->pwin + (32768 - g->pwl), g->pwl);
            st.win_len = g->pwl;
            g->par_threads = 0;
            break;
        }
        ++g->par_batches;
        const uint64_t direct = B.emit(dst + got, (uint64_t)(room - got));
        for (size_t i = 0; i < B.chain.size(); ++i) {
            const pgpar::Chunk &c = B.ch[(size_t)B.chain[i]];
            g->crc = (uint32_t)crc32_combine(g->crc, c.crc, (z_off_t)c.n);
        }
        g->isize += (uint32_t)B.total;
        got += (int64_t)direct;
        g->pbit = B.end_bit;
        memcpy(g->pwin, B.window_out(), 32768);
        g->pwl = B.w_len_out;
        if (B.ended) {
            const uint8_t *t = g->map + (B.end_bit >> 3);
            if ((g->map + g->map_len) - t < 8) return -2;
            const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t want_len = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            if (want_crc != g->crc || want_len != g->isize) return -4;
            g->map_at = (size_t)(t + 8 - g->map);
            g->in_member = false;
            g->member_done = true;
        }
    }
    g->total_out += got;
    if (g->par_threads <= 0 && got < room && !g->eof) {
        const int64_t r = gz_fill_fast(g, dst + got, room - got);                    // (counts its own bytes)
        if (r < 0) return r;
        got += r;
    }
    return got;
}

static int64_t gz_fill(pg_gz *g, uint8_t *dst, int64_t room) {
    if (g->fi) {
        if (g->map_len == 0) { g->eof = true; return 0; }
        if (g->pb && (g->par_threads > 0 || g->ptext_at < g->pb->spill_len)) return gz_fill_par(g, dst, room);
        return gz_fill_fast(g, dst, room);
    }
    int64_t got = 0;
    while (got < room && !g->eof) {
        if (g->zs.avail_in == 0) {
            const ssize_t r = read(g->fd, g->in.data(), g->in.size());
            if (r < 0) return -1;
            if (r == 0) {
                // the end of the file: fine between members, an error inside one
                if (!g->member_done && g->total_in > 0) return -2;
                g->eof = true;
                break;
            }
            g->zs.next_in = g->in.data();
            g->zs.avail_in = (uInt)r;
            g->total_in += r;
        }
        if (g->member_done) {
            // bytes behind a member: the next member of a concatenated file (trailing zero padding is skipped, as gzip does)
            while (g->zs.avail_in && *g->zs.next_in == 0) { ++g->zs.next_in; --g->zs.avail_in; }
            if (!g->zs.avail_in) continue;
            if (inflateReset(&g->zs) != Z_OK) return -3;
            g->member_done = false;
        }
        const int64_t piece = std::min<int64_t>(room - got, 1 << 30);
        g->zs.next_out = dst + got;
        g->zs.avail_out = (uInt)piece;
        const int rc = inflate(&g->zs, Z_NO_FLUSH);
        got += piece - (int64_t)g->zs.avail_out;
        if (rc == Z_STREAM_END) g->member_done = true;
        else if (rc != Z_OK && rc != Z_BUF_ERROR) return -4;
    }
    g->total_out += got;
    return got;
}

// At least `want` bytes of text (fewer only at the end of the input), extended to the next line feed: dst[0 .. *got_out).  cap must
// exceed want by the longest line the caller expects; a line that does not end inside cap comes back without its line feed
// (*complete_out = 0: call again and append).  *eof_out = 1 when nothing is left.
extern "C" int pg_gzip_read_lines(pg_gz *g, uint8_t *dst, int64_t cap, int64_t want, int64_t *got_out, int *complete_out, int *eof_out) {
    if (!g || !dst || !got_out || !complete_out || !eof_out || cap < 1 || want < 1) return pg_fail(PG_ERR_ARG, "pg_gzip_read_lines: bad argument");
    if (want > cap) want = cap;
    int64_t n = 0;
    // what the last call inflated behind the line feed it stopped at
    if (g->pending_at < g->pending.size()) {
        const int64_t take = std::min<int64_t>((int64_t)(g->pending.size() - g->pending_at), cap);
        memcpy(dst, g->pending.data() + g->pending_at, (size_t)take);
        g->pending_at += (size_t)take;
        n = take;
    }
    if (g->pending_at >= g->pending.size()) { g->pending.clear(); g->pending_at = 0; }
    auto bad = [&](int64_t r) {
        return pg_fail(PG_ERR_PARSE, "damaged gzip stream (%s; %lld bytes of text were read before)", r == -2 ? "the file ends inside a member" : r == -1 ? "read error" : r == -5 ? "bytes behind the last member that are no gzip member" : "invalid deflate data or a wrong checksum", (long long)g->total_out);
    };
    // the bulk: straight into dst
    if (n < want) {
        const int64_t r = gz_fill(g, dst + n, want - n);
        if (r < 0) return bad(r);
        n += r;
    }
    bool complete = false;
    if (n >= want && n > 0) {
        // the first line feed at or behind byte want - 1 ends the block; what is already here behind it waits for the next call
        const void *nl = memchr(dst + want - 1, '\n', (size_t)(n - (want - 1)));
        if (nl) {
            const int64_t cut = (static_cast<const uint8_t *>(nl) - dst) + 1;
            if (cut < n) {
                std::vector<uint8_t> rest(dst + cut, dst + n);
                rest.insert(rest.end(), g->pending.begin() + (long)g->pending_at, g->pending.end());
                g->pending.swap(rest);
                g->pending_at = 0;
            }
            n = cut;
            complete = true;
        }
    }
    // ... else on to the next line feed, a piece at a time
    while (!complete && n < cap && g->pending_at >= g->pending.size()) {
        uint8_t piece[4096];
        const int64_t r = gz_fill(g, piece, (int64_t)sizeof(piece));
        if (r < 0) return bad(r);
        if (r == 0) break;
        const void *nl = memchr(piece, '\n', (size_t)r);
        int64_t take = nl ? (static_cast<const uint8_t *>(nl) - piece) + 1 : r;
        if (take > cap - n) { take = cap - n; nl = nullptr; }
        memcpy(dst + n, piece, (size_t)take);
        n += take;
        if (take < r) {
            g->pending.assign(piece + take, piece + r);
            g->pending_at = 0;
        }
        if (nl) complete = true;
    }
    const bool drained = g->eof && g->pending_at >= g->pending.size();
    if (drained) complete = true;                                    // (the input's last line may lack its line feed)
    *got_out = n;
    *complete_out = complete ? 1 : 0;
    *eof_out = (n == 0 && drained) ? 1 : 0;
    return PG_OK;
}

// ---- host: text -> BGZF (what `bgzip` writes; tools/bgzip.py, bench.py's compressed samples, tests) ------------------------------------
// text[0 .. len) as members of `block` bytes of text each (bgzip: 65280), deflated at `level` by a pool of threads, + the empty EOF
// member when eof_marker != 0.  *out_len_out = bytes written; PG_ERR_ARG when out_cap is too small (len + len / 1000 + 64 KiB + 28
// per member is always enough).
extern "C" int pg_bgzf_compress(const uint8_t *text, int64_t len, int level, int block, int eof_marker, uint8_t *out, int64_t out_cap,
                                int64_t *out_len_out, int n_threads) {
    if (len < 0 || (len > 0 && !text) || !out || !out_len_out || block < 1 || block > 65280 || level < -1 || level > 9)
        return pg_fail(PG_ERR_ARG, "pg_bgzf_compress: bad argument");
    const int64_t n = (len + block - 1) / block;
    int nt = n_threads > 0 ? n_threads : pg_host_threads();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n));
    std::vector<std::vector<uint8_t>> parts((size_t)n);
    std::atomic<int64_t> next(0);
    std::atomic<int> bad(0);
    // level 6 (bgzip's default, and this writer's): the library's own compressor (pg_fast_deflate.h: the ratio of zlib's level 6 on
    // row-by-row text at several times its speed; PG_BGZF_ZLIB=1 or any other level: zlib's deflate)
    static const bool force_zlib = getenv("PG_BGZF_ZLIB") && atoi(getenv("PG_BGZF_ZLIB")) != 0;
    const bool own = level == 6 && !force_zlib;
    auto work = [&]() {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad.store(1); return; }
        pgfd::Work *wk = own ? new pgfd::Work() : nullptr;
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n || bad.load()) break;
            const int64_t a = k * block;
            const uInt m = (uInt)std::min<int64_t>(block, len - a);
            std::vector<uint8_t> &o = parts[(size_t)k];
            o.resize(18 + deflateBound(&zs, m) + 8);
            size_t cl;
            if (own) {
                cl = pgfd::deflate_member(text + a, m, o.data() + 18, o.size() - 26, *wk);
                if (!cl) { bad.store(1); break; }
            } else {
                deflateReset(&zs);
                zs.next_in = const_cast<Bytef *>(text + a);
                zs.avail_in = m;
                zs.next_out = o.data() + 18;
                zs.avail_out = (uInt)(o.size() - 26);
                if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { bad.store(1); break; }
                cl = o.size() - 26 - zs.avail_out;
            }
            size_t total = 18 + cl + 8;
            if (total > 65536) {
                // text that does not deflate (binary or already compressed input, high-entropy INFO strings) in a block near 64 KiB:
                // one stored block (5 + m bytes; 65280 + 5 + 26 <= 65536 is why bgzip's block is 65280) -- ADVICE round 5
                uint8_t *d = o.data() + 18;
                d[0] = 1;
                d[1] = (uint8_t)(m & 255); d[2] = (uint8_t)(m >> 8);
                d[3] = (uint8_t)(~m & 255); d[4] = (uint8_t)((~m >> 8) & 255);
                memcpy(d + 5, text + a, m);
                cl = 5 + (size_t)m;
                total = 18 + cl + 8;
            }
            static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            memcpy(o.data(), head, 16);
            o[16] = (uint8_t)((total - 1) & 255);
            o[17] = (uint8_t)((total - 1) >> 8);
            const uint32_t crc = pg_crc32(0u, text + a, m);
            uint8_t *t = o.data() + 18 + cl;
            for (int b = 0; b < 4; ++b) { t[b] = (uint8_t)(crc >> (8 * b)); t[4 + b] = (uint8_t)((uint32_t)m >> (8 * b)); }
            o.resize(total);
        }
        deflateEnd(&zs);
        delete wk;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    if (bad.load()) return pg_fail(PG_ERR_ARG, "pg_bgzf_compress: deflate failed");
    int64_t at = 0;
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (auto &o : parts) {
        if (at + (int64_t)o.size() > out_cap) return pg_fail(PG_ERR_ARG, "pg_bgzf_compress: output buffer too small");
        memcpy(out + at, o.data(), o.size());
        at += (int64_t)o.size();
    }
    if (eof_marker) {
        if (at + 28 > out_cap) return pg_fail(PG_ERR_ARG, "pg_bgzf_compress: output buffer too small");
        memcpy(out + at, eof, 28);
        at += 28;
    }
    *out_len_out = at;
    return PG_OK;
}

// ---- device ------------------------------------------------------------------------------------------------------------------------
// the member table of a block -> page-locked array -> device; queues k_inflate (+ k_crc32) on `st`.  comp_d: the compressed bytes on
// the device (padded: n_dw dwords may be read); text_d: where byte 0 of the first member's text goes.
// nl_cap > 0: the members' line feeds in front of byte text_limit of text_d are listed as they are inflated (I.nl_list, I.nl_cnt),
// scanned (I.mem_base; total -> *d_total, a list that was too short -> *d_over)
static int inflate_queue(pg_ctx *c, hipStream_t st, pg_ctx::Inflate &I, const uint32_t *comp_d, uint32_t n_dw, const uint32_t *in_off,
                         const uint32_t *in_len, const uint32_t *out_len, const uint32_t *crc, int64_t n_members, uint8_t *text_d,
                         uint32_t nl_cap = 0, uint64_t text_limit = 0, int64_t *d_total = nullptr, int32_t *d_over = nullptr,
                         hipStream_t crc_st = nullptr) {
    int rc;
    if (nl_cap) {
        if ((rc = I.nl_list.ensure_roomy((size_t)(n_members + 1) * nl_cap)) != PG_OK) return rc;
        if ((rc = I.nl_cnt.ensure_roomy((size_t)n_members + 1)) != PG_OK) return rc;
        if ((rc = I.mem_base.ensure_roomy((size_t)n_members + 1)) != PG_OK) return rc;
    }
    if ((rc = I.h_members.ensure_roomy((size_t)n_members + 1)) != PG_OK) return rc;
    if ((rc = I.members.ensure_roomy((size_t)n_members + 1)) != PG_OK) return rc;
    if ((rc = I.status.ensure(4)) != PG_OK) return rc;              // [0, 1] the decoder's error bits | first bad member, [2, 3] the same of the CRC check on its own stream
    if ((rc = I.sink.ensure_roomy((size_t)(n_members + 1) * 128)) != PG_OK) return rc;
    if ((rc = I.h_status.ensure(4)) != PG_OK) return rc;
    uint64_t at = 0;
    for (int64_t k = 0; k < n_members; ++k) {
        I.h_members.p[k] = PgiMember{in_off[k], in_len[k], at, out_len[k], crc ? crc[k] : 0u};
        at += out_len[k];
    }
    if (!I.crc_tab.p) {
        const std::vector<uint32_t> &t = crc_tables();
        if ((rc = I.crc_tab.ensure(t.size())) != PG_OK) return rc;
        HIPCHK(hipMemcpy(I.crc_tab.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    I.h_status.p[0] = I.h_status.p[2] = 0;
    I.h_status.p[1] = I.h_status.p[3] = 0x7FFFFFFF;
    HIPCHK(hipMemcpyAsync(I.status.p, I.h_status.p, 16, hipMemcpyHostToDevice, st));
    if (n_members == 0) return PG_OK;
    HIPCHK(hipMemcpyAsync(I.members.p, I.h_members.p, (size_t)n_members * sizeof(PgiMember), hipMemcpyHostToDevice, st));
    // The members' CRC-32: in k_inflate's flush, where the text is in registers (round 6; PG_BGZF_CRC_FOLD=0: k_crc32, a kernel that
    // reads the text again -- 0.5 ms per GiB of it)
    static const bool fold = !(getenv("PG_BGZF_CRC_FOLD") && atoi(getenv("PG_BGZF_CRC_FOLD")) == 0);
    const bool check = crc && !getenv("PG_BGZF_NO_CRC");
    if (check && fold && !I.crc_fold.p) {
        std::vector<uint32_t> t(PGI_CRC_TAB);
        pgi_make_crc_tables(t.data());
        if ((rc = I.crc_fold.ensure(t.size())) != PG_OK) return rc;
        HIPCHK(hipMemcpy(I.crc_fold.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)n_members), dim3(64), 0, st, comp_d, n_dw, I.members.p, (int)n_members, text_d, I.sink.p, I.status.p,
                       nl_cap ? I.nl_list.p : nullptr, nl_cap, nl_cap ? I.nl_cnt.p : nullptr, text_limit, check && fold ? I.crc_fold.p : nullptr);
    HIPCHK(hipGetLastError());
    if (nl_cap) {
        hipLaunchKernelGGL(k_member_scan, dim3(1), dim3(256), 0, st, I.nl_cnt.p, (int)n_members, nl_cap, I.mem_base.p, d_total, d_over);
        HIPCHK(hipGetLastError());
    }
    I.crc_pending = false;
    if (check && !fold) {
        if (crc_st) {
            // the check needs the text and nobody needs the check before the block's rows are handed out: on a stream of its own,
            // beside the line-feed scan and the tokenizer's kernels (0.5 ms per GiB of text off the chain of the block)
            if (!I.ev_inflated) HIPCHK(hipEventCreateWithFlags(&I.ev_inflated, hipEventDisableTiming));
            if (!I.ev_crc) HIPCHK(hipEventCreateWithFlags(&I.ev_crc, hipEventDisableTiming));
            HIPCHK(hipEventRecord(I.ev_inflated, st));
            HIPCHK(hipStreamWaitEvent(crc_st, I.ev_inflated, 0));
        }
        hipLaunchKernelGGL(k_crc32, dim3((unsigned)((n_members + 3) / 4)), dim3(256), 0, crc_st ? crc_st : st, text_d, I.members.p, (int)n_members,
                           I.crc_tab.p, crc_st ? I.status.p + 2 : I.status.p);
        HIPCHK(hipGetLastError());
        if (crc_st) {
            HIPCHK(hipEventRecord(I.ev_crc, crc_st));
            I.crc_pending = true;
        }
    }
    return PG_OK;
}

static int inflate_error(const int32_t *status) {
    if (!status[0]) return PG_OK;
    const int b = status[0];
    return pg_fail(PG_ERR_PARSE, "damaged BGZF member (member %d of the block: %s%s%s%s%s%s%s)", status[1],
                   b & PGI_ERR_BTYPE ? "invalid block type " : "", b & PGI_ERR_STORED ? "invalid stored block lengths " : "",
                   b & PGI_ERR_CODE ? "invalid code " : "", b & PGI_ERR_DIST ? "invalid distance too far back " : "",
                   b & PGI_ERR_OUT ? "does not inflate to its recorded size " : "", b & PGI_ERR_IN ? "unexpected end of its bytes " : "",
                   b & PGI_ERR_CRC ? "CRC check failed" : "");
}

// members of comp[0 .. comp_len) (table as pg_bgzf_walk gives it) -> dst[0 .. sum out_len), inflated on the device.  crc may be null
// (no check).  kernel_ms_out (may be null): device time of k_inflate + k_crc32 (HIP events).
extern "C" int pg_inflate_device(pg_ctx *c, const uint8_t *comp, int64_t comp_len, const uint32_t *in_off, const uint32_t *in_len,
                                 const uint32_t *out_len, const uint32_t *crc, int64_t n_members, uint8_t *dst, double *kernel_ms_out) {
    if (!c || comp_len < 0 || n_members < 0 || (n_members > 0 && (!comp || !in_off || !in_len || !out_len || !dst)))
        return pg_fail(PG_ERR_ARG, "pg_inflate_device: bad argument");
    if (comp_len >= (1ll << 32)) return pg_fail(PG_ERR_ARG, "pg_inflate_device: at most 4 GiB of compressed bytes at a time");
    HIPCHK(hipSetDevice(c->device));
    int64_t total = 0;
    for (int64_t k = 0; k < n_members; ++k) {
        if ((int64_t)in_off[k] + in_len[k] > comp_len) return pg_fail(PG_ERR_ARG, "pg_inflate_device: member %lld lies outside the compressed bytes", (long long)k);
        total += out_len[k];
    }
    pg_ctx::Inflate &I = c->inf;
    int rc;
    const size_t n_dw = ((size_t)comp_len + 3) / 4;
    if ((rc = I.comp.ensure(n_dw + 1)) != PG_OK) return rc;
    if ((rc = I.text.ensure((size_t)total + 64)) != PG_OK) return rc;
    hipStream_t st = c->stream_up;
    HIPCHK(hipMemcpyAsync(I.comp.p, comp, (size_t)comp_len, hipMemcpyHostToDevice, st));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
    rc = inflate_queue(c, st, I, I.comp.p, (uint32_t)n_dw, in_off, in_len, out_len, crc, n_members, I.text.p);
    if (rc == PG_OK) {
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipMemcpyAsync(I.h_status.p, I.status.p, 8, hipMemcpyDeviceToHost, st));
        if (total) HIPCHK(hipMemcpyAsync(dst, I.text.p, (size_t)total, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (kernel_ms_out) *kernel_ms_out = ms;
        rc = inflate_error(I.h_status.p);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// used by pg_tokenize.hip ------------------------------------------------------------------------------------------------------------
int pg_inflate_queue(pg_ctx *c, hipStream_t st, pg_ctx::Inflate &I, const uint32_t *comp_d, uint32_t n_dw, const uint32_t *in_off,
                     const uint32_t *in_len, const uint32_t *out_len, const uint32_t *crc, int64_t n_members, uint8_t *text_d,
                     uint32_t nl_cap, uint64_t text_limit, int64_t *d_total, int32_t *d_over, hipStream_t crc_st) {
    return inflate_queue(c, st, I, comp_d, n_dw, in_off, in_len, out_len, crc, n_members, text_d, nl_cap, text_limit, d_total, d_over, crc_st);
}
// the block's list of line feeds from the members' lists (pg_tokenize_parse, a block that arrived deflated)
void pg_launch_nl_gather(hipStream_t st, pg_ctx::Inflate &I, uint32_t nl_cap, int64_t n_members, int64_t text_base, int64_t *nl_pos) {
    if (n_members > 0)
        hipLaunchKernelGGL(k_nl_gather, dim3((unsigned)n_members), dim3(64), 0, st, I.nl_list.p, nl_cap, I.nl_cnt.p, I.mem_base.p, I.members.p, text_base, nl_pos);
}
int pg_inflate_error(const int32_t *status) { return inflate_error(status); }
void pg_launch_gather_bytes(hipStream_t st, const uint8_t *text, const int64_t *off, const int32_t *len, const int64_t *dst, int n,
                            uint8_t *out) {
    if (n > 0) hipLaunchKernelGGL(k_gather_bytes, dim3((unsigned)n), dim3(64), 0, st, text, off, len, dst, out);
}

// used by pg_deflate.hip: the CRC-32 of every `piece` bytes of text[0 .. *total_d) -> crc_out (grid sized for at most max_pieces)
int pg_launch_crc32_pieces(pg_ctx *c, hipStream_t st, const uint8_t *text, const long long *total_d, uint32_t piece, int64_t max_pieces,
                           uint32_t *crc_out) {
    pg_ctx::Inflate &I = c->inf;
    int rc;
    if (!I.crc_tab.p) {
        const std::vector<uint32_t> &t = crc_tables();
        if ((rc = I.crc_tab.ensure(t.size())) != PG_OK) return rc;
        HIPCHK(hipMemcpy(I.crc_tab.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((max_pieces + 3) / 4, 2048));
    hipLaunchKernelGGL(k_crc32_pieces, dim3((unsigned)blocks), dim3(256), 0, st, text, total_d, piece, I.crc_tab.p, crc_out);
    HIPCHK(hipGetLastError());
    return PG_OK;
}
