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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

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
                                                int32_t *__restrict__ nl_cnt, uint64_t text_limit) {
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
                              nl_list ? nl_list + (size_t)m * nl_cap : nullptr, nl_cap, nl_lim, &nl_n, lane);
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

__global__ __launch_bounds__(256) void k_crc32(const uint8_t *__restrict__ text, const PgiMember *__restrict__ mem, int n_members,
                                               const uint32_t *__restrict__ tab_g, int32_t *__restrict__ status) {
    __shared__ uint32_t tab[CRC_TAB];
    for (int k = (int)threadIdx.x; k < CRC_TAB; k += 256) tab[k] = tab_g[k];
    __syncthreads();
    const int lane = (int)threadIdx.x & 63;
    const int m = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (m >= n_members) return;
    const uint8_t *p = text + mem[m].out_off;
    const uint32_t n = mem[m].out_len;
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
        if ((s ^ 0xFFFFFFFFu) != mem[m].crc) {
            atomicOr(status, PGI_ERR_CRC);
            atomicMin(status + 1, m);
        }
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
    auto work = [&]() {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) { bad.store(0); return; }
        for (;;) {
            const int64_t k0 = next.fetch_add(16);
            if (k0 >= n_members || bad.load() >= 0) break;
            for (int64_t k = k0; k < std::min(n_members, k0 + 16); ++k) {
                inflateReset(&zs);
                zs.next_in = const_cast<Bytef *>(comp + in_off[k]);
                zs.avail_in = in_len[k];
                zs.next_out = dst + out_off[k];
                zs.avail_out = out_len[k];
                const int rc = out_len[k] || in_len[k] ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
                const bool ok = rc == Z_STREAM_END && zs.avail_out == 0 &&
                                (!crc || (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + out_off[k], out_len[k]) == crc[k]);
                if (!ok) { bad.store(k); break; }
            }
        }
        inflateEnd(&zs);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    if (bad.load() >= 0) return pg_fail(PG_ERR_PARSE, "damaged BGZF member (member %lld of the block does not inflate to its recorded size and checksum)", (long long)bad.load());
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
    auto work = [&]() {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad.store(1); return; }
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n || bad.load()) break;
            const int64_t a = k * block;
            const uInt m = (uInt)std::min<int64_t>(block, len - a);
            std::vector<uint8_t> &o = parts[(size_t)k];
            o.resize(18 + deflateBound(&zs, m) + 8);
            deflateReset(&zs);
            zs.next_in = const_cast<Bytef *>(text + a);
            zs.avail_in = m;
            zs.next_out = o.data() + 18;
            zs.avail_out = (uInt)(o.size() - 26);
            if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { bad.store(1); break; }
            size_t cl = o.size() - 26 - zs.avail_out, total = 18 + cl + 8;
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
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), text + a, m);
            uint8_t *t = o.data() + 18 + cl;
            for (int b = 0; b < 4; ++b) { t[b] = (uint8_t)(crc >> (8 * b)); t[4 + b] = (uint8_t)((uint32_t)m >> (8 * b)); }
            o.resize(total);
        }
        deflateEnd(&zs);
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
                         uint32_t nl_cap = 0, uint64_t text_limit = 0, int64_t *d_total = nullptr, int32_t *d_over = nullptr) {
    int rc;
    if (nl_cap) {
        if ((rc = I.nl_list.ensure_roomy((size_t)(n_members + 1) * nl_cap)) != PG_OK) return rc;
        if ((rc = I.nl_cnt.ensure_roomy((size_t)n_members + 1)) != PG_OK) return rc;
        if ((rc = I.mem_base.ensure_roomy((size_t)n_members + 1)) != PG_OK) return rc;
    }
    if ((rc = I.h_members.ensure_roomy((size_t)n_members + 1)) != PG_OK) return rc;
    if ((rc = I.members.ensure_roomy((size_t)n_members + 1)) != PG_OK) return rc;
    if ((rc = I.status.ensure(2)) != PG_OK) return rc;
    if ((rc = I.sink.ensure_roomy((size_t)(n_members + 1) * 128)) != PG_OK) return rc;
    if ((rc = I.h_status.ensure(2)) != PG_OK) return rc;
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
    I.h_status.p[0] = 0;
    I.h_status.p[1] = 0x7FFFFFFF;
    HIPCHK(hipMemcpyAsync(I.status.p, I.h_status.p, 8, hipMemcpyHostToDevice, st));
    if (n_members == 0) return PG_OK;
    HIPCHK(hipMemcpyAsync(I.members.p, I.h_members.p, (size_t)n_members * sizeof(PgiMember), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)n_members), dim3(64), 0, st, comp_d, n_dw, I.members.p, (int)n_members, text_d, I.sink.p, I.status.p,
                       nl_cap ? I.nl_list.p : nullptr, nl_cap, nl_cap ? I.nl_cnt.p : nullptr, text_limit);
    HIPCHK(hipGetLastError());
    if (nl_cap) {
        hipLaunchKernelGGL(k_member_scan, dim3(1), dim3(256), 0, st, I.nl_cnt.p, (int)n_members, nl_cap, I.mem_base.p, d_total, d_over);
        HIPCHK(hipGetLastError());
    }
    if (crc && !getenv("PG_BGZF_NO_CRC")) {
        hipLaunchKernelGGL(k_crc32, dim3((unsigned)((n_members + 3) / 4)), dim3(256), 0, st, text_d, I.members.p, (int)n_members,
                           I.crc_tab.p, I.status.p);
        HIPCHK(hipGetLastError());
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
                     uint32_t nl_cap, uint64_t text_limit, int64_t *d_total, int32_t *d_over) {
    return inflate_queue(c, st, I, comp_d, n_dw, in_off, in_len, out_len, crc, n_members, text_d, nl_cap, text_limit, d_total, d_over);
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
