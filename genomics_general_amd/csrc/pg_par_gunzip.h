// ONE gzip stream on MANY host threads.
//
// `gzip file.geno` gives a single deflate stream (README.md:106 of the reference; read there by gzip.open, genomics.py:1917): no
// member boundaries, every block's matches reach into the 32 KiB before it.  A single thread of pg_fast_inflate.h does about 2 GB/s
// of `.geno` text; the north star as plain gzip (81 GB of text) would still take most of a minute.  This file decodes such a
// stream in pieces that run side by side, the way pugz / rapidgzip do:
//   1. the compressed bytes of a batch are cut into as many chunks as there are threads; every thread looks for the first deflate
//      block that starts in its chunk -- a dynamic-block header that parses (code-length code complete, literal / length and
//      distance codes valid, an end-of-block code) and whose first thousand symbols decode;
//   2. every thread decodes from its block start, NOT knowing the 32 KiB of text in front of it: the output is 16 bits per byte, a
//      byte copied from the unknown window becomes a marker naming its place there (pgfi::inflate_t<uint16_t, true>); it stops in
//      front of the block at which another thread started (or where the stream ends);
//   3. the chunks are chained by their bit positions (a chunk whose start no predecessor arrived at is dropped: its "block start"
//      was none); the windows are handed on -- the last 32 KiB of a chunk, its markers replaced from the window before it -- one
//      chunk after the other (32 KiB each: microseconds), then all chunks are turned into bytes in parallel.
// The member's CRC-32 over the resulting text is checked by the caller (pg_inflate.hip), so a wrong guess anywhere cannot pass;
// whatever goes wrong (no block start found, a decoder error in a chained chunk) makes the caller decode that stretch serially.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include <zlib.h>

#include "pg_crc32_fast.h"
#include "pg_fast_inflate.h"

namespace pgpar {

// the first bit position in [from, to) at which a dynamic deflate block (not the last one of the stream) starts, else ~0
static uint64_t find_block(const uint8_t *base, uint64_t len, uint64_t from, uint64_t to, pgfi::State &st, std::vector<uint16_t> &scratch) {
    if (len < 16) return ~0ull;
    const uint64_t last = (len - 12) * 8;                                               // (room for the loads below)
    if (to > last) to = last;
    scratch.resize(4096);
    for (uint64_t b = from; b < to; ++b) {
        const uint64_t v = pgfi::load64(base + (b >> 3)) >> (b & 7u);                  // 57 valid bits
        if ((v & 7u) != 4u) continue;                                                   // BFINAL 0, BTYPE 2
        if (((v >> 3) & 31u) > 29u || ((v >> 8) & 31u) > 29u) continue;                 // HLIT, HDIST
        // the code-length code must be complete: sum over its lengths of 2^(7 - length) = 128
        const int hclen = (int)((v >> 13) & 15u) + 4;
        const uint64_t w = pgfi::load64(base + ((b + 17) >> 3)) >> ((b + 17) & 7u);     // 57 bits = the 19 lengths
        uint32_t kraft = 0;
        for (int k = 0; k < hclen; ++k) {
            const uint32_t l = (uint32_t)(w >> (3 * k)) & 7u;
            if (l) kraft += 128u >> l;
        }
        if (kraft != 128u) continue;
        // the whole header and the block's first symbols
        pgfi::start_at(st, base, len, b);
        st.win_len = 32768;
        st.n_stops = 0;
        uint64_t n = 0;
        const int rc = pgfi::inflate_t<uint16_t, true>(st, scratch.data(), 1024 + 300, &n, 0);
        if (rc == pgfi::NEED_OUTPUT || (rc == pgfi::STREAM_END && n > 0)) return b;
        // (a block that ends before a thousand symbols and is followed by a valid one also passes: rc would still be NEED_OUTPUT or
        // the decoder is inside the next block; an error anywhere on the way refuses the candidate)
    }
    return ~0ull;
}

struct Chunk {
    uint64_t start = ~0ull, end = 0;
    uint16_t *out = nullptr;                    // (malloc'd, kept from batch to batch: no zero-filling, pages stay mapped)
    uint64_t cap = 0;
    uint64_t n = 0;
    int rc = pgfi::ERR_DATA;
    uint32_t crc = 0;                           // CRC-32 of the chunk's bytes (emit)
    bool grow(uint64_t want) {
        if (want <= cap) return true;
        void *p = realloc(out, (size_t)want * 2);
        if (!p) return false;
        out = static_cast<uint16_t *>(p);
        cap = want;
        return true;
    }
};

// A batch: about n_threads * chunk_bytes compressed bytes from bit `start` (a block boundary), decoded side by side.  decode()
// leaves the chained chunks, their offsets in the batch's text and the windows in front of them; emit() turns them into bytes --
// the chunks that fit into `direct` go there, the others into the spill buffer -- and takes the CRC-32 of every chunk on the way.
struct Batch {
    std::vector<Chunk> ch;
    std::vector<uint64_t> found, stops, off;
    std::vector<int> chain;
    std::vector<std::vector<uint8_t>> win;
    uint64_t total = 0, end_bit = 0;
    uint32_t w_len_out = 0;
    bool ended = false;
    uint8_t *spill = nullptr;
    uint64_t spill_cap = 0, spill_len = 0;
    std::vector<std::vector<uint16_t>> scratch;
    ~Batch() {
        for (auto &c : ch) free(c.out);
        free(spill);
    }

    // 0, or < 0: the caller decodes serially from `start`
    int decode(const uint8_t *base, uint64_t len, uint64_t start, const uint8_t *window, uint32_t w_len, int n_threads, uint64_t chunk_bytes) {
        const int n = std::max(2, n_threads);
        static const bool trace = getenv("PG_GZIP_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        auto ms = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3; };
        double t_find = 0, t_dec = 0;
        if ((int)ch.size() != n) {
            for (auto &c : ch) free(c.out);
            ch.assign((size_t)n, Chunk());
        }
        for (auto &c : ch) { c.start = ~0ull; c.end = 0; c.n = 0; c.rc = pgfi::ERR_DATA; }
        found.assign((size_t)n + 1, ~0ull);
        found[0] = start;
        const uint64_t byte0 = start >> 3;
        // 1. block starts
        {
            std::vector<std::thread> th;
            for (int k = 1; k <= n; ++k)
                th.emplace_back([&, k]() {
                    const uint64_t a = (byte0 + (uint64_t)k * chunk_bytes) * 8, b = (byte0 + (uint64_t)(k + 1) * chunk_bytes) * 8;
                    if (a / 8 + 16 >= len) return;
                    std::vector<uint16_t> scr;
                    pgfi::State *st = new pgfi::State();
                    found[(size_t)k] = find_block(base, len, a, b, *st, scr);
                    delete st;
                });
            for (auto &x : th) x.join();
        }
        t_find = ms();
        stops.clear();
        for (int k = 1; k <= n; ++k)
            if (found[(size_t)k] != ~0ull) stops.push_back(found[(size_t)k]);
        std::sort(stops.begin(), stops.end());
        // 2. the chunks, side by side
        {
            std::vector<std::thread> th;
            for (int k = 0; k < n; ++k) {
                if (found[(size_t)k] == ~0ull) continue;
                th.emplace_back([&, k]() {
                    Chunk &c = ch[(size_t)k];
                    c.start = found[(size_t)k];
                    pgfi::State *st = new pgfi::State();
                    pgfi::start_at(*st, base, len, c.start);
                    st->win_len = 32768;
                    const uint64_t *first = std::upper_bound(stops.data(), stops.data() + stops.size(), c.start);
                    st->stops = first;
                    st->n_stops = (int)(stops.data() + stops.size() - first);
                    // (a chunk that runs on and on -- no block start was found behind it, or its own start was none -- is cut off: the
                    // caller then decodes serially)
                    const uint64_t limit = std::max<uint64_t>(chunk_bytes, 1u << 20) * 200;
                    if (!c.grow(std::max<uint64_t>(chunk_bytes * 30, 1u << 20))) { c.rc = pgfi::ERR_DATA; delete st; return; }
                    for (;;) {
                        uint64_t got = 0;
                        c.rc = pgfi::inflate_t<uint16_t, true>(*st, c.out, c.cap, &got, c.n);
                        c.n += got;
                        if (c.rc != pgfi::NEED_OUTPUT) break;
                        if (c.n >= limit || !c.grow(c.cap * 2)) { c.rc = pgfi::ERR_DATA; break; }
                    }
                    if (c.rc == pgfi::STOPPED) c.end = st->stopped_at;
                    else if (c.rc == pgfi::STREAM_END) {
                        const uint8_t *t = nullptr;
                        c.end = pgfi::stream_tail(*st, &t) ? (uint64_t)(t - base) * 8 : ~0ull;
                        if (c.end == ~0ull) c.rc = pgfi::ERR_INPUT;
                    }
                    delete st;
                });
            }
            for (auto &x : th) x.join();
        }
        t_dec = ms();
        // 3. the chain
        chain.clear();
        int cur = 0;
        ended = false;
        for (;;) {
            const Chunk &c = ch[(size_t)cur];
            if (c.rc != pgfi::STOPPED && c.rc != pgfi::STREAM_END) return -1;
            chain.push_back(cur);
            if (c.rc == pgfi::STREAM_END) { ended = true; break; }
            int next = -1;
            for (int k = cur + 1; k <= n; ++k)
                if (found[(size_t)k] == c.end) { next = k; break; }
            if (next < 0) return -1;
            if (next == n) break;                                                      // the next batch starts there
            cur = next;
        }
        // the windows in front of the chained chunks, handed on one after the other (32 KiB each)
        if (win.size() < chain.size() + 1) win.resize(chain.size() + 1);
        for (auto &w : win) w.resize(32768);
        memcpy(win[0].data(), window, 32768);
        off.assign(chain.size() + 1, 0);
        uint64_t wl = w_len;
        total = 0;
        for (size_t i = 0; i < chain.size(); ++i) {
            const Chunk &c = ch[(size_t)chain[i]];
            const uint8_t *w = win[i].data();
            uint8_t *nw = win[i + 1].data();
            const uint64_t tail = std::min<uint64_t>(c.n, 32768);
            if (tail < 32768) memcpy(nw, w + tail, (size_t)(32768 - tail));             // what stays of the old window
            for (uint64_t k = 0; k < tail; ++k) {
                const uint16_t e = c.out[(size_t)(c.n - tail + k)];
                nw[32768 - tail + k] = e < 0x8000u ? (uint8_t)e : w[e & 0x7FFFu];
            }
            wl = std::min<uint64_t>(32768, wl + c.n);
            off[i + 1] = off[i] + c.n;
            total += c.n;
        }
        w_len_out = (uint32_t)wl;
        end_bit = ch[(size_t)chain.back()].end;
        if (trace) fprintf(stderr, "PG_GZIP_TRACE batch: %d of %d chunks chained, %.1f MB of text; block starts %.1f ms, decode %.1f, chain + windows %.1f\n",
                           (int)chain.size(), n, total / 1e6, t_find, t_dec - t_find, ms() - t_dec);
        return 0;
    }

    const uint8_t *window_out() const { return win[chain.size()].data(); }

    // the chunks into bytes: chunk i goes to direct + off[i] when it ends inside direct_cap, else into the spill buffer (the chunks
    // behind the first that does not fit, in order).  Returns the number of bytes that went to `direct`; spill_len = the others.
    uint64_t emit(uint8_t *direct, uint64_t direct_cap) {
        const auto t_e = std::chrono::steady_clock::now();
        size_t first_spill = chain.size();
        for (size_t i = 0; i < chain.size(); ++i)
            if (off[i + 1] > direct_cap) { first_spill = i; break; }
        const uint64_t direct_len = off[first_spill];
        spill_len = total - direct_len;
        if (spill_len > spill_cap) {
            free(spill);
            spill = static_cast<uint8_t *>(malloc((size_t)spill_len));
            spill_cap = spill ? spill_len : 0;
        }
        std::vector<std::thread> th;
        for (size_t i = 0; i < chain.size(); ++i)
            th.emplace_back([&, i]() {
                Chunk &c = ch[(size_t)chain[i]];
                const uint8_t *w = win[i].data();
                uint8_t *dst = i < first_spill ? direct + off[i] : spill + (off[i] - direct_len);
                const uint16_t *src = c.out;
                // a piece at a time, so that the checksum reads what the conversion has just written (still in the cache)
                uint32_t crc = 0;
                for (uint64_t a = 0; a < c.n; a += 1u << 18) {
                    const uint64_t b = std::min<uint64_t>(c.n, a + (1u << 18));
                    uint64_t k = a;
#if defined(__x86_64__)
                    // sixteen elements at a time: no marker among them (the usual case behind a chunk's first 32 KiB) -> their low bytes
                    for (; k + 16 <= b; k += 16) {
                        const __m128i lo = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k));
                        const __m128i hi = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k + 8));
                        if (_mm_movemask_epi8(_mm_or_si128(lo, hi)) & 0xAAAA) {
                            for (int j = 0; j < 16; ++j) {
                                const uint16_t e = src[k + j];
                                dst[k + j] = e < 0x8000u ? (uint8_t)e : w[e & 0x7FFFu];
                            }
                        } else {
                            _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + k), _mm_packus_epi16(lo, hi));
                        }
                    }
#endif
                    for (; k < b; ++k) {
                        const uint16_t e = src[k];
                        dst[k] = e < 0x8000u ? (uint8_t)e : w[e & 0x7FFFu];
                    }
                    crc = pg_crc32(crc, dst + a, (size_t)(b - a));
                }
                c.crc = crc;
            });
        for (auto &x : th) x.join();
        if (getenv("PG_GZIP_TRACE")) fprintf(stderr, "PG_GZIP_TRACE emit: %.1f MB direct, %.1f MB spilled, %.1f ms\n", direct_len / 1e6, spill_len / 1e6,
                                             std::chrono::duration<double>(std::chrono::steady_clock::now() - t_e).count() * 1e3);
        return direct_len;
    }
};

}  // namespace pgpar
