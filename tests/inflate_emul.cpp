// Test infrastructure: the wavefront DEFLATE decoder of the device (genomics_general_amd/csrc/pg_inflate_core.h) compiled as a
// lockstep emulation of its 64 lanes, so that the CPU suite can hold the very source the GPU runs against zlib.  Built on the fly
// by tests/test_inflate.py (g++); never loaded by the product.
#define PG_INFLATE_EMULATE 1
#include "../genomics_general_amd/csrc/pg_inflate_core.h"

#include <cstring>
#include <vector>

// the line feeds the decoder reports beside the text (pgi_member's nl_list): set the capacity and the limit first, read the list after
static uint32_t g_nl_cap = 0, g_nl_lim = 0xFFFFFFFFu, g_nl_n = 0;
static std::vector<uint16_t> g_nl;
static int g_crc_on = 0;
static uint32_t g_want_crc = 0;
// the CRC-32 the decoder is to hold the text against (on = 0: no check)
extern "C" void pgi_emul_crc_setup(int on, uint32_t want) { g_crc_on = on; g_want_crc = want; }
extern "C" void pgi_emul_nl_setup(uint32_t cap, uint32_t lim) { g_nl_cap = cap; g_nl_lim = lim; }
extern "C" uint32_t pgi_emul_nl_result(uint16_t *out, uint32_t cap) {
    for (uint32_t k = 0; k < g_nl.size() && k < cap; ++k) out[k] = g_nl[k];
    return g_nl_n;
}

// comp: n_comp bytes (any alignment); the deflate stream is comp[in_off : in_off + in_len]; dst: out_len bytes.  The member is
// inflated to an address that is `misalign` bytes behind a 16-byte boundary (the kernel flushes its ring in aligned 16-byte pieces;
// a member's text starts at any byte of the block's text) and copied to dst.
extern "C" int pgi_emul_inflate_at(const uint8_t *comp, uint32_t n_comp, uint32_t in_off, uint32_t in_len, uint8_t *dst, uint32_t out_len,
                                   int misalign) {
    std::vector<uint32_t> words((n_comp + 3) / 4 + 1, 0u);
    if (n_comp) std::memcpy(words.data(), comp, n_comp);
    PgiShared sh;
    std::memset(&sh, 0xAB, sizeof(sh));
    alignas(16) uint8_t sink[128];
    std::vector<uint16_t> nl(g_nl_cap ? g_nl_cap : 1);
    uint32_t nl_n = 0;
    std::vector<PgiU4> out((size_t)out_len / 16 + 4);
    uint8_t *at = reinterpret_cast<uint8_t *>(out.data()) + (misalign & 15);
    std::memset(out.data(), 0xCD, out.size() * 16);
    static uint32_t crc_tab[PGI_CRC_TAB];
    static bool crc_made = false;
    if (!crc_made) { pgi_make_crc_tables(crc_tab); crc_made = true; }
    const int rc = pgi_member(words.data(), (uint32_t)((n_comp + 3) / 4), in_off, in_len, at, out_len, sink, &sh, g_nl_cap ? nl.data() : nullptr,
                              g_nl_cap, g_nl_lim, &nl_n, g_crc_on ? crc_tab : nullptr, g_want_crc);
    g_nl_n = rc ? 0u : nl_n;
    g_nl.assign(nl.begin(), nl.begin() + (rc ? 0u : (nl_n < g_nl_cap ? nl_n : g_nl_cap)));
    // nothing in front of the member's first byte or behind its last may have been touched (its neighbours' text lives there)
    for (int k = 0; k < (misalign & 15); ++k)
        if (reinterpret_cast<uint8_t *>(out.data())[k] != 0xCD) return 1 << 20;
    for (size_t k = (size_t)(misalign & 15) + out_len; k < out.size() * 16; ++k)
        if (reinterpret_cast<uint8_t *>(out.data())[k] != 0xCD) return 1 << 21;
    if (out_len) std::memcpy(dst, at, out_len);
    return rc;
}

extern "C" int pgi_emul_inflate(const uint8_t *comp, uint32_t n_comp, uint32_t in_off, uint32_t in_len, uint8_t *dst, uint32_t out_len) {
    return pgi_emul_inflate_at(comp, n_comp, in_off, in_len, dst, out_len, (int)((in_off * 7u + out_len) & 15u));
}
