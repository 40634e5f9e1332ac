// Test infrastructure: the wavefront DEFLATE decoder of the device (genomics_general_amd/csrc/pg_inflate_core.h) compiled as a
// lockstep emulation of its 64 lanes, so that the CPU suite can hold the very source the GPU runs against zlib.  Built on the fly
// by tests/test_inflate.py (g++); never loaded by the product.
#define PG_INFLATE_EMULATE 1
#include "../genomics_general_amd/csrc/pg_inflate_core.h"

#include <cstring>
#include <vector>

// comp: n_comp bytes (any alignment); the deflate stream is comp[in_off : in_off + in_len]; dst: out_len bytes.
extern "C" int pgi_emul_inflate(const uint8_t *comp, uint32_t n_comp, uint32_t in_off, uint32_t in_len, uint8_t *dst, uint32_t out_len) {
    std::vector<uint32_t> words((n_comp + 3) / 4 + 1, 0u);
    if (n_comp) std::memcpy(words.data(), comp, n_comp);
    PgiShared sh;
    std::memset(&sh, 0xAB, sizeof(sh));
    uint8_t sink[64];
    return pgi_member(words.data(), (uint32_t)((n_comp + 3) / 4), in_off, in_len, dst, out_len, sink, &sh);
}
