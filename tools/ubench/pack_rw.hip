// What a kernel with the pack kernel's memory traffic but none of its arithmetic takes: the north-star rows (10^8 rows of 400
// bytes, 40 GB) read in k_pack3's pattern -- a block of two waves per group of 2048 rows, a lane reads one dword of each of the 32
// rows of a word, the next word's 32 loads are requested before the current word is consumed --, optionally with k_pack3's stores:
// per four words 32 bytes per lane into a called-plane-like buffer (2.5 GB in all), and per nine words 32 bytes per lane into an
// XV-like buffer (1.1 GB).  The loaded dwords are folded into the stored values so that nothing is optimised away.
// FLUSH = 0: reads only; 4: the stores leave as they are produced (k_pack3 today); 16 / 32 / 64: they are staged in LDS and
// written in one burst every FLUSH words (LDS per block 18 / 37 / 74 KB: 8 / 4 / 2 blocks per CU); PADLDS: extra LDS that only
// lowers the number of blocks per CU (what the burst costs in occupancy, separated from what it gains).
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value pack_rw.hip -o pack_rw && ./pack_rw [n_rows]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

constexpr int S = 400, LANES = 100;          // bytes per row, lanes that hold data

// PERIOD > 0 (with FLUSH >= 16): a burst may only START while (s_memrealtime mod PERIOD) < SLOT (ticks of 10 ns, one clock for the whole
// chip): the stores of all blocks are gathered into common time slots, so that between the slots the HBM sees reads only (round 4:
// does clustering the read <-> write turn-arounds chip-wide help where bursts per block did little?)
template <int FLUSH, int PADLDS, int PERIOD = 0, int SLOT = 0>
__global__ __launch_bounds__(128) void k_rw(const int8_t *__restrict__ gt, int64_t n_rows, uint4 *__restrict__ vp,
                                            uint4 *__restrict__ xv, uint32_t *__restrict__ out) {
    constexpr int NV = FLUSH >= 16 ? FLUSH / 4 : 1, NX = FLUSH >= 16 ? (FLUSH + 8) / 9 : 1;     // store events per flush
    __shared__ uint4 lds[(NV + NX) * 2 * LANES + PADLDS / 16];
    const int64_t r0 = (int64_t)blockIdx.x * 2048;
    if (r0 >= n_rows) return;
    const int t = threadIdx.x, h0 = 4 * t;
    const int rows = (int)((n_rows - r0) < 2048 ? (n_rows - r0) : 2048);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(gt + r0 * S), 0, rows * S, 0x00020000);
    const int voff = h0 < S ? h0 : 0x7ffffff0;                       // lanes past the row read zeros (out of range)
    uint32_t d[32], dn[32], acc = 0, hold[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (PADLDS && t == 0) lds[(NV + NX) * 2 * LANES] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 32; ++s) dn[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, s * S, 0);
    int nv = 0, nx = 0, xdone = 0;
    uint4 *vrow = vp + (size_t)blockIdx.x * 16 * 2 * LANES, *xrow = xv + (size_t)blockIdx.x * 8 * 2 * LANES;
    for (int w = 0; w < 64; ++w) {
#pragma unroll
        for (int s = 0; s < 32; ++s) d[s] = dn[s];
#pragma unroll
        for (int s = 0; s < 32; ++s) dn[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, ((w + 1) * 32 + s) * S, 0);
#pragma unroll
        for (int s = 0; s < 32; ++s) hold[s & 7] ^= d[s];
        if (FLUSH && (w & 3) == 3 && t < LANES) {                      // the called plane of four words: 32 bytes per lane
            uint4 *o = FLUSH >= 16 ? lds + nv * 2 * LANES + 2 * t : vrow + (w >> 2) * 2 * LANES + 2 * t;
            o[0] = make_uint4(hold[0], hold[1], hold[2], hold[3]);
            o[1] = make_uint4(hold[4], hold[5], hold[6], hold[7]);
        }
        if (FLUSH && (w & 3) == 3) ++nv;
        if (FLUSH && (w % 9) == 8 && t < LANES) {                      // one dense word of the virtual-site planes: 32 bytes per lane
            uint4 *o = FLUSH >= 16 ? lds + (NV + nx) * 2 * LANES + 2 * t : xrow + (w / 9) * 2 * LANES + 2 * t;
            o[0] = make_uint4(hold[1], hold[0], hold[3], hold[2]);
            o[1] = make_uint4(hold[5], hold[4], hold[7], hold[6]);
        }
        if (FLUSH && (w % 9) == 8) ++nx;
        if (FLUSH >= 16 && (w + 1) % FLUSH == 0) {                     // the burst
            if (PERIOD > 0) {
                if (t == 0)
                    while ((int)(__builtin_amdgcn_s_memrealtime() % (unsigned long long)PERIOD) >= SLOT) __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
            const int v0 = (w + 1 - FLUSH) / 4;
            for (int k = t; k < nv * 2 * LANES; k += 128) vrow[v0 * 2 * LANES + k] = lds[k];
            for (int k = t; k < nx * 2 * LANES; k += 128) xrow[xdone * 2 * LANES + k] = lds[NV * 2 * LANES + k];
            xdone += nx;
            nv = nx = 0;
            __syncthreads();
        }
        acc ^= hold[w & 7];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int FLUSH, int PADLDS, int PERIOD = 0, int SLOT = 0>
void run(const char *what, const int8_t *gt, int64_t n_rows, uint4 *vp, uint4 *xv, uint32_t *out) {
    const unsigned blocks = (unsigned)((n_rows + 2047) / 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, worst = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rw<FLUSH, PADLDS, PERIOD, SLOT>), dim3(blocks), dim3(128), 0, 0, gt, n_rows, vp, xv, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
    }
    printf("  %-64s %.3f - %.3f ms\n", what, best, worst);
}

int main(int argc, char **argv) {
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 100000000ll;
    int8_t *gt; uint4 *vp, *xv; uint32_t *out;
    const size_t blocks = (size_t)((n_rows + 2047) / 2048);
    if (hipMalloc(&gt, (size_t)n_rows * S + 4096) != hipSuccess) { printf("allocation failed\n"); return 1; }
    hipMalloc(&vp, blocks * 16 * 200 * 16); hipMalloc(&xv, blocks * 8 * 200 * 16); hipMalloc(&out, 64);
    hipMemset(gt, 1, (size_t)n_rows * S); hipMemset(vp, 0, blocks * 16 * 200 * 16); hipMemset(xv, 0, blocks * 8 * 200 * 16);
    printf("%.1f GB of rows read in the pack kernel's pattern, %.2f + %.2f GB stored, no arithmetic:\n", (double)n_rows * S / 1e9,
           blocks * 16.0 * 200 * 16 / 1e9, blocks * 7.0 * 200 * 16 / 1e9);
    run<0, 0>("reads only", gt, n_rows, vp, xv, out);
    run<0, 36000>("reads only, 4 blocks per CU", gt, n_rows, vp, xv, out);
    run<0, 72000>("reads only, 2 blocks per CU", gt, n_rows, vp, xv, out);
    run<4, 0>("stores as they are produced (k_pack3 today)", gt, n_rows, vp, xv, out);
    run<4, 36000>("stores as they are produced, 4 blocks per CU", gt, n_rows, vp, xv, out);
    run<4, 72000>("stores as they are produced, 2 blocks per CU", gt, n_rows, vp, xv, out);
    run<16, 0>("bursts every 16 words (18 KB of LDS: 8 blocks per CU)", gt, n_rows, vp, xv, out);
    run<32, 0>("bursts every 32 words (37 KB: 4 blocks per CU)", gt, n_rows, vp, xv, out);
    run<64, 0>("one burst per block (74 KB: 2 blocks per CU)", gt, n_rows, vp, xv, out);
    run<4, 0>("stores as they are produced, again", gt, n_rows, vp, xv, out);
    // time-slot gating of the bursts (ticks of 10 ns)
    run<32, 0, 4000, 500>("bursts every 32 words, slots of 5 us every 40 us", gt, n_rows, vp, xv, out);
    run<32, 0, 2000, 300>("bursts every 32 words, slots of 3 us every 20 us", gt, n_rows, vp, xv, out);
    run<32, 0, 8000, 1000>("bursts every 32 words, slots of 10 us every 80 us", gt, n_rows, vp, xv, out);
    run<32, 0, 4000, 1000>("bursts every 32 words, slots of 10 us every 40 us", gt, n_rows, vp, xv, out);
    run<16, 0, 4000, 500>("bursts every 16 words, slots of 5 us every 40 us", gt, n_rows, vp, xv, out);
    run<16, 0, 2000, 400>("bursts every 16 words, slots of 4 us every 20 us", gt, n_rows, vp, xv, out);
    run<64, 0, 8000, 1000>("one burst per block, slots of 10 us every 80 us", gt, n_rows, vp, xv, out);
    run<32, 0>("bursts every 32 words, no gating, again", gt, n_rows, vp, xv, out);
    return 0;
}
