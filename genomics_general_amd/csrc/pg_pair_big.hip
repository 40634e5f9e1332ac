// Called counts on the matrix cores with one wave per SIMD: k_pairC_big.
//
// The same exact products as pg_pair_tile.hip / pg_pair_mfma.hip (C = V V^T over the called plane as MX fp4 nibbles;
// genomics.py:1042-1047), for planes of up to 224 units (7 tiles of 32: the shapes of BASELINE.json's popgenWindows configs).
// What the counters say about the other two forms (HISTORY.md section 4): they spend 5 - 9 vector instructions per matrix
// instruction and fill 77 % of the chip's issue slots while the matrix pipes idle half the time.  Here a wave owns up to 14 tiles
// (accumulators in the accumulator half of the register file), so a fragment it expands feeds up to seven products -- 1.75 to 3.5
// vector instructions per product --, and the whole main loop is one generated, hand-scheduled instruction stream
// (gen_pairc_big.py -> pg_pairc_big.inc: LDS-DMA ring, fragment reads, expansions between the products), because with a single
// wave on a SIMD the matrix pipe is busy only while that wave's next instruction is a product whose operands are ready.
// A block is one window part: one wave up to 10 tiles, two waves beyond (28 tiles at 200 units), the plane fetched once.
#include "pg_internal.h"
#include "pg_pairc_big.inc"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bool win_decode(int per_win, int n_win, int &win, int &rem) {       // as in pg_pair_tile.hip
    const int xcd = blockIdx.x & 7;
    const int v = blockIdx.x >> 3;
    const int full = n_win >> 3;
    if (v < full * per_win) {
        win = (v / per_win) * 8 + xcd;
        rem = v % per_win;
        return true;
    }
    const int total = (n_win & 7) * per_win, q = (total + 7) >> 3;
    const int vt = v - full * per_win, lin = xcd * q + vt;
    if (vt >= q || lin >= total) return false;
    win = full * 8 + lin / per_win;
    rem = lin % per_win;
    return true;
}

// accumulator tile (count / 4) -> upper triangle of the window's matrix (column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
__device__ __forceinline__ void store_tile(const v16f &acc, int I, int J, int lane, int n, int diag, int atomic, int32_t *__restrict__ M) {
    const int col = 32 * J + (lane & 31);
    if (col >= n) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * I + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row >= n || row > col || (row == col && !diag)) continue;
        const int32_t v = (int32_t)(acc[reg] * 4.0f);
        int32_t *dst = &M[(size_t)row * n + col];
        if (atomic) { if (v) atomicAdd(dst, v); }
        else *dst = v;
    }
}

template <int T> struct Tiles;
#define PG_BIG_TILES(T_)                                                              \
    template <> struct Tiles<T_> {                                                    \
        static constexpr int W = PG_CBIG_WAVES_T##T_;                                  \
    };                                                                                \
    __device__ const signed char g_tiles_##T_[2][14][2] = PG_CBIG_TILES_T##T_;
PG_BIG_TILES(1) PG_BIG_TILES(2) PG_BIG_TILES(3) PG_BIG_TILES(4) PG_BIG_TILES(5) PG_BIG_TILES(6) PG_BIG_TILES(7)

template <int T> __device__ __forceinline__ const signed char (*tiles_of())[14][2];
#define PG_BIG_TILES_OF(T_) template <> __device__ __forceinline__ const signed char (*tiles_of<T_>())[14][2] { return g_tiles_##T_; }
PG_BIG_TILES_OF(1) PG_BIG_TILES_OF(2) PG_BIG_TILES_OF(3) PG_BIG_TILES_OF(4) PG_BIG_TILES_OF(5) PG_BIG_TILES_OF(6) PG_BIG_TILES_OF(7)

#define PG_BIG_RUN(ASM)                                                                                                                  \
    asm volatile(ASM                                                                                                                     \
                 : [a0] "+a"(acc[0]), [a1] "+a"(acc[1]), [a2] "+a"(acc[2]), [a3] "+a"(acc[3]), [a4] "+a"(acc[4]), [a5] "+a"(acc[5]),          \
                   [a6] "+a"(acc[6]), [a7] "+a"(acc[7]), [a8] "+a"(acc[8]), [a9] "+a"(acc[9]), [a10] "+a"(acc[10]), [a11] "+a"(acc[11]),      \
                   [a12] "+a"(acc[12]), [a13] "+a"(acc[13])                                                                                \
                 : [ga] "v"(ga), [lrd] "v"(lrd), [kb] "v"(kb), [lds] "s"(ring), [stride] "s"(stride), [npair] "s"(npair), [qrem] "s"(qrem),   \
                   [wave] "s"(wave)                                                                                                        \
                 : PG_CBIG_CLOBBERS)

template <int T>
__global__ __launch_bounds__(64 * Tiles<T>::W) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_pairC_big(const uint32_t *__restrict__ Vp, const int64_t *__restrict__ vgoff, int n_win, int kparts, int NPv, int n_units,
                 int diag, int32_t *__restrict__ Cmat) {
    extern __shared__ uint4 lds[];                            // ring: PG_CBIG_RING_PAIRS x T KiB
    int win, kp;
    if (!win_decode(kparts, n_win, win, kp)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int64_t vg = vgoff[win];
    const int nwq = (int)(vgoff[win + 1] - vg);
    const int q0 = (int)((long long)nwq * kp / kparts), q1 = (int)((long long)nwq * (kp + 1) / kparts);
    const int qrem = q1 - q0, npair = (qrem + 1) / 2;
    const int atomic = kparts > 1;
    int32_t *Cw = Cmat + (size_t)win * n_units * n_units;
    v16f acc[14];
#pragma unroll
    for (int n = 0; n < 14; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;
    if (npair > 0) {
        const uint4 *ga = reinterpret_cast<const uint4 *>(Vp) + ((size_t)vg + q0 + kb) * NPv + r;
        const uint32_t ring = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)lds;
        const uint32_t lrd = ring + 16u * lane;
        const int stride = 2 * NPv * 16;
        if constexpr (T == 1) PG_BIG_RUN(PG_CBIG_ASM_T1);
        else if constexpr (T == 2) PG_BIG_RUN(PG_CBIG_ASM_T2);
        else if constexpr (T == 3) PG_BIG_RUN(PG_CBIG_ASM_T3);
        else if constexpr (T == 4) PG_BIG_RUN(PG_CBIG_ASM_T4);
        else if constexpr (T == 5) PG_BIG_RUN(PG_CBIG_ASM_T5);
        else if constexpr (T == 6) PG_BIG_RUN(PG_CBIG_ASM_T6);
        else PG_BIG_RUN(PG_CBIG_ASM_T7);
    }
    if (npair > 0 || !atomic) {                               // (an empty window: the counts are zero and nobody else writes them)
        const signed char (*tl)[14][2] = tiles_of<T>();
#pragma unroll
        for (int n = 0; n < 14; ++n) {
            const int i = tl[wave][n][0], j = tl[wave][n][1];
            if (i >= 0) store_tile(acc[n], i, j, lane, n_units, diag, atomic, Cw);
        }
    }
}

int pick_parts(int n_win, int waves_per_win, int64_t steps_per_window, int min_steps) {
    const int64_t waves = (int64_t)n_win * waves_per_win;
    int kp = 1;
    while (kp < 64 && waves * kp < 2048 && steps_per_window / (kp * 2) >= min_steps) kp *= 2;
    return kp;
}

template <int T>
void launch(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, int kparts, int NPv, int n_units, int diag, int32_t *Cmat) {
    const int64_t blocks = (int64_t)((n_win + 7) / 8) * kparts * 8;
    hipLaunchKernelGGL((k_pairC_big<T>), dim3((unsigned)blocks), dim3(64 * Tiles<T>::W), (size_t)PG_CBIG_RING_PAIRS * T * 1024, st, Vp, vgoff,
                       n_win, kparts, NPv, n_units, diag, Cmat);
}

}  // namespace

// planes of up to 7 tiles of 32 units (the ring holds the T tile rows of a pair of groups, whatever the plane's stride); the
// default form of the called counts at these sizes (PG_PAIR_TILE without a 'b' turns it off: pg_pair_tile.hip)
bool pg_pair_big_fits(int NPv, int n_units) {
    const char *sel = getenv("PG_PAIR_TILE");
    if (sel && !strchr(sel, 'b')) return false;
    const int T = (n_units + 31) / 32;
    return T >= 1 && T <= 7 && NPv >= 32 * T;
}

void pg_launch_pairC_big(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, int NPv, int n_units, int diag,
                         int64_t avg_wq, int64_t max_sites, int32_t *Cmat) {
    if (n_win <= 0 || n_units <= 0) return;
    const int T = (n_units + 31) / 32;
    const int W = T * (T + 1) / 2 <= 14 ? 1 : 2;
    // an f32 accumulator holds count / 4 exactly while count < 2^24: parts below 2^23 sites; more parts when the windows are few
    const int kparts = std::max(pick_parts(n_win, W, avg_wq / 2, 32), (int)((max_sites + (1 << 23) - 1) >> 23));
    if (kparts > 1) (void)hipMemsetAsync(Cmat, 0, (size_t)n_win * n_units * n_units * 4, st);
    switch (T) {
        case 1: launch<1>(st, Vp, vgoff, n_win, kparts, NPv, n_units, diag, Cmat); break;
        case 2: launch<2>(st, Vp, vgoff, n_win, kparts, NPv, n_units, diag, Cmat); break;
        case 3: launch<3>(st, Vp, vgoff, n_win, kparts, NPv, n_units, diag, Cmat); break;
        case 4: launch<4>(st, Vp, vgoff, n_win, kparts, NPv, n_units, diag, Cmat); break;
        case 5: launch<5>(st, Vp, vgoff, n_win, kparts, NPv, n_units, diag, Cmat); break;
        case 6: launch<6>(st, Vp, vgoff, n_win, kparts, NPv, n_units, diag, Cmat); break;
        default: launch<7>(st, Vp, vgoff, n_win, kparts, NPv, n_units, diag, Cmat); break;
    }
}
