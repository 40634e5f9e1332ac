// Pairwise counts on the matrix cores, one wave per block: C = V V^T and D = A B^T + B A^T as exact MX fp4 products.
//
// The bit-plane kernels of pg_pair2.hip (k_pairC: v_and + accumulating v_bcnt, k_pairD: v_xor + 2 x v_bitop3 + v_bcnt) sit at
// 0.94 / 0.99 of the measured VALU issue ceiling of their instruction mixes (profiles/r02/valu_rate.txt): the vector ALU cannot
// count pairs faster.  The counts are Gram matrices of 0/1 vectors (D = C - sum_b X_b X_b^T is array_equal to the
// reference's pair loop, genomics.py:903-916, 1219-1221, 1042-1047), and v_mfma_scale_f32_32x32x64_f8f6f4 multiplies
// 32 x 32 x 64 of them per instruction (both operands e2m1, scales 2^0).
//
// The planes stay what k_pack3 writes (1 bit per call / virtual site: no extra HBM bytes); a wave expands the words it needs
// into fp4 nibbles in registers: a site is one nibble, 0b0001 = 0.5, so a product of two set sites is 0.25 and an accumulator
// holds count / 4 -- exact in f32 while count < 2^24, which the launcher guarantees by cutting the word range (integer atomics
// combine the parts):
//     fragment dword m of lane (r, kb) = (word >> m) & 0x11111111,  m = 0..3     (each lane half expands its own words)
// i.e. the lane halves hold different words, in the same (permuted) site order in the A and the B operand, which is all a dot
// product needs.
//
//   k_pairC_fp4   units x units "both called" counts from the called plane Vp:   C(I,J) += V_I V_J^T
//   k_pairD_fp4   haplotype x haplotype differences from the virtual-site planes XV (x = carries the tested allele,
//                 v = called and not excluded):  a = x & v, b = ~x & v,  D(I,J) += a_I b_J^T + b_I a_J^T
//                 ((x_i ^ x_j) & v_i & v_j = a_i b_j + b_i a_j, bit by bit)
// A wave (= a block) owns up to 2 x 2 tiles of 32 x 32 of the upper triangle and keeps their accumulators in registers over its
// part of the window's words.  These kernels serve the shapes the LDS-staged block kernels of pg_pair_tile.hip do not take
// (planes of more than ~340 units per word) and, by default, the D counts.
#include "pg_internal.h"

#include <algorithm>
#include <cstdlib>

namespace {

typedef int v16i __attribute__((ext_vector_type(16)));

#define PG_MFMA_WAVES 4
constexpr int TB = 2;                  // a wave owns up to TB x TB tiles of 32 x 32: 4 x 16 accumulator registers

// s-th task of the upper triangle of T x T tiles: tile rows I0 .. I0+nr-1, tile columns J0 .. J0+nc-1; the first task of a
// block row starts on the diagonal (J0 == I0, nr == nc: the row fragments are the column fragments, the tiles below the diagonal
// are skipped)
__device__ __forceinline__ bool task_decode(int T, int s, int &I0, int &J0, int &nr, int &nc) {
    for (int i0 = 0; i0 < T; i0 += TB) {
        const int n = T - i0, k = (n + TB - 1) / TB;
        if (s < k) {
            I0 = i0;
            nr = n < TB ? n : TB;
            J0 = i0 + s * TB;
            nc = T - J0 < TB ? T - J0 : TB;
            return true;
        }
        s -= k;
    }
    return false;
}

int task_count(int T) {
    int n = 0;
    for (int i0 = 0; i0 < T; i0 += TB) n += (T - i0 + TB - 1) / TB;
    return n;
}

// XCD-aware block -> (window, rest): block b runs on XCD b % 8; all blocks of a window go to one XCD, so the window's planes are
// served by that XCD's L2 (the last n_win % 8 windows are dealt over all XCDs in contiguous runs).  Same dealing as pair_decode.
__device__ __forceinline__ bool win_decode(int per_win, int n_win, int &win, int &rem) {
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

// accumulator tile -> upper triangle of the window's matrix.  C/D layout of the 32x32 MFMA: column = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
__device__ __forceinline__ void store_tile(const v16i &acc, int I, int J, int lane, int n, int diag, int atomic, int32_t *__restrict__ M) {
    const int col = 32 * J + (lane & 31);
    if (col >= n) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * I + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row >= n || row > col || (row == col && !diag)) continue;
        int32_t *dst = &M[(size_t)row * n + col];
        if (atomic) { if (acc[reg]) atomicAdd(dst, acc[reg]); }
        else *dst = acc[reg];
    }
}

__device__ __forceinline__ uint32_t comp(const uint4 &v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

// ---- C ----
template <int NR, int NC, bool DG>
struct WordsC {
    uint4 r[DG ? 1 : NR], c[NC];
    __device__ __forceinline__ void load(const uint4 *__restrict__ prow, const uint4 *__restrict__ pcol) {
        if (!DG) {
#pragma unroll
            for (int i = 0; i < NR; ++i) r[i] = prow[32 * i];
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) c[j] = pcol[32 * j];
    }
};

// ---- D ----
template <int NR, int NC, bool DG>
struct WordsD {
    uint2 r[DG ? 1 : NR], c[NC];
    __device__ __forceinline__ void load(const uint2 *__restrict__ prow, const uint2 *__restrict__ pcol) {
        if (!DG) {
#pragma unroll
            for (int i = 0; i < NR; ++i) r[i] = prow[32 * i];
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) c[j] = pcol[32 * j];
    }
};

// ---- MX fp4 products (v_mfma_scale_f32_32x32x64_f8f6f4, both operands e2m1, scales 2^0) -----------------------------------------
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v8i expand4(uint32_t w) {
    v8i f;                                   // fp4 operands are the first four registers; the others are not read
    f[0] = (int)(w & 0x11111111u);
    f[1] = (int)((w >> 1) & 0x11111111u);
    f[2] = (int)((w >> 2) & 0x11111111u);
    f[3] = (int)((w >> 3) & 0x11111111u);
    return f;
}

// Off the diagonal the two operands of a product are different fragments, and they need not be in the same form: a nibble with
// only bit 0 / 1 / 2 set is 0.5 / 1.0 / 2.0, so the COLUMN fragment keeps bits 0, 1, 2 of every nibble where they are (three
// ANDs; only bit 3, the sign, is shifted down: 5 operations instead of 7) and the ROW fragment puts the same sites into the same
// places with the reciprocal values 2.0 / 1.0 / 0.5 / 2.0: every product of two set sites is 1.0, the accumulator is the count.
// (A diagonal task multiplies a fragment with itself and stays with the form above: count / 4.)
__device__ __forceinline__ v8i expand_col(uint32_t w) {
    v8i f;
    f[0] = (int)(w & 0x11111111u);
    f[1] = (int)(w & 0x22222222u);
    f[2] = (int)(w & 0x44444444u);
    f[3] = (int)((w >> 3) & 0x11111111u);
    return f;
}
__device__ __forceinline__ v8i expand_row(uint32_t w) {
    v8i f;
    f[0] = (int)((w << 2) & 0x44444444u);
    f[1] = (int)(w & 0x22222222u);
    f[2] = (int)((w >> 2) & 0x11111111u);
    f[3] = (int)((w >> 1) & 0x44444444u);
    return f;
}

// (scale operands 0, 0 select the unscaled encoding v_mfma_f32_32x32x64_f8f6f4 -- both scales 2^0 -- : one instruction word pair
// less per product than the v_mfma_ld_scale + v_mfma pair, and no scale register to read)
__device__ __forceinline__ v16f mfma4(const v8i &a, const v8i &b, const v16f &c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
}

template <int UNIT4>     // UNIT4 = 1: the accumulator holds count / 4 (0.5 x 0.5 products, scales 2^0); 0: the count itself
__device__ __forceinline__ void store_tile4(const v16f &acc, int I, int J, int lane, int n, int diag, int atomic, int32_t *__restrict__ M) {
    v16i q;
#pragma unroll
    for (int e = 0; e < 16; ++e) q[e] = UNIT4 ? (int)(acc[e] * 4.0f) : (int)acc[e];
    store_tile(q, I, J, lane, n, diag, atomic, M);
}

// C: the two lane halves work on DIFFERENT word groups -- lanes 0..31 on group g, lanes 32..63 on group g+1 -- so that a 16-byte
// load per lane fetches 1 KiB of distinct words (both halves reading the same group left half of the L1 / TA cycles to
// duplicates) and K step t is simply word t of the lane's own group: any fixed assignment of sites to (step, half) is a valid
// dot product as long as both operands use it.  `live` = this lane's group exists (an odd range ends on a half pair).
template <int NR, int NC, bool DG>
__device__ __forceinline__ void pairC4_pair(const WordsC<NR, NC, DG> &w, bool live, v16f (&acc)[NR][NC]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v8i fc[NC], fr[NR];
#pragma unroll
        for (int j = 0; j < NC; ++j) fc[j] = DG ? expand4(live ? comp(w.c[j], t) : 0u) : expand_col(comp(w.c[j], t));
#pragma unroll
        for (int i = 0; i < NR; ++i) fr[i] = DG ? fc[i] : expand_row(live ? comp(w.r[DG ? 0 : i], t) : 0u);
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int j = 0; j < NC; ++j)
                if (!DG || j >= i) acc[i][j] = mfma4(fr[i], fc[j], acc[i][j]);
    }
}

template <int NR, int NC, bool DG>
__device__ __forceinline__ void pairC4_task(const uint4 *__restrict__ base, int q0, int q1, int NPv, int I0, int J0, int lane,
                                            int n_units, int diag, int atomic, int32_t *__restrict__ Cw) {
    const int r = lane & 31, kb = lane >> 5;
    v16f acc[NR][NC];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // the lane half kb works on group q + kb of the pair that starts at group q
    const uint4 *prow = base + ((size_t)q0 + kb) * NPv + 32 * I0 + r, *pcol = base + ((size_t)q0 + kb) * NPv + 32 * J0 + r;
    int q = q0;
    for (; q + 4 <= q1; q += 4) {                          // two pairs per trip, both requested in front of the first pair's products
        WordsC<NR, NC, DG> wa, wb;
        wa.load(prow, pcol);
        wb.load(prow + 2 * (size_t)NPv, pcol + 2 * (size_t)NPv);
        __builtin_amdgcn_sched_barrier(0);
        pairC4_pair<NR, NC, DG>(wa, true, acc);
        pairC4_pair<NR, NC, DG>(wb, true, acc);
        prow += 4 * (size_t)NPv;
        pcol += 4 * (size_t)NPv;
    }
    for (; q < q1; q += 2) {                               // the last one to three groups: a lane whose group is past the end reads
        const bool live = q + kb < q1;                     // the pair's first group instead and contributes zeros
        const size_t back = live ? 0 : (size_t)NPv;
        WordsC<NR, NC, DG> wa;
        wa.load(prow - back, pcol - back);
        pairC4_pair<NR, NC, DG>(wa, live, acc);
        prow += 2 * (size_t)NPv;
        pcol += 2 * (size_t)NPv;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (!DG || j >= i) store_tile4<DG ? 1 : 0>(acc[i][j], I0 + i, J0 + j, lane, n_units, diag, atomic, Cw);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PG_MFMA_WAVES, PG_MFMA_WAVES)))
void k_pairC_fp4(const uint32_t *__restrict__ Vp, const int64_t *__restrict__ vgoff, int n_win, int T, int ntask, int kparts, int NPv,
                 int n_units, int diag, int32_t *__restrict__ Cmat) {
    int win, rem;
    if (!win_decode(ntask * kparts, n_win, win, rem)) return;
    const int s = rem % ntask, kp = rem / ntask;
    int I0, J0, nr, nc;
    if (!task_decode(T, s, I0, J0, nr, nc)) return;
    const int64_t vg = vgoff[win];
    const int nwq = (int)(vgoff[win + 1] - vg);
    const int q0 = (int)((long long)nwq * kp / kparts), q1 = (int)((long long)nwq * (kp + 1) / kparts);
    const int lane = threadIdx.x & 63, atomic = kparts > 1;
    int32_t *Cw = Cmat + (size_t)win * n_units * n_units;
    if (q1 <= q0) {
        if (!atomic) {
            v16i z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0;
            for (int i = 0; i < nr; ++i)
                for (int j = 0; j < nc; ++j) store_tile(z, I0 + i, J0 + j, lane, n_units, diag, 0, Cw);
        }
        return;
    }
    const uint4 *base = reinterpret_cast<const uint4 *>(Vp) + (size_t)vg * NPv;
#define PG_C_TASK(NR, NC, DG) pairC4_task<NR, NC, DG>(base, q0, q1, NPv, I0, J0, lane, n_units, diag, atomic, Cw)
    if (J0 == I0) {
        if (nr == 2) PG_C_TASK(2, 2, true);
        else PG_C_TASK(1, 1, true);
    } else if (nr == 2) {
        if (nc == 2) PG_C_TASK(2, 2, false);
        else PG_C_TASK(2, 1, false);
    } else {
        if (nc == 2) PG_C_TASK(1, 2, false);
        else PG_C_TASK(1, 1, false);
    }
#undef PG_C_TASK
}

// D: a K step is two slots (64 virtual sites); lane half kb reads slot s + kb (zeros past the end of an odd range)
template <int NR, int NC, bool DG>
__device__ __forceinline__ void pairD4_step(const WordsD<NR, NC, DG> &w, bool live, v16f (&acc)[NR][NC]) {
    v8i ca[NC], cb[NC], ra[NR], rb[NR];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const uint32_t v = (live || !DG) ? w.c[j].y : 0u, a = w.c[j].x & v, b = v ^ a;     // (off the diagonal the rows carry `live`)
        ca[j] = DG ? expand4(a) : expand_col(a);
        cb[j] = DG ? expand4(b) : expand_col(b);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        if (DG) {
            ra[i] = ca[i];
            rb[i] = cb[i];
        } else {
            const uint32_t v = live ? w.r[DG ? 0 : i].y : 0u, a = w.r[DG ? 0 : i].x & v, b = v ^ a;
            ra[i] = expand_row(a);
            rb[i] = expand_row(b);
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (!DG || j >= i) acc[i][j] = mfma4(ra[i], cb[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (!DG || j >= i) acc[i][j] = mfma4(rb[i], ca[j], acc[i][j]);
}

template <int NR, int NC, bool DG>
__device__ __forceinline__ void pairD4_task(const uint2 *__restrict__ xv, int s0, int s1, int NP, int I0, int J0, int lane, int N,
                                            int atomic, int32_t *__restrict__ Dw) {
    const int r = lane & 31, kb = lane >> 5;
    v16f acc[NR][NC];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // the lane half kb works on slot s + kb of the step that starts at slot s
    const uint2 *prow = xv + ((size_t)s0 + kb) * NP + 32 * I0 + r, *pcol = xv + ((size_t)s0 + kb) * NP + 32 * J0 + r;
    int s = s0;
    for (; s + 4 <= s1; s += 4) {                          // two steps per trip, both requested in front of the first step's products
        WordsD<NR, NC, DG> wa, wb;
        wa.load(prow, pcol);
        wb.load(prow + 2 * (size_t)NP, pcol + 2 * (size_t)NP);
        __builtin_amdgcn_sched_barrier(0);
        pairD4_step<NR, NC, DG>(wa, true, acc);
        pairD4_step<NR, NC, DG>(wb, true, acc);
        prow += 4 * (size_t)NP;
        pcol += 4 * (size_t)NP;
    }
    for (; s < s1; s += 2) {                               // the last one to three slots: a lane whose slot is past the end reads the
        const bool live = s + kb < s1;                     // step's first slot instead and contributes zeros
        const size_t back = live ? 0 : (size_t)NP;
        WordsD<NR, NC, DG> wa;
        wa.load(prow - back, pcol - back);
        pairD4_step<NR, NC, DG>(wa, live, acc);
        prow += 2 * (size_t)NP;
        pcol += 2 * (size_t)NP;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (!DG || j >= i) store_tile4<DG ? 1 : 0>(acc[i][j], I0 + i, J0 + j, lane, N, 0, atomic, Dw);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PG_MFMA_WAVES, PG_MFMA_WAVES)))
void k_pairD_fp4(const uint32_t *__restrict__ XV, const int32_t *__restrict__ nw, const int64_t *__restrict__ goff, int n_win, int T,
                 int ntask, int kparts, int NP, int N, int32_t *__restrict__ Dmat, int capg) {
    int win, rem;
    if (!win_decode(ntask * kparts, n_win, win, rem)) return;
    const int s = rem % ntask, kp = rem / ntask;
    int I0, J0, nr, nc;
    if (!task_decode(T, s, I0, J0, nr, nc)) return;
    const uint2 *xv = reinterpret_cast<const uint2 *>(XV + (size_t)goff[win] * capg * PG_XV_PLANES * (size_t)NP);
    const int capw = (int)(goff[win + 1] - goff[win]) * capg;
    const int n_all = __builtin_amdgcn_readfirstlane(nw[win]);
    const int n_words = n_all < capw ? n_all : capw;
    const int a = (int)((long long)n_words * kp / kparts), b = (int)((long long)n_words * (kp + 1) / kparts);
    int32_t *Dw = Dmat + (size_t)win * N * N;
    const int lane = threadIdx.x & 63, atomic = kparts > 1;
    if (b <= a) {
        if (!atomic) {
            v16i z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0;
            for (int i = 0; i < nr; ++i)
                for (int j = 0; j < nc; ++j) store_tile(z, I0 + i, J0 + j, lane, N, 0, 0, Dw);
        }
        return;
    }
#define PG_D_TASK(NR, NC, DG) pairD4_task<NR, NC, DG>(xv, a, b, NP, I0, J0, lane, N, atomic, Dw)
    if (J0 == I0) {
        if (nr == 2) PG_D_TASK(2, 2, true);
        else PG_D_TASK(1, 1, true);
    } else if (nr == 2) {
        if (nc == 2) PG_D_TASK(2, 2, false);
        else PG_D_TASK(2, 1, false);
    } else {
        if (nc == 2) PG_D_TASK(1, 2, false);
        else PG_D_TASK(1, 1, false);
    }
#undef PG_D_TASK
}

// extra cut of the word range across blocks: wanted when windows x segments cannot give every SIMD a few waves
int pick_parts(int n_win, int ntask, int64_t steps_per_window, int min_steps) {
    const int64_t waves = (int64_t)n_win * ntask;
    int kp = 1;
    while (kp < 64 && waves * kp < 4096 && steps_per_window / (kp * 2) >= min_steps) kp *= 2;
    return kp;
}

// fp4 path: an f32 accumulator holds count / 4 exactly while count < 2^24; no part of any window may see more sites than that
// (2^23 keeps a margin); the parts are combined by integer atomics
int exact_parts(int64_t max_sites_per_window) { return (int)((max_sites_per_window + (1 << 23) - 1) >> 23); }

}  // namespace

void pg_launch_pairC_mfma(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, int NPv, int n_units, int diag,
                          int64_t avg_wq, int64_t max_sites, int32_t *Cmat) {
    if (n_win <= 0 || n_units <= 0) return;
    const int T = (n_units + 31) / 32, ntask = task_count(T);
    int kparts = std::max(pick_parts(n_win, ntask, avg_wq, 8), exact_parts(max_sites));
    // L2 locality: the tasks of a window start together and read the same words, but they drift apart (diagonal and edge tasks
    // issue fewer products per step) and an XCD runs some 50 windows at once against 4 MB of L2 -- PMC: 9.1 GB fetched per
    // north-star launch for a 2.5 GB plane.  Parts of at most 1 MiB of plane end before the drift matters (measured on the
    // north-star shape: 1.27-1.32 ms in one part, 1.16-1.18 in two or four, 1.30 in eight, 2.3 in sixteen, zeroing included)
    const int64_t plane_bytes = avg_wq * (int64_t)NPv * 16;
    kparts = std::max(kparts, (int)std::min<int64_t>(4, (plane_bytes + (1 << 20) - 1) >> 20));
    if (kparts > 1) (void)hipMemsetAsync(Cmat, 0, (size_t)n_win * n_units * n_units * 4, st);
    const int64_t blocks = (int64_t)((n_win + 7) / 8) * ntask * kparts * 8;
    hipLaunchKernelGGL(k_pairC_fp4, dim3((unsigned)blocks), dim3(64), 0, st, Vp, vgoff, n_win, T, ntask, kparts, NPv, n_units, diag, Cmat);
}

void pg_launch_pairD_mfma(hipStream_t st, const uint32_t *XV, const int32_t *nw, const int64_t *goff, int n_win, int NP, int N,
                          int64_t avg_words, int64_t max_vsites, int32_t *Dmat, int capg) {
    if (n_win <= 0 || N <= 0) return;
    const int T = (N + 31) / 32, ntask = task_count(T);
    const int kparts = std::max(pick_parts(n_win, ntask, avg_words, 8), exact_parts(max_vsites));
    if (kparts > 1) (void)hipMemsetAsync(Dmat, 0, (size_t)n_win * N * N * 4, st);
    const int64_t blocks = (int64_t)((n_win + 7) / 8) * ntask * kparts * 8;
    hipLaunchKernelGGL(k_pairD_fp4, dim3((unsigned)blocks), dim3(64), 0, st, XV, nw, goff, n_win, T, ntask, kparts, NP, N, Dmat, capg);
}
