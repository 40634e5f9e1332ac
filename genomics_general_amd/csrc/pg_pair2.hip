// v2 pairwise pipeline: the pairwise matrices split into the two terms that need different amounts of work.
//
//   C[i][j] = sum over ALL sites of v_i & v_j                      -> k_pairC on the "called" plane only (2 VALU / 32 pair-sites)
//   D[i][j] = sum over POLYMORPHIC sites of differ(i,j) & v_i & v_j -> k_pairD on compacted allele planes (5 VALU / 32 pair-sites)
//
// A site whose called haplotypes all carry the same allele adds the same amount to C and to "same allele", i.e. nothing to D
// (genomics.py:903-905, 1219-1221: numHamming counts differences among jointly called sites).  k_pack2 therefore detects
// polymorphic sites (>= 2 alleles present among the called haplotypes of the window's slots) while it transposes, and
// bit-compacts only those into the allele planes.  Data-dependent, exact, and the algorithmic pair-sites stay the denominator
// of every reported rate.
//
// Layouts (uint32 words, 32 sites per word):
//   Vp[(vgoff[b] + wq) * NPv + unit][4]        called plane, 4 consecutive words of one unit contiguous (one 16-byte load per
//                                               lane per 128 sites; 16 rows x 4 words = 4 x s_load_dwordx16)
//   XV[((goff[b] + g) * PG_GROUP + k) * 5 + p][NP]  compacted planes of group g (64 input words): p = 0..3 X_a (allele a called),
//                                               p = 4 V (called);  nw[goff[b]+g] = words used (0..64)
// differ & both called  ==  OR_a (X_a,i & Y_a,j), Y_a = V ^ X_a: row operands X in SGPRs, column operands Y in VGPRs.
#include "pg_internal.h"

typedef __attribute__((address_space(4))) const uint32_t CU32;

// ------------------------------------------------------------------------------------------------------
// k_pack2.  Thread = 4 haplotype slots; block = all slots of one compaction group (64 input words) of one window.
//   SWAR transposition: two sites are merged per dword (one-hot nibbles -> one byte), so each allele plane gathers two
//   site bits per op; after 4 merged pairs a byte holds 8 site bits of one haplotype (bit order inside a word is a fixed
//   permutation, identical for every haplotype and plane, which is all popcount needs).
//   DIP = 1: every individual is diploid and owns slots (2k,2k+1); the called plane is written per INDIVIDUAL (k_pairC then
//   works on n_hap/2 units, 4x fewer pairs).  A window in which the two haplotypes of some individual differ in calledness
//   raises *mismatch; the host then redoes the batch with DIP = 0.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bgather(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, int k) {
    return ((a0 >> (8 * k)) & 0xFFu) | (((a1 >> (8 * k)) & 0xFFu) << 8) | (((a2 >> (8 * k)) & 0xFFu) << 16) |
           (((a3 >> (8 * k)) & 0xFFu) << 24);
}

// 32 sites x 4 haplotypes -> x[a][k] (allele plane a of haplotype k); sites >= ns are zero bits.
// The 32 loads are raw buffer loads: descriptor base = the word's first site row (wave-uniform, SGPRs), scalar offset =
// row * S, lane offset = h0 -> no VALU address arithmetic at all (a flat load needs a 64-bit add per load).  The descriptor's
// size is ns rows, so rows past the end of a window read as zero without a select.
__device__ __forceinline__ void load_word(const int8_t *__restrict__ rows, int h0, int S, int ns, uint32_t x[4][4]) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(rows), 0, ns * S, 0x00020000);
    uint32_t acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t d[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) d[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, h0, (q * 8 + s) * S, 0);
        uint32_t a0 = 0u, a1 = 0u, a2 = 0u, a3 = 0u;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const uint32_t t = d[2 * pr] | (d[2 * pr + 1] << 4);
            a0 = (a0 << 1) | (t & 0x11111111u);
            a1 = (a1 << 1) | ((t >> 1) & 0x11111111u);
            a2 = (a2 << 1) | ((t >> 2) & 0x11111111u);
            a3 = (a3 << 1) | ((t >> 3) & 0x11111111u);
        }
        acc[0][q] = a0; acc[1][q] = a1; acc[2][q] = a2; acc[3][q] = a3;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) x[p][k] = bgather(acc[p][0], acc[p][1], acc[p][2], acc[p][3], k);
}

// OR over the 64 lanes of a wave with DPP row shifts / broadcasts (six VALU ops, no LDS traffic); the total ends in lane 63.
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#define PG_DPP_OR(ctrl, rmask) v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false)
    PG_DPP_OR(0x111, 0xf);      // row_shr:1
    PG_DPP_OR(0x112, 0xf);      // row_shr:2
    PG_DPP_OR(0x114, 0xf);      // row_shr:4
    PG_DPP_OR(0x118, 0xf);      // row_shr:8   -> lane 15 of each row holds the row total
    PG_DPP_OR(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    PG_DPP_OR(0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
#undef PG_DPP_OR
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

#define PG_DENSE_BITS 14      // a word with at least this many polymorphic sites is emitted whole instead of bit by bit

// More than 1024 haplotype slots do not fit one block: k_presence (same loads and transposition, grid.z = blocks of 1024 slots)
// first ORs the per-site allele-presence words of all slot blocks into pres[word][4]; k_pack2<.,.,PRES=1> then reads them.
__global__ __launch_bounds__(256) void k_presence(const int8_t *__restrict__ gt, int S, const int64_t *__restrict__ win_lo,
                                                  const int64_t *__restrict__ win_hi, const int64_t *__restrict__ goff,
                                                  uint32_t *__restrict__ pres) {
    const int b = blockIdx.y, g = blockIdx.x;
    const int64_t lo = win_lo[b], hi = win_hi[b];
    const int W = (int)((hi - lo + 31) >> 5);
    const int w_begin = g * PG_GROUP;
    if (w_begin >= W) return;
    const int w_end = (w_begin + PG_GROUP < W) ? w_begin + PG_GROUP : W;
    const int h0 = 4 * (blockIdx.z * 256 + threadIdx.x);
    uint32_t *dst = pres + (size_t)(goff[b] + g) * PG_GROUP * 4u;
    for (int w = w_begin; w < w_end; ++w) {
        uint32_t x[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) x[p][k] = 0u;
        if (h0 < S) {
            const int64_t s0 = lo + 32ll * w;
            const int ns = (int)((hi - s0) < 32 ? (hi - s0) : 32);
            load_word(gt + s0 * (int64_t)S, h0, S, ns, x);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t pr = wave_or(x[p][0] | x[p][1] | x[p][2] | x[p][3]);
            if ((threadIdx.x & 63) == 0 && pr) atomicOr(&dst[(size_t)(w - w_begin) * 4u + p], pr);
        }
    }
}

template <int TPB, int DIP, int PRES>
__global__ __launch_bounds__(TPB) void k_pack2(const int8_t *__restrict__ gt, int S, const int64_t *__restrict__ win_lo,
                                               const int64_t *__restrict__ win_hi, const int64_t *__restrict__ goff,
                                               const int64_t *__restrict__ vgoff, uint32_t *__restrict__ Vp, int NPv,
                                               uint32_t *__restrict__ XV, int NP, int32_t *__restrict__ nw,
                                               int32_t *__restrict__ mismatch, const uint32_t *__restrict__ pres) {
    constexpr int NWAVE = TPB / 64;
    __shared__ uint32_t sh_pres[2][NWAVE][4];
    const int b = blockIdx.y, g = blockIdx.x;
    const int64_t lo = win_lo[b], hi = win_hi[b];
    const int W = (int)((hi - lo + 31) >> 5);
    const int w_begin = g * PG_GROUP;
    if (w_begin >= W) return;
    const int w_end = (w_begin + PG_GROUP < W) ? w_begin + PG_GROUP : W;
    const int t = blockIdx.z * TPB + threadIdx.x;
    const int h0 = 4 * t;
    const bool has_data = h0 < S;            // pad threads (h0 >= S) still write zero planes up to NP
    const bool in_np = h0 < NP;
    const int u0 = DIP ? 2 * t : h0;         // first unit of the called plane owned by this thread
    const bool in_npv = u0 < NPv;
    uint32_t out[5][4];
#pragma unroll
    for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) out[p][k] = 0u;
    int cnt = 0, nflush = 0, parity = 0;
    uint32_t bad = 0u;
    uint32_t *xv_base = XV + (size_t)(goff[b] + g) * PG_GROUP * 5u * (size_t)NP;
    const int64_t vg_base = vgoff[b] + (int64_t)(w_begin >> 2);
    for (int wq = 0; 4 * wq + w_begin < w_end; ++wq) {
        uint32_t vhold[4][4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int w = w_begin + 4 * wq + k4;
            uint32_t x[4][4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = 0u;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) x[p][k] = 0u;
            const bool live = w < w_end;            // block-uniform
            if (live && has_data) {
                const int64_t s0 = lo + 32ll * w;
                const int ns = (int)((hi - s0) < 32 ? (hi - s0) : 32);
                load_word(gt + s0 * (int64_t)S, h0, S, ns, x);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = x[0][k] | x[1][k] | x[2][k] | x[3][k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) vhold[k][k4] = v[k];
            if (DIP) bad |= (v[0] ^ v[1]) | (v[2] ^ v[3]);
            if (live) {
                // alleles present among called haplotypes, per site, across the whole block
                uint32_t pr[4];
                if (PRES) {
                    const uint32_t *src = pres + ((size_t)(goff[b] + g) * PG_GROUP + (size_t)(w - w_begin)) * 4u;
#pragma unroll
                    for (int p = 0; p < 4; ++p) pr[p] = src[p];
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p) pr[p] = wave_or(x[p][0] | x[p][1] | x[p][2] | x[p][3]);
                }
                if (!PRES && NWAVE > 1) {
                    if ((t & 63) == 0) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) sh_pres[parity][t >> 6][p] = pr[p];
                    }
                    __syncthreads();
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        uint32_t a = 0u;
#pragma unroll
                        for (int wv = 0; wv < NWAVE; ++wv) a |= sh_pres[parity][wv][p];
                        pr[p] = a;
                    }
                    parity ^= 1;
                }
                uint32_t m = (pr[0] & pr[1]) | (pr[0] & pr[2]) | (pr[0] & pr[3]) | (pr[1] & pr[2]) | (pr[1] & pr[3]) | (pr[2] & pr[3]);
                m = __builtin_amdgcn_readfirstlane(m);
                const bool dense = __builtin_popcount(m) >= PG_DENSE_BITS;
                if (dense) {
                    // flush the partial word, then emit this word verbatim (monomorphic sites add nothing to D)
                    if (cnt) {
                        if (in_np) {
                            uint32_t *o = xv_base + (size_t)nflush * 5u * (size_t)NP + h0;
#pragma unroll
                            for (int p = 0; p < 5; ++p)
                                *reinterpret_cast<uint4 *>(o + (size_t)p * NP) = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
                        }
#pragma unroll
                        for (int p = 0; p < 5; ++p)
#pragma unroll
                            for (int k = 0; k < 4; ++k) out[p][k] = 0u;
                        cnt = 0;
                        ++nflush;
                    }
                    if (in_np) {
                        uint32_t *o = xv_base + (size_t)nflush * 5u * (size_t)NP + h0;
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            *reinterpret_cast<uint4 *>(o + (size_t)p * NP) = make_uint4(x[p][0], x[p][1], x[p][2], x[p][3]);
                        *reinterpret_cast<uint4 *>(o + (size_t)4 * NP) = make_uint4(v[0], v[1], v[2], v[3]);
                    }
                    ++nflush;
                    m = 0u;
                }
                while (m) {
                    const int bit = __builtin_ctz(m);
                    m &= m - 1u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) out[p][k] = (out[p][k] << 1) | ((x[p][k] >> bit) & 1u);
                        out[4][k] = (out[4][k] << 1) | ((v[k] >> bit) & 1u);
                    }
                    if (++cnt == 32) {
                        if (in_np) {
                            uint32_t *o = xv_base + (size_t)nflush * 5u * (size_t)NP + h0;
#pragma unroll
                            for (int p = 0; p < 5; ++p)
                                *reinterpret_cast<uint4 *>(o + (size_t)p * NP) = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
                        }
#pragma unroll
                        for (int p = 0; p < 5; ++p)
#pragma unroll
                            for (int k = 0; k < 4; ++k) out[p][k] = 0u;
                        cnt = 0;
                        ++nflush;
                    }
                }
            }
        }
        if (in_npv) {
            if (DIP) {
                uint32_t *o = Vp + ((size_t)(vg_base + wq) * NPv + u0) * 4u;
                *reinterpret_cast<uint4 *>(o) = make_uint4(vhold[0][0], vhold[0][1], vhold[0][2], vhold[0][3]);
                *reinterpret_cast<uint4 *>(o + 4) = make_uint4(vhold[2][0], vhold[2][1], vhold[2][2], vhold[2][3]);
            } else {
                uint32_t *o = Vp + ((size_t)(vg_base + wq) * NPv + u0) * 4u;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<uint4 *>(o + 4 * k) = make_uint4(vhold[k][0], vhold[k][1], vhold[k][2], vhold[k][3]);
            }
        }
    }
    if (cnt) {
        if (in_np) {
            uint32_t *o = xv_base + (size_t)nflush * 5u * (size_t)NP + h0;
#pragma unroll
            for (int p = 0; p < 5; ++p)
                *reinterpret_cast<uint4 *>(o + (size_t)p * NP) = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
        }
        ++nflush;
    }
    if (t == 0) nw[goff[b] + g] = nflush;
    if (DIP && bad) atomicOr(mismatch, 1);
}

template <int DIP>
static void launch_pack2(hipStream_t st, int threads, dim3 grid, const int8_t *gt, int S, const int64_t *win_lo,
                         const int64_t *win_hi, const int64_t *goff, const int64_t *vgoff, uint32_t *Vp, int NPv, uint32_t *XV,
                         int NP, int32_t *nw, int32_t *mismatch, uint32_t *pres) {
    if (threads <= 64)
        hipLaunchKernelGGL((k_pack2<64, DIP, 0>), grid, dim3(64), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres);
    else if (threads <= 128)
        hipLaunchKernelGGL((k_pack2<128, DIP, 0>), grid, dim3(128), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres);
    else if (threads <= 256)
        hipLaunchKernelGGL((k_pack2<256, DIP, 0>), grid, dim3(256), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres);
    else {
        grid.z = (threads + 255) / 256;
        hipLaunchKernelGGL(k_presence, grid, dim3(256), 0, st, gt, S, win_lo, win_hi, goff, pres);
        hipLaunchKernelGGL((k_pack2<256, DIP, 1>), grid, dim3(256), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres);
    }
}

// pres: scratch of total_groups * PG_GROUP * 4 words, only used (and zeroed here) when there are more than 1024 slots
void pg_launch_pack2(hipStream_t st, const int8_t *gt, int S, const int64_t *win_lo, const int64_t *win_hi,
                     const int64_t *goff, const int64_t *vgoff, int n_win, int max_groups, int64_t total_groups, uint32_t *Vp,
                     int NPv, uint32_t *XV, int NP, int32_t *nw, int dip, int32_t *mismatch, uint32_t *pres) {
    if (n_win <= 0 || max_groups <= 0) return;
    const int threads = NP / 4;
    if (threads > 256) (void)hipMemsetAsync(pres, 0, (size_t)total_groups * PG_GROUP * 16u, st);
    dim3 grid(max_groups, n_win);
    if (dip) launch_pack2<1>(st, threads, grid, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres);
    else launch_pack2<0>(st, threads, grid, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres);
}

// ------------------------------------------------------------------------------------------------------
// Task decoding shared by k_pairC / k_pairD.  1-D XCD-aware grid: block b runs on XCD b % 8; all waves of a window go
// to one XCD so its planes are served by that XCD's L2.
// ------------------------------------------------------------------------------------------------------
struct PairCtx {
    int win, row0, nsub, col0, lower, lane, ks;
};

// ksplit > 1: the word range of a window is cut into ksplit parts handled by different waves (more waves in flight
// when there are few windows); partial counts are then combined with integer atomics (exact, order independent).
__device__ __forceinline__ bool pair_decode(const PgTask2 *__restrict__ tasks, int n_tasks, int tasks_wg, int ksplit, int n_win,
                                            PairCtx &c) {
    const int xcd = blockIdx.x & 7;
    const int v = blockIdx.x >> 3;
    const int per_win = tasks_wg * ksplit;
    c.win = (v / per_win) * 8 + xcd;
    if (c.win >= n_win) return false;
    const int rem = v % per_win;
    c.ks = rem / tasks_wg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    const int t = (rem % tasks_wg) * 4 + wave;
    if (t >= n_tasks) return false;
    const PgTask2 tk = tasks[t];
    c.row0 = __builtin_amdgcn_readfirstlane(tk.row0);
    c.nsub = __builtin_amdgcn_readfirstlane(tk.nsub);
    c.col0 = __builtin_amdgcn_readfirstlane(tk.col0);
    c.lower = __builtin_amdgcn_readfirstlane(tk.lower);
    return true;
}

// store acc[r] for pair (row0+r, j): upper tasks keep i<j (i<=j with diag), lower tasks keep j<i (j<=i) and write (j,i)
template <int R>
__device__ __forceinline__ void pair_store(const uint32_t (&acc)[R], int row0, int j, int n, int lower, int diag, int atomic,
                                           int32_t *__restrict__ M) {
    if (j >= n) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = row0 + r;
        if (i >= n) continue;
        const bool keep = lower ? (j < i || (diag && i == j)) : (i < j || (diag && i == j));
        if (!keep) continue;
        int32_t *dst = lower ? &M[(size_t)j * n + i] : &M[(size_t)i * n + j];
        if (atomic) { if (acc[r]) atomicAdd(dst, (int32_t)acc[r]); }
        else *dst = (int32_t)acc[r];
    }
}

// ------------------------------------------------------------------------------------------------------
// k_pairC: units x units "both called" counts.  Wave = 8*NSUB rows (SGPR operands) x 64 columns, 4 words per iteration.
// ------------------------------------------------------------------------------------------------------
template <int NSUB>
__device__ __forceinline__ void pairC_body(const uint32_t *__restrict__ Vp, int64_t vg0, int nwq, int NPv, const PairCtx &c,
                                           int n_units, int diag, int atomic, int32_t *__restrict__ Cw) {
    constexpr int R = 8 * NSUB;
    uint32_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0u;
    const int j = c.col0 + c.lane;
    const uint32_t *base = Vp + (size_t)vg0 * NPv * 4u;
    for (int wq = 0; wq < nwq; ++wq) {
        const uint32_t *pw = base + (size_t)wq * NPv * 4u;
        const uint4 jv = *reinterpret_cast<const uint4 *>(pw + (size_t)j * 4u);
        const CU32 *pr = (const CU32 *)(pw + (size_t)c.row0 * 4u);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] += __popc(pr[4 * r + 0] & jv.x);
            acc[r] += __popc(pr[4 * r + 1] & jv.y);
            acc[r] += __popc(pr[4 * r + 2] & jv.z);
            acc[r] += __popc(pr[4 * r + 3] & jv.w);
        }
    }
    pair_store<R>(acc, c.row0, j, n_units, c.lower, diag, atomic, Cw);
}

__global__ __launch_bounds__(256) void k_pairC(const uint32_t *__restrict__ Vp, const int64_t *__restrict__ vgoff, int n_win,
                                               const PgTask2 *__restrict__ tasks, int n_tasks, int tasks_wg, int ksplit, int NPv,
                                               int n_units, int diag, int32_t *__restrict__ Cmat) {
    PairCtx c;
    if (!pair_decode(tasks, n_tasks, tasks_wg, ksplit, n_win, c)) return;
    const int64_t vg_all = vgoff[c.win];
    const int nwq_all = (int)(vgoff[c.win + 1] - vg_all);
    const int q0 = (int)((long long)nwq_all * c.ks / ksplit), q1 = (int)((long long)nwq_all * (c.ks + 1) / ksplit);
    int32_t *Cw = Cmat + (size_t)c.win * n_units * n_units;
    if (c.nsub == 1) pairC_body<1>(Vp, vg_all + q0, q1 - q0, NPv, c, n_units, diag, ksplit > 1, Cw);
    else pairC_body<2>(Vp, vg_all + q0, q1 - q0, NPv, c, n_units, diag, ksplit > 1, Cw);
}

// waves wanted in flight: 256 CUs x 4 SIMDs x 8
static int pick_ksplit(int n_win, int n_tasks, int64_t steps_per_wave, int min_steps) {
    int64_t waves = (int64_t)n_win * n_tasks;
    int ks = 1;
    while (ks < 16 && waves * ks < 8192 && steps_per_wave / (ks * 2) >= min_steps) ks *= 2;
    return ks;
}

void pg_launch_pairC(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, const PgTask2 *tasks, int n_tasks,
                     int NPv, int n_units, int diag, int64_t avg_wq, int32_t *Cmat) {
    if (n_win <= 0 || n_tasks <= 0) return;
    const int tasks_wg = (n_tasks + 3) / 4;
    const int ks = pick_ksplit(n_win, n_tasks, avg_wq, 24);
    if (ks > 1) (void)hipMemsetAsync(Cmat, 0, (size_t)n_win * n_units * n_units * 4, st);
    const int64_t blocks = (int64_t)((n_win + 7) / 8) * tasks_wg * ks * 8;
    hipLaunchKernelGGL(k_pairC, dim3((unsigned)blocks), dim3(256), 0, st, Vp, vgoff, n_win, tasks, n_tasks, tasks_wg, ks, NPv,
                       n_units, diag, Cmat);
}

// ------------------------------------------------------------------------------------------------------
// k_pairD: haplotype x haplotype difference counts over the compacted polymorphic words of a window.
//   differ & both called == OR_a (X_a,i & Y_a,j) with Y_a,j = called_j & ~X_a,j = V_j ^ X_a,j, formed once per column word.
// ------------------------------------------------------------------------------------------------------
template <int NSUB>
__device__ __forceinline__ void pairD_body(const uint32_t *__restrict__ XV, const int32_t *__restrict__ nw, int64_t g0, int ng,
                                           int NP, const PairCtx &c, int N, int atomic, int32_t *__restrict__ Dw) {
    constexpr int R = 8 * NSUB;
    uint32_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0u;
    const int j = c.col0 + c.lane;
    const size_t wstride = (size_t)5 * NP;
    for (int g = 0; g < ng; ++g) {
        const int n = __builtin_amdgcn_readfirstlane(nw[g0 + g]);
        const uint32_t *gb = XV + (size_t)(g0 + g) * PG_GROUP * wstride;
        for (int w = 0; w < n; ++w) {
            const uint32_t *pw = gb + (size_t)w * wstride;
            const uint32_t jv = pw[(size_t)4 * NP + j];
            const uint32_t y0 = jv ^ pw[j], y1 = jv ^ pw[(size_t)NP + j], y2 = jv ^ pw[(size_t)2 * NP + j],
                           y3 = jv ^ pw[(size_t)3 * NP + j];
            const CU32 *pr = (const CU32 *)(pw + c.row0);
#pragma unroll
            for (int r = 0; r < R; ++r)
                acc[r] += __popc((pr[r] & y0) | (pr[NP + r] & y1) | (pr[2 * NP + r] & y2) | (pr[3 * NP + r] & y3));
        }
    }
    pair_store<R>(acc, c.row0, j, N, c.lower, 0, atomic, Dw);
}

__global__ __launch_bounds__(256) void k_pairD(const uint32_t *__restrict__ XV, const int32_t *__restrict__ nw,
                                               const int64_t *__restrict__ goff, int n_win, const PgTask2 *__restrict__ tasks,
                                               int n_tasks, int tasks_wg, int ksplit, int NP, int N, int32_t *__restrict__ Dmat) {
    PairCtx c;
    if (!pair_decode(tasks, n_tasks, tasks_wg, ksplit, n_win, c)) return;
    const int64_t g_all = goff[c.win];
    const int ng_all = (int)(goff[c.win + 1] - g_all);
    const int a = (int)((long long)ng_all * c.ks / ksplit), b = (int)((long long)ng_all * (c.ks + 1) / ksplit);
    int32_t *Dw = Dmat + (size_t)c.win * N * N;
    if (c.nsub == 1) pairD_body<1>(XV, nw, g_all + a, b - a, NP, c, N, ksplit > 1, Dw);
    else pairD_body<2>(XV, nw, g_all + a, b - a, NP, c, N, ksplit > 1, Dw);
}

void pg_launch_pairD(hipStream_t st, const uint32_t *XV, const int32_t *nw, const int64_t *goff, int n_win,
                     const PgTask2 *tasks, int n_tasks, int NP, int N, int64_t avg_groups, int32_t *Dmat) {
    if (n_win <= 0 || n_tasks <= 0) return;
    const int tasks_wg = (n_tasks + 3) / 4;
    const int ks = pick_ksplit(n_win, n_tasks, avg_groups, 4);
    if (ks > 1) (void)hipMemsetAsync(Dmat, 0, (size_t)n_win * N * N * 4, st);
    const int64_t blocks = (int64_t)((n_win + 7) / 8) * tasks_wg * ks * 8;
    hipLaunchKernelGGL(k_pairD, dim3((unsigned)blocks), dim3(256), 0, st, XV, nw, goff, n_win, tasks, n_tasks, tasks_wg, ks, NP, N,
                       Dmat);
}

// ------------------------------------------------------------------------------------------------------
// k_expand: full symmetric [N][N] matrices (zero diagonal) for pg_pairwise from the upper triangles of D (haplotype level)
// and C (unit level: cshift = 1 when the units are diploid individuals).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_expand(const int32_t *__restrict__ Cmat, const int32_t *__restrict__ Dmat, int N, int cN,
                                                int cshift, int32_t *__restrict__ Cfull, int32_t *__restrict__ Dfull) {
    const int32_t *Cw = Cmat + (size_t)blockIdx.y * cN * cN;
    const int32_t *Dw = Dmat + (size_t)blockIdx.y * N * N;
    int32_t *Co = Cfull + (size_t)blockIdx.y * N * N;
    int32_t *Do = Dfull + (size_t)blockIdx.y * N * N;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N * N; idx += gridDim.x * blockDim.x) {
        const int i = idx / N, j = idx - i * N;
        if (i == j) { Co[idx] = 0; Do[idx] = 0; continue; }
        const int a = i < j ? i : j, b = i < j ? j : i;
        Co[idx] = Cw[(size_t)(a >> cshift) * cN + (b >> cshift)];
        Do[idx] = Dw[(size_t)a * N + b];
    }
}

void pg_launch_expand(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                      int32_t *Cfull, int32_t *Dfull) {
    if (n_win <= 0) return;
    int bx = (N * N + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_expand, dim3(bx, n_win), dim3(256), 0, st, Cmat, Dmat, N, cN, cshift, Cfull, Dfull);
}
