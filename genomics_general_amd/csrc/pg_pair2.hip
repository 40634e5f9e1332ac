// v2 pairwise pipeline: the pairwise matrices split into the two terms that need different amounts of work.
//
//   C[i][j] = sum over ALL sites of v_i & v_j                       -> k_pairC on the "called" plane only (2 VALU / 32 pair-sites)
//   D[i][j] = sum over POLYMORPHIC sites of differ(i,j) & v_i & v_j -> k_pairD on the polymorphic-site planes (3 VALU / 32 pair-sites)
// (The planes are written by the pack kernels of this file; by default they are consumed by the matrix-core kernels of
// pg_pair_mfma.hip -- the same two sums as exact products -- and the popcount kernels k_pairC / k_pairD below run with
// PG_PAIR_VALU=1.)
//
// A site whose called haplotypes all carry the same allele adds the same amount to C and to "same allele", i.e. nothing to D
// (genomics.py:903-905, 1219-1221: numHamming counts differences among jointly called sites).  k_pack2 therefore detects
// polymorphic sites (>= 2 alleles present among the called haplotypes of the window's slots) while it builds the called
// plane, and transposes those sites only.  Data-dependent, exact, and the algorithmic pair-sites stay the denominator of every
// reported rate.
//
// Every polymorphic site becomes k-1 VIRTUAL BIALLELIC sites (k = alleles present, a_0 < a_1 < ... in A,C,G,T order):
//   virtual site t:  x = "carries a_t",  v = "called and carries none of a_0 .. a_(t-1)"
//   differ(i,j) & both called  ==  sum over t of (x_i ^ x_j) & v_i & v_j
// (a pair with alleles (a_0, other) differs at t = 0 and is excluded afterwards; a pair without a_0 is compared at t = 1; ...),
// so k_pairD needs two planes and three VALU ops per 32 pair-sites whatever the number of alleles; real data is almost entirely
// biallelic (k - 1 = 1).
//
// Layouts (uint32 words, 32 sites per word):
//   Vp[(vgoff[b] + wq) * NPv + unit][4]   called plane, 4 consecutive words of one unit contiguous (one 16-byte load per lane per
//                                          128 sites; 8 rows x 4 words = 2 x s_load_dwordx16); two padding word groups at the end
//   XV[(goff[b] * PG_XV_CAP + k) * 2 * NP + 2 * hap + p]   dense words of the virtual sites of window b, k = 0 .. nw[b]-1, in no
//                                          particular order (a compaction group of 2048 sites takes the next free word of its
//                                          window whenever it has collected 32 virtual sites, and once more for its last,
//                                          partial word); p = 0: x, p = 1: v; the planes of a haplotype are adjacent (rows: two
//                                          s_load_dwordx16 per 16 haplotypes, column: one 8-byte load); two padding words at the end
#include "pg_internal.h"

typedef __attribute__((address_space(4))) const uint32_t CU32;

// ------------------------------------------------------------------------------------------------------
// k_pack2.  Thread = 4 haplotype slots (one dword of a site row); block = all slots of one compaction group (64 input
// words = 2048 sites) of one window.  Two phases, both on the same raw-buffer descriptor (base = first row of the group,
// scalar offset = row * S, lane offset = h0: no VALU address arithmetic; rows past the end of the window read as zero):
//
//   phase A, every word (32 rows): two rows are merged per dword (one-hot nibbles -> one byte), four merged dwords are
//     byte-transposed with v_perm_b32 so that a dword holds 8 sites of ONE haplotype; from that (i) the called bit of every
//     site (nibble != 0) -> called plane Vp, (ii) the OR over all haplotypes -> per-site allele presence nibbles, reduced
//     over the wave with DPP and over the block through LDS.  A site is polymorphic when its presence nibble has >= 2 bits;
//     that test and the list bookkeeping run on the scalar unit.
//   phase B, every 32 virtual sites: their rows are loaded again (scalar row offsets read from a lane-resident list; they were
//     touched a few words ago), the four allele planes are transposed the same way, and x / v of each virtual site are
//     selected from them with per-word masks (which allele a virtual site tests and which it excludes: computed per list
//     entry from the word's presence nibbles, turned into masks with v_cmp ballots) and stored as one dense word of XV.
//     k_pairD therefore only ever sees polymorphic sites, and the expensive transposition runs on ~10 % of the rows of
//     typical whole-genome data instead of all of them.
//
//   The bit order inside a produced word is a fixed permutation of the 32 sites, identical for every haplotype and plane,
//   which is all the popcounts of k_pairC / k_pairD need.
//   DIP = 1: every individual is diploid and owns slots (2k,2k+1); the called plane is written per INDIVIDUAL (k_pairC then
//   works on n_hap/2 units, 4x fewer pairs).  A window in which the two haplotypes of some individual differ in calledness
//   raises *mismatch; the host then redoes the batch with DIP = 0.
// ------------------------------------------------------------------------------------------------------
// 16-byte store of plane words (non-temporal stores and non-temporal row loads were measured: no gain / slower)
__device__ __forceinline__ void store16(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    *reinterpret_cast<uint4 *>(p) = make_uint4(a, b, c, d);
}

// 4x4 byte transpose: r[k] = { t0.byte k, t1.byte k, t2.byte k, t3.byte k }
__device__ __forceinline__ void btrans4(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t r[4]) {
    const uint32_t lo01 = __builtin_amdgcn_perm(t1, t0, 0x05010400u), hi01 = __builtin_amdgcn_perm(t1, t0, 0x07030602u);
    const uint32_t lo23 = __builtin_amdgcn_perm(t3, t2, 0x05010400u), hi23 = __builtin_amdgcn_perm(t3, t2, 0x07030602u);
    r[0] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
    r[1] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
    r[2] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
    r[3] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
}

// The 32 row dwords of a word (issued one word ahead of their use, see k_pack2).  ro[s] = s * S are computed once per kernel
// and kept in SGPRs; the word's base goes into the lane offset (two VALU adds per word), so the loads need no scalar
// address arithmetic at all.
struct RowOff { int v[16]; };

__device__ __forceinline__ RowOff make_row_off(int S) {
    RowOff ro;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        int t = s * S;
        asm volatile("" : "+s"(t));          // opaque: stays in an SGPR instead of being rematerialised per load
        ro.v[s] = t;
    }
    return ro;
}

__device__ __forceinline__ void word_load(__amdgpu_buffer_rsrc_t rsrc, int h0, int S, int row0, const RowOff &ro, uint32_t d[32]) {
    const int v0 = h0 + row0 * S, v1 = v0 + 16 * S;
#pragma unroll
    for (int s = 0; s < 16; ++s) d[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, v0, ro.v[s], 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) d[16 + s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, v1, ro.v[s], 0);
}

// Phase A for one word, bit 4j+q of every produced word <-> site q*8+j:
//   v[k]  = called bits of haplotype h0+k,   pa[a] = sites at which allele a occurs among this lane's four haplotypes.
__device__ __forceinline__ void word_called_presence(const uint32_t d[32], uint32_t v[4], uint32_t pa[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t *e = d + 8 * q;
        uint32_t r[4];
        btrans4(e[0] | (e[1] << 4), e[2] | (e[3] << 4), e[4] | (e[5] << 4), e[6] | (e[7] << 4), r);
        const uint32_t o = r[0] | r[1] | r[2] | r[3];                    // 8 sites x presence nibble
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const uint32_t piece = (a >= q ? (o >> (a - q)) : (o << (q - a))) & (0x11111111u << q);
            pa[a] = q ? (pa[a] | piece) : piece;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t u = r[k] | (r[k] >> 1);
            const uint32_t c = (u | (u >> 2)) & 0x11111111u;
            v[k] = q ? (v[k] | (c << q)) : c;
        }
    }
}

// Phase B: lane i of vlist holds list entry i: group-relative row | (ordinal t of the virtual site << 16); entries past the end
// hold a row beyond the descriptor and read as zero.  SA[a]: entries (bits, in output order) whose tested allele is a (A, C or
// G: the highest allele present is never tested); SE[a]: entries that exclude allele a (A or C).
//   x[0][k] = x, x[1][k] = v of haplotype h0+k over the 32 listed virtual sites.
__device__ __forceinline__ void poly_word(__amdgpu_buffer_rsrc_t rsrc, int h0, int S, uint32_t vlist, const uint32_t SA[3],
                                          const uint32_t SE[2], uint32_t x[PG_XV_PLANES][4]) {
    uint32_t al[4][4];                                                   // [allele][haplotype]: carries the allele
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t d[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = __builtin_amdgcn_readlane((int)vlist, q * 8 + s) & 0xffff;
            d[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, h0, row * S, 0);
        }
        uint32_t r[4];
        btrans4(d[0] | (d[1] << 4), d[2] | (d[3] << 4), d[4] | (d[5] << 4), d[6] | (d[7] << 4), r);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint32_t piece = (a >= q ? (r[k] >> (a - q)) : (r[k] << (q - a))) & (0x11111111u << q);
                al[a][k] = q ? (al[a][k] | piece) : piece;
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t ex = (al[0][k] & SE[0]) | (al[1][k] & SE[1]);
        x[0][k] = (al[0][k] & SA[0]) | (al[1][k] & SA[1]) | (al[2][k] & SA[2]);
        x[1][k] = (al[0][k] | al[1][k] | al[2][k] | al[3][k]) & ~ex;
    }
}

// OR over the 64 lanes of a wave with DPP row shifts / broadcasts (six VALU ops per value, no LDS traffic); four values go
// through the steps in lockstep so that no DPP read follows the write of its own operand (no s_nop wait states).
__device__ __forceinline__ void wave_or4(uint32_t v[4]) {
#define PG_DPP_OR4(ctrl, rmask)                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        v[i] |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[i], ctrl, rmask, 0xf, false)
    PG_DPP_OR4(0x111, 0xf);      // row_shr:1
    PG_DPP_OR4(0x112, 0xf);      // row_shr:2
    PG_DPP_OR4(0x114, 0xf);      // row_shr:4
    PG_DPP_OR4(0x118, 0xf);      // row_shr:8   -> lane 15 of each row holds the row total
    PG_DPP_OR4(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    PG_DPP_OR4(0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
#undef PG_DPP_OR4
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (uint32_t)__builtin_amdgcn_readlane((int)v[i], 63);
}

// sites (bits) at which at least two of the four alleles occur: scalar unit, 7 ops
__device__ __forceinline__ uint32_t poly_mask(const uint32_t p[4]) {
    return (p[0] & p[1]) | (p[2] & p[3]) | ((p[0] ^ p[1]) & (p[2] ^ p[3]));
}

// sites at which at least three / all four alleles occur
__device__ __forceinline__ uint32_t tri_mask(const uint32_t p[4]) {
    return (p[0] & p[1] & (p[2] | p[3])) | (p[2] & p[3] & (p[0] | p[1]));
}
__device__ __forceinline__ uint32_t quad_mask(const uint32_t p[4]) { return p[0] & p[1] & p[2] & p[3]; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t group_rsrc(const int8_t *gt, int S, int64_t first_row, int nrows) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(gt + first_row * (int64_t)S), 0, nrows * S, 0x00020000);
}

// More than 1024 haplotype slots do not fit one block: k_presence (phase A only, grid.z = blocks of 1024 slots) first ORs the
// presence nibbles of all slot blocks into pres[word][4]; k_pack2<.,.,PRES=1> then reads them.
__global__ __launch_bounds__(256) void k_presence(const int8_t *__restrict__ gt, int S, const int64_t *__restrict__ win_lo,
                                                  const int64_t *__restrict__ win_hi, const int64_t *__restrict__ goff,
                                                  uint32_t *__restrict__ pres, int grp) {
    const int b = blockIdx.y, g = blockIdx.x;
    const int64_t lo = win_lo[b], hi = win_hi[b];
    const int W = (int)((hi - lo + 31) >> 5);
    const int w_begin = g * grp;
    if (w_begin >= W) return;
    const int w_end = (w_begin + grp < W) ? w_begin + grp : W;
    const int h0 = 4 * (blockIdx.z * 256 + threadIdx.x);
    const int64_t first = lo + 32ll * w_begin;
    const int nrows = (int)((hi - first) < 32ll * grp ? (hi - first) : 32ll * grp);
    const __amdgpu_buffer_rsrc_t rsrc = group_rsrc(gt, S, first, nrows);
    uint32_t *dst = pres + (size_t)(goff[b] + g) * grp * 4u;
    const RowOff ro = make_row_off(S);
    for (int w = w_begin; w < w_end; ++w) {
        uint32_t v[4], pa[4] = {0u, 0u, 0u, 0u};
        if (h0 < S) {
            uint32_t d[32];
            word_load(rsrc, h0, S, (w - w_begin) * 32, ro, d);
            word_called_presence(d, v, pa);
        }
        wave_or4(pa);
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if ((threadIdx.x & 63) == 0 && pa[a]) atomicOr(&dst[(size_t)(w - w_begin) * 4u + a], pa[a]);
    }
}

// PRES mode (32 virtual sites per word; a site with k alleles is k-1 virtual sites): words of XV each group will produce (from the presence nibbles of k_presence) and their exclusive prefix inside
// the window -> nw[n_win + group] = first word of the group, nw[window] = words of the window.  Block = one window.
__global__ __launch_bounds__(256) void k_word_scan(const int64_t *__restrict__ win_lo, const int64_t *__restrict__ win_hi,
                                                   const int64_t *__restrict__ goff, const uint32_t *__restrict__ pres,
                                                   int32_t *__restrict__ nw, int grp) {
    __shared__ int sh[256];
    __shared__ int carry;
    const int b = blockIdx.x, n_win = gridDim.x;
    const int64_t lo = win_lo[b], hi = win_hi[b];
    const int W = (int)((hi - lo + 31) >> 5);
    const int G = (W + grp - 1) / grp;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int g0 = 0; g0 < G; g0 += 256) {
        const int g = g0 + threadIdx.x;
        int words = 0;
        if (g < G) {
            const int nwords = (W - g * grp < grp) ? W - g * grp : grp;
            const uint32_t *src = pres + (size_t)(goff[b] + g) * grp * 4u;
            int cnt = 0;
            for (int w = 0; w < nwords; ++w) {
                const uint32_t p[4] = {src[4 * w], src[4 * w + 1], src[4 * w + 2], src[4 * w + 3]};
                cnt += __builtin_popcount(poly_mask(p)) + __builtin_popcount(tri_mask(p)) + __builtin_popcount(quad_mask(p));
            }
            words = (cnt + 31) >> 5;
        }
        sh[threadIdx.x] = words;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {                 // inclusive scan
            const int v = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += v;
            __syncthreads();
        }
        if (g < G) nw[n_win + goff[b] + g] = carry + sh[threadIdx.x] - words;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) nw[b] = carry;
}

template <int TPB, int DIP, int PRES>
__global__ __launch_bounds__(TPB) void k_pack2(const int8_t *__restrict__ gt, int S, const int64_t *__restrict__ win_lo,
                                               const int64_t *__restrict__ win_hi, const int64_t *__restrict__ goff,
                                               const int64_t *__restrict__ vgoff, uint32_t *__restrict__ Vp, int NPv,
                                               uint32_t *__restrict__ XV, int NP, int32_t *__restrict__ nw,
                                               int32_t *__restrict__ mismatch, const uint32_t *__restrict__ pres, int capg, int grp) {
    constexpr int NWAVE = TPB / 64;
    __shared__ uint32_t sh_pres[2][NWAVE][4];
    __shared__ uint4 sh_gp[PRES ? 1 : PG_GROUP_MAX];          // presence nibbles of the group's words (PRES: read from `pres`)
    __shared__ int sh_slot;
    const int b = blockIdx.y, g = blockIdx.x, n_win = gridDim.y;
    const int64_t lo = win_lo[b], hi = win_hi[b];
    const int W = (int)((hi - lo + 31) >> 5);
    const int w_begin = g * grp;
    if (w_begin >= W) return;
    const int w_end = (w_begin + grp < W) ? w_begin + grp : W;
    const int t = blockIdx.z * TPB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int h0 = 4 * t;
    // pad threads (h0 >= S = round_up(n_hap, 16)) neither load nor store: the pair kernels read pad units / haplotypes only as rows
    // of their last 8- / 16-row task (all below S) and discard those rows
    const bool has_data = h0 < S;
    const int u0 = DIP ? 2 * t : h0;         // first unit of the called plane owned by this thread
    const int64_t first = lo + 32ll * w_begin;
    const int nrows = (int)((hi - first) < 32ll * grp ? (hi - first) : 32ll * grp);
    const __amdgpu_buffer_rsrc_t rsrc = group_rsrc(gt, S, first, nrows);
    uint32_t vlist = (uint32_t)nrows;        // lane i = list entry i; "nrows" is one row past the descriptor: reads as zero
    int cnt = 0, nflush = 0, parity = 0;
    uint32_t bad = 0u;
    uint32_t *xv_base = XV + (size_t)goff[b] * capg * PG_XV_PLANES * (size_t)NP;
    const int capw = (int)(goff[b + 1] - goff[b]) * capg;          // words reserved for this window
    const uint32_t *pres_g = pres + (size_t)(goff[b] + g) * grp * 4u;       // PRES only
    const int64_t vg_base = vgoff[b] + (int64_t)(w_begin >> 2);
    // PRES: the group's words start at gbase (k_word_scan); otherwise every flush takes the window's next free word
    const int gbase = PRES ? nw[n_win + goff[b] + g] : 0;
    auto flush = [&]() {                     // the first 32 list entries become one dense word of XV
        int slot;
        if (PRES) {
            slot = gbase + nflush;
        } else {
            if (NWAVE == 1) {
                int s0 = 0;
                if (lane == 0) s0 = atomicAdd(&nw[b], 1);
                slot = __builtin_amdgcn_readfirstlane(s0);
            } else {
                if (threadIdx.x == 0) sh_slot = atomicAdd(&nw[b], 1);
                __syncthreads();
                slot = __builtin_amdgcn_readfirstlane(sh_slot);
                __syncthreads();
            }
        }
        // which allele each of the 32 entries tests (A) and which alleles it excludes (E), from the presence nibble of its site
        uint32_t A = 0u, E = 0u;
        {
            const int row = (int)(vlist & 0xffffu), ord = (int)(vlist >> 16);
            if (row < nrows) {
                const int wi = row >> 5, st = row & 31, bit = 4 * (st & 7) + (st >> 3);
                const uint4 pp = PRES ? *reinterpret_cast<const uint4 *>(pres_g + 4 * wi) : sh_gp[PRES ? 0 : wi];
                const uint32_t P = ((pp.x >> bit) & 1u) | (((pp.y >> bit) & 1u) << 1) | (((pp.z >> bit) & 1u) << 2) |
                                   (((pp.w >> bit) & 1u) << 3);
                const uint32_t A0 = P & (0u - P), P1 = P ^ A0, A1 = P1 & (0u - P1), P2 = P1 ^ A1, A2 = P2 & (0u - P2);
                A = ord == 0 ? A0 : (ord == 1 ? A1 : A2);
                E = (A - 1u) & P;
            }
        }
        // output bit 4j+q <-> entry q*8+j: lane 4j+q fetches that entry's A / E, the ballots then are the masks
        const int src = (lane & 3) * 8 + ((lane >> 2) & 7);
        A = (uint32_t)__shfl((int)A, src, 64);
        E = (uint32_t)__shfl((int)E, src, 64);
        const uint32_t SA[3] = {(uint32_t)__builtin_amdgcn_ballot_w64(A == 1u), (uint32_t)__builtin_amdgcn_ballot_w64(A == 2u),
                                (uint32_t)__builtin_amdgcn_ballot_w64(A == 4u)};
        const uint32_t SE[2] = {(uint32_t)__builtin_amdgcn_ballot_w64((E & 1u) != 0u),
                                (uint32_t)__builtin_amdgcn_ballot_w64((E & 2u) != 0u)};
        if (slot >= capw) {                  // more virtual sites than reserved: the host redoes the batch with the worst-case reservation
            if (threadIdx.x == 0) atomicOr(mismatch, 2);
        } else if (has_data) {
            uint32_t x[PG_XV_PLANES][4];
            poly_word(rsrc, h0, S, vlist, SA, SE, x);
            uint32_t *o = xv_base + (size_t)slot * PG_XV_PLANES * (size_t)NP + 2 * h0;
            store16(o, x[0][0], x[1][0], x[0][1], x[1][1]);
            store16(o + 4, x[0][2], x[1][2], x[0][3], x[1][3]);
        }
        ++nflush;
    };
    // the rows of word w+1 are requested before word w is processed (rows past the group read as zero, so the look-ahead
    // needs no bounds test); a wave therefore always has 32 row loads in flight
    uint32_t dn[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) dn[s] = 0u;
    const RowOff ro = make_row_off(S);
    if (has_data) word_load(rsrc, h0, S, 0, ro, dn);
    for (int wq = 0; 4 * wq + w_begin < w_end; ++wq) {
        uint32_t vhold[4][4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int w = w_begin + 4 * wq + k4;
            uint32_t v[4] = {0u, 0u, 0u, 0u}, pa[4] = {0u, 0u, 0u, 0u};
            const bool live = w < w_end;            // block-uniform
            uint32_t d[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) d[s] = dn[s];
            if (has_data) word_load(rsrc, h0, S, (w + 1 - w_begin) * 32, ro, dn);
            if (live && has_data) word_called_presence(d, v, pa);
#pragma unroll
            for (int k = 0; k < 4; ++k) vhold[k][k4] = v[k];
            if (DIP) bad |= (v[0] ^ v[1]) | (v[2] ^ v[3]);
            if (live) {
                // sites at which each allele occurs, across the whole block (uniform)
                uint32_t pr[4];
                if (PRES) {
                    const uint32_t *src = pres + ((size_t)(goff[b] + g) * grp + (size_t)(w - w_begin)) * 4u;
#pragma unroll
                    for (int a = 0; a < 4; ++a) pr[a] = __builtin_amdgcn_readfirstlane(src[a]);
                } else {
#pragma unroll
                    for (int a = 0; a < 4; ++a) pr[a] = pa[a];
                    wave_or4(pr);
                }
                if (!PRES && NWAVE > 1) {
                    if (lane == 0) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) sh_pres[parity][threadIdx.x >> 6][a] = pr[a];
                    }
                    __syncthreads();
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        uint32_t x = 0u;
#pragma unroll
                        for (int wv = 0; wv < NWAVE; ++wv) x |= sh_pres[parity][wv][a];
                        pr[a] = __builtin_amdgcn_readfirstlane(x);
                    }
                    parity ^= 1;
                }
                if (!PRES && threadIdx.x == 0) sh_gp[PRES ? 0 : w - w_begin] = make_uint4(pr[0], pr[1], pr[2], pr[3]);
                const int row_w = (w - w_begin) * 32;
                // pass 0: virtual site 0 of every polymorphic site; passes 1, 2 (rare): the second / third virtual site of the
                // sites with three / four alleles.  One rolled loop: a single copy of the list and flush code per word slot.
                const uint32_t m3 = tri_mask(pr);
#pragma unroll 1
                for (int pass = 0; pass < 3; ++pass) {
                    uint32_t m = pass == 0 ? poly_mask(pr) : (pass == 1 ? m3 : quad_mask(pr));
                    if (pass && !m3) break;
                    const uint32_t tag = (uint32_t)pass << 16;
                    while (m) {                      // scalar loop: the word's sites in m join the list (<= 32 of them)
                        const int bit = __builtin_ctz(m);
                        m &= m - 1u;
                        vlist = lane == cnt ? ((uint32_t)(row_w + (bit & 3) * 8 + (bit >> 2)) | tag) : vlist;
                        ++cnt;
                    }
                    if (cnt >= 32) {
                        flush();
                        const uint32_t up = (uint32_t)__shfl((int)vlist, (lane + 32) & 63, 64);
                        vlist = lane < 32 ? up : (uint32_t)nrows;
                        cnt -= 32;
                    }
                }
            }
        }
        if (has_data) {
            if (DIP) {
                uint32_t *o = Vp + ((size_t)(vg_base + wq) * NPv + u0) * 4u;
                store16(o, vhold[0][0], vhold[0][1], vhold[0][2], vhold[0][3]);
                store16(o + 4, vhold[2][0], vhold[2][1], vhold[2][2], vhold[2][3]);
            } else {
                uint32_t *o = Vp + ((size_t)(vg_base + wq) * NPv + u0) * 4u;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    store16(o + 4 * k, vhold[k][0], vhold[k][1], vhold[k][2], vhold[k][3]);
            }
        }
    }
    if (cnt) flush();
    if (DIP && bad) atomicOr(mismatch, 1);
}

// ------------------------------------------------------------------------------------------------------
// k_pack3: k_pack2 without its second visit of the polymorphic rows.  Phase A is k_pack2's; but the byte-transposed dwords
// r[q][k] (8 sites x one-hot nibble of haplotype h0+k) stay in registers until the block-wide polymorphic mask of the word is
// known, and the nibbles of the polymorphic sites are then copied, with two VALU ops per haplotype (v_bfe_u32 / v_lshl_or_b32,
// bit positions in SGPRs), into `cur[k]` = the next 8 list entries of haplotype h0+k.  Every 8 entries the dword is turned
// into 8 bits of the two output planes with nibble masks kept on the scalar unit (MA: the allele each entry tests, ME: the
// alleles it excludes; both derived from the site's presence nibble when the entry is appended):
//     x = nibble & MA != 0,   v = nibble & ~ME != 0,   "nibble != 0" = bit 3 of nibble + 7 (nibbles are 0 or one-hot).
// Every 32 entries the two planes are stored as one dense word of XV.  No list, no LDS, no second fetch: the kernel reads
// every row exactly once (k_pack2 fetched the ~11 % polymorphic rows of typical data a second time, PMC FETCH_SIZE).
// The entries of a word are appended q-major (sites q*8+j share r[q]): the bit order inside an XV word is arbitrary as
// long as it is the same for every haplotype and both planes.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void word_called_presence_keep(const uint32_t d[32], uint32_t v[4], uint32_t pa[4], uint32_t R[4][4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t *e = d + 8 * q;
        btrans4(e[0] | (e[1] << 4), e[2] | (e[3] << 4), e[4] | (e[5] << 4), e[6] | (e[7] << 4), R[q]);
        const uint32_t o = R[q][0] | R[q][1] | R[q][2] | R[q][3];          // 8 sites x presence nibble
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const uint32_t piece = (a >= q ? (o >> (a - q)) : (o << (q - a))) & (0x11111111u << q);
            pa[a] = q ? (pa[a] | piece) : piece;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // called = nibble != 0 = bit 3 of (nibble + 7); moved to bit q of the nibble
            const uint32_t t = R[q][k] + 0x77777777u;
            const uint32_t c = (q == 3 ? t : (t >> (3 - q))) & (0x11111111u << q);
            v[k] = q ? (v[k] | c) : c;
        }
    }
}

// BURST: the plane stores of a thread wait in LDS cells of its own (no barrier) and leave together -- the called plane every
// `fq` word quadruples, the virtual-site words as many at a time as the rest of the cells hold, with ONE reservation of consecutive slots -- because a store
// burst costs the HBM fewer read <-> write turn-arounds than the same bytes trickling out between the row loads
// (tools/ubench/pack_rw.hip: - 4.5 % on the kernel's bare traffic; the kernel's time does not depend on the waves per CU down to
// three blocks, so the 48 KB of LDS cost nothing).  Blocks of one or two waves only (LDS).
constexpr int PACK_CELLS = 24;                       // uint4 LDS cells per thread (48 KB per two-wave block: three blocks per CU)
template <int TPB, int DIP, int BURST>
__global__ __launch_bounds__(TPB) void k_pack3(const int8_t *__restrict__ gt, int S, const int64_t *__restrict__ win_lo,
                                               const int64_t *__restrict__ win_hi, const int64_t *__restrict__ goff,
                                               const int64_t *__restrict__ vgoff, uint32_t *__restrict__ Vp, int NPv,
                                               uint32_t *__restrict__ XV, int NP, int32_t *__restrict__ nw,
                                               int32_t *__restrict__ mismatch, int capg, int grp, int fq, int perm) {
    constexpr int NWAVE = TPB / 64;
    constexpr int VN = DIP ? 2 : 4;                      // uint4 of called plane per thread and word quadruple
    const int FQ = fq, xc = (PACK_CELLS - fq * VN) / 2;  // quadruples per burst of the called plane; virtual-site words per burst  // quadruples per burst of the called plane; virtual-site words per burst
    __shared__ uint32_t sh_pres[2][NWAVE][4];
    __shared__ int sh_slot;
    __shared__ uint4 stage[BURST ? PACK_CELLS * TPB : 1];
    uint4 *const stage_v = stage, *const stage_x = stage + FQ * VN * TPB;
    int nq = 0, nxs = 0, wq_first = 0;                   // staged quadruples / virtual-site words (block-uniform), first staged quadruple
    // perm (coprime with the number of windows, 1 = the windows in order): blocks that run at the same time -- consecutive
    // blockIdx.y -- work on windows `perm` apart, i.e. on rows and planes spread over the whole batch instead of one moving
    // stretch of it
    const int b = (int)(((unsigned long long)blockIdx.y * (unsigned)perm) % gridDim.y), g = blockIdx.x;
    const int64_t lo = win_lo[b], hi = win_hi[b];
    const int W = (int)((hi - lo + 31) >> 5);
    const int w_begin = g * grp;
    if (w_begin >= W) return;
    const int w_end = (w_begin + grp < W) ? w_begin + grp : W;
    const int t = threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int h0 = 4 * t;
    const bool has_data = h0 < S;
    const int u0 = DIP ? 2 * t : h0;
    const int64_t first = lo + 32ll * w_begin;
    const int nrows = (int)((hi - first) < 32ll * grp ? (hi - first) : 32ll * grp);
    const __amdgpu_buffer_rsrc_t rsrc = group_rsrc(gt, S, first, nrows);
    int cnt = 0, parity = 0;                 // cnt: entries of the pending output word (uniform, 0..31)
    uint32_t MA = 0u, ME = 0u;               // nibble masks of the pending 8-entry dword (uniform)
    uint32_t cur[4] = {0u, 0u, 0u, 0u}, xo[4] = {0u, 0u, 0u, 0u}, vo[4] = {0u, 0u, 0u, 0u};
    uint32_t bad = 0u;
    uint32_t *xv_base = XV + (size_t)goff[b] * capg * PG_XV_PLANES * (size_t)NP;
    const int capw = (int)(goff[b + 1] - goff[b]) * capg;          // words reserved for this window
    const int64_t vg_base = vgoff[b] + (int64_t)(w_begin >> 2);
    auto reserve = [&](int n) -> int {       // n consecutive words of the window's XV area (block-uniform result)
        int slot;
        if (NWAVE == 1) {
            int s0 = 0;
            if (lane == 0) s0 = atomicAdd(&nw[b], n);
            slot = __builtin_amdgcn_readfirstlane(s0);
        } else {
            if (threadIdx.x == 0) sh_slot = atomicAdd(&nw[b], n);
            __syncthreads();
            slot = __builtin_amdgcn_readfirstlane(sh_slot);
            __syncthreads();
        }
        return slot;
    };
    auto flush_x = [&]() {                   // BURST: the staged words leave together, into consecutive slots
        if (!nxs) return;
        const int slot0 = reserve(nxs);
        for (int k = 0; k < nxs; ++k) {
            if (slot0 + k >= capw) {         // more virtual sites than reserved: the host redoes the batch with the worst-case reservation
                if (threadIdx.x == 0) atomicOr(mismatch, 2);
            } else if (has_data) {
                uint4 *o = reinterpret_cast<uint4 *>(xv_base + (size_t)(slot0 + k) * PG_XV_PLANES * (size_t)NP + 2 * h0);
                o[0] = stage_x[(2 * k) * TPB + t];
                o[1] = stage_x[(2 * k + 1) * TPB + t];
            }
        }
        nxs = 0;
    };
    auto store_word = [&]() {                // the pending planes become one dense word of XV (the window's next free word)
        if (BURST) {
            if (has_data) {
                stage_x[(2 * nxs) * TPB + t] = make_uint4(xo[0], vo[0], xo[1], vo[1]);
                stage_x[(2 * nxs + 1) * TPB + t] = make_uint4(xo[2], vo[2], xo[3], vo[3]);
            }
            if (++nxs == xc) flush_x();
        } else {
            const int slot = reserve(1);
            if (slot >= capw) {              // more virtual sites than reserved: the host redoes the batch with the worst-case reservation
                if (threadIdx.x == 0) atomicOr(mismatch, 2);
            } else if (has_data) {
                uint32_t *o = xv_base + (size_t)slot * PG_XV_PLANES * (size_t)NP + 2 * h0;
                store16(o, xo[0], vo[0], xo[1], vo[1]);
                store16(o + 4, xo[2], vo[2], xo[3], vo[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) xo[k] = vo[k] = 0u;
    };
    auto flush_v = [&]() {                   // BURST: the staged quadruples of the called plane
        if (has_data)
            for (int q = 0; q < nq; ++q) {
                uint4 *o = reinterpret_cast<uint4 *>(Vp + ((size_t)(vg_base + wq_first + q) * NPv + u0) * 4u);
#pragma unroll
                for (int k = 0; k < VN; ++k) o[k] = stage_v[(q * VN + k) * TPB + t];
            }
        nq = 0;
    };
    auto finish_dword = [&](int qd) {        // 8 entries (nibbles of cur[k]) -> bits 4j+qd of the two planes
        const uint32_t nME = ~ME;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t cx = (((cur[k] & MA) + 0x77777777u) >> 3) & 0x11111111u;
            const uint32_t cv = (((cur[k] & nME) + 0x77777777u) >> 3) & 0x11111111u;
            xo[k] |= cx << qd;
            vo[k] |= cv << qd;
            cur[k] = 0u;
        }
        MA = 0u;
        ME = 0u;
    };
    auto append = [&](const uint32_t (&Rq)[4], int j, uint32_t A, uint32_t E) {       // j, A, E uniform
        const int sh = 4 * (cnt & 7);
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] |= __builtin_amdgcn_ubfe(Rq[k], 4 * j, 4) << sh;
        MA |= A << sh;
        ME |= E << sh;
        ++cnt;
        if ((cnt & 7) == 0) {
            finish_dword((cnt >> 3) - 1);
            if (cnt == 32) {
                store_word();
                cnt = 0;
            }
        }
    };
    uint32_t dn[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) dn[s] = 0u;
    const RowOff ro = make_row_off(S);
    if (has_data) word_load(rsrc, h0, S, 0, ro, dn);
    for (int wq = 0; 4 * wq + w_begin < w_end; ++wq) {
        uint32_t vhold[4][4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int w = w_begin + 4 * wq + k4;
            uint32_t v[4] = {0u, 0u, 0u, 0u}, pa[4] = {0u, 0u, 0u, 0u};
            uint32_t R[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) R[q][k] = 0u;
            const bool live = w < w_end;            // block-uniform
            uint32_t d[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) d[s] = dn[s];
            if (has_data) word_load(rsrc, h0, S, (w + 1 - w_begin) * 32, ro, dn);
            if (live && has_data) word_called_presence_keep(d, v, pa, R);
#pragma unroll
            for (int k = 0; k < 4; ++k) vhold[k][k4] = v[k];
            if (DIP) bad |= (v[0] ^ v[1]) | (v[2] ^ v[3]);
            if (live) {
                uint32_t pr[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) pr[a] = pa[a];
                wave_or4(pr);
                if (NWAVE > 1) {
                    if (lane == 0) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) sh_pres[parity][threadIdx.x >> 6][a] = pr[a];
                    }
                    __syncthreads();
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        uint32_t x = 0u;
#pragma unroll
                        for (int wv = 0; wv < NWAVE; ++wv) x |= sh_pres[parity][wv][a];
                        pr[a] = __builtin_amdgcn_readfirstlane(x);
                    }
                    parity ^= 1;
                }
                const uint32_t m0 = poly_mask(pr);
                if (m0) {
                    // lane l (mod 32) works out the presence nibble of the site at mask bit l and its lowest allele, so that the
                    // scalar loop below fetches them with one v_readlane per entry instead of a dozen scalar bit operations
                    const int lb = lane & 31;
                    const uint32_t Pl = __builtin_amdgcn_ubfe(pr[0], lb, 1) | (__builtin_amdgcn_ubfe(pr[1], lb, 1) << 1) |
                                        (__builtin_amdgcn_ubfe(pr[2], lb, 1) << 2) | (__builtin_amdgcn_ubfe(pr[3], lb, 1) << 3);
                    const uint32_t A0l = Pl & (0u - Pl);
                    auto nib = [&](int bit) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)Pl, bit); };
                    // virtual site 0 of every polymorphic site: tests the lowest allele present, excludes nothing
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t mq = m0 & (0x11111111u << q);
                        while (mq) {
                            const int bit = __builtin_ctz(mq);
                            mq &= mq - 1u;
                            append(R[q], bit >> 2, (uint32_t)__builtin_amdgcn_readlane((int)A0l, bit), 0u);
                        }
                    }
                    // rare: the second / third virtual site of the sites with three / four alleles
                    const uint32_t m3 = tri_mask(pr);
                    if (m3) {
#pragma unroll 1
                        for (int pass = 1; pass < 3; ++pass) {
                            uint32_t m = pass == 1 ? m3 : quad_mask(pr);
                            while (m) {
                                const int bit = __builtin_ctz(m);
                                m &= m - 1u;
                                const uint32_t P = nib(bit);
                                const uint32_t A0 = P & (0u - P), P1 = P ^ A0, A1 = P1 & (0u - P1), P2 = P1 ^ A1, A2 = P2 & (0u - P2);
                                const uint32_t A = pass == 1 ? A1 : A2;
                                const int q = bit & 3;                        // uniform
                                if (q == 0) append(R[0], bit >> 2, A, (A - 1u) & P);
                                else if (q == 1) append(R[1], bit >> 2, A, (A - 1u) & P);
                                else if (q == 2) append(R[2], bit >> 2, A, (A - 1u) & P);
                                else append(R[3], bit >> 2, A, (A - 1u) & P);
                            }
                        }
                    }
                }
            }
        }
        if (BURST) {
            if (nq == 0) wq_first = wq;
            if (has_data) {
                if (DIP) {
                    stage_v[(nq * VN) * TPB + t] = make_uint4(vhold[0][0], vhold[0][1], vhold[0][2], vhold[0][3]);
                    stage_v[(nq * VN + 1) * TPB + t] = make_uint4(vhold[2][0], vhold[2][1], vhold[2][2], vhold[2][3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) stage_v[(nq * VN + k) * TPB + t] = make_uint4(vhold[k][0], vhold[k][1], vhold[k][2], vhold[k][3]);
                }
            }
            if (++nq == FQ) flush_v();
        } else if (has_data) {
            uint32_t *o = Vp + ((size_t)(vg_base + wq) * NPv + u0) * 4u;
            if (DIP) {
                store16(o, vhold[0][0], vhold[0][1], vhold[0][2], vhold[0][3]);
                store16(o + 4, vhold[2][0], vhold[2][1], vhold[2][2], vhold[2][3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) store16(o + 4 * k, vhold[k][0], vhold[k][1], vhold[k][2], vhold[k][3]);
            }
        }
    }
    if (cnt & 7) finish_dword(cnt >> 3);
    if (cnt) store_word();
    if (BURST) {
        flush_v();
        flush_x();
    }
    if (DIP && bad) atomicOr(mismatch, 1);
}

template <int DIP>
static void launch_pack2(hipStream_t st, int threads, dim3 grid, const int8_t *gt, int S, const int64_t *win_lo,
                         const int64_t *win_hi, const int64_t *goff, const int64_t *vgoff, uint32_t *Vp, int NPv, uint32_t *XV,
                         int NP, int32_t *nw, int32_t *mismatch, uint32_t *pres, int capg, int grp) {
    // Up to 4096 slots: k_pack3 (one block of up to 16 waves per group; every row fetched once; PMC: 2.27 instead of 2.56 GB per C2 pass, 44.5 instead of 49.2 GB per
    // north-star pass).  Same-box A/B (profiles/r02/ab_pack_*.txt): C2 0.450-0.485 vs 0.457-0.463 ms, north-star shape 7.7-7.9 vs
    // 8.0-9.2 ms -- once the per-entry allele look-up had moved from the scalar unit to a lane-parallel table (before that
    // k_pack3 lost on two-wave blocks, 8.3-9.0 ms: every wave of a block runs the per-entry scalar loop).  More than 4096 slots:
    // k_pack2 behind the presence pre-pass (round 2 took that route from 1024 slots on: C4 read its rows twice, 0.27 of HBM).
    // PG_PACK2=1 forces k_pack2 (A/B runs and tests).
    const bool force2 = getenv("PG_PACK2") != nullptr;
    if (threads <= 1024 && !force2) {
#define PG_PACK3B(T, B) hipLaunchKernelGGL((k_pack3<T, DIP, B>), grid, dim3(T), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, capg, grp, fq, perm)
#define PG_PACK3(T) PG_PACK3B(T, 0)
        // quadruples of the called plane per burst (the rest of the 24 LDS cells per thread holds virtual-site words)
        const int fq = DIP ? 8 : 4;
        // PG_PACK_PERM=k (experiment): windows in the order 0, P, 2P, ... (mod n), P the number coprime with n next to n / k
        int perm = 1;
        if (const char *pe = getenv("PG_PACK_PERM")) {
            const int k = atoi(pe), n = (int)grid.y;
            if (k > 1 && n > 2 * k) {
                auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
                perm = n / k;
                while (gcd(perm, n) != 1) ++perm;
            }
        }
        const bool burst = getenv("PG_PACK_BURST") == nullptr || atoi(getenv("PG_PACK_BURST")) != 0;       // (0: A/B runs, tests)
        if (threads <= 64 && burst) PG_PACK3B(64, 1);
        else if (threads <= 128 && burst) PG_PACK3B(128, 1);
        else if (threads <= 64) PG_PACK3(64);
        else if (threads <= 128) PG_PACK3(128);
        else if (threads <= 256) PG_PACK3(256);
        else if (threads <= 512) PG_PACK3(512);            // up to 2048 slots (C4: 2000 haplotypes): eight waves meet in LDS per word
        else PG_PACK3(1024);                               // up to 4096 slots
#undef PG_PACK3
#undef PG_PACK3B
        return;
    }
    if (threads <= 64)
        hipLaunchKernelGGL((k_pack2<64, DIP, 0>), grid, dim3(64), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres, capg, grp);
    else if (threads <= 128)
        hipLaunchKernelGGL((k_pack2<128, DIP, 0>), grid, dim3(128), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres, capg, grp);
    else if (threads <= 256)
        hipLaunchKernelGGL((k_pack2<256, DIP, 0>), grid, dim3(256), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres, capg, grp);
    else {
        grid.z = (threads + 255) / 256;
        hipLaunchKernelGGL(k_presence, grid, dim3(256), 0, st, gt, S, win_lo, win_hi, goff, pres, grp);
        hipLaunchKernelGGL(k_word_scan, dim3(grid.y), dim3(256), 0, st, win_lo, win_hi, goff, pres, nw, grp);
        hipLaunchKernelGGL((k_pack2<256, DIP, 1>), grid, dim3(256), 0, st, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres, capg, grp);
    }
}

// more slots than one k_pack3 block takes (or k_pack2 forced beyond one of ITS blocks): the presence pre-pass runs
bool pg_pack_needs_presence(int NP) { return NP / 4 > 1024 || (getenv("PG_PACK2") != nullptr && NP / 4 > 256); }

// pres: scratch of total_groups * PG_GROUP * 4 words, only used (and zeroed here) in that mode
void pg_launch_pack2(hipStream_t st, const int8_t *gt, int S, const int64_t *win_lo, const int64_t *win_hi,
                     const int64_t *goff, const int64_t *vgoff, int n_win, int max_groups, int64_t total_groups, uint32_t *Vp,
                     int NPv, uint32_t *XV, int NP, int32_t *nw, int dip, int32_t *mismatch, uint32_t *pres, int capg, int grp) {
    // nw[0 .. n_win): the caller hands over zeroed per-window word counters (atomic allocation; k_pairD reads them even when
    // every window of the batch is empty); nw[n_win ..): one slot per group in the > 1024-slot mode (k_word_scan)
    if (n_win <= 0 || max_groups <= 0) return;
    const int threads = NP / 4;
    if (pg_pack_needs_presence(NP)) (void)hipMemsetAsync(pres, 0, (size_t)total_groups * grp * 16u, st);
    dim3 grid(max_groups, n_win);
    if (dip) launch_pack2<1>(st, threads, grid, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres, capg, grp);
    else launch_pack2<0>(st, threads, grid, gt, S, win_lo, win_hi, goff, vgoff, Vp, NPv, XV, NP, nw, mismatch, pres, capg, grp);
}

// ------------------------------------------------------------------------------------------------------
// Task decoding shared by k_pairC / k_pairD.  1-D XCD-aware grid: block b runs on XCD b % 8; all waves of a window go
// to one XCD so its planes are served by that XCD's L2 (except the last n_win % 8 windows, which are spread over all XCDs).
// ------------------------------------------------------------------------------------------------------
struct PairCtx {
    int win, row0, nsub, col0, lower, lane, ks;
};

// A block of 4 waves owns one task (8*nsub rows x 64 columns of one window); each wave takes a quarter of the word range
// and the four partial results meet in LDS (block_reduce), so the common case needs neither atomics nor a zeroed matrix.
// kso > 1 additionally cuts the word range across blocks (more waves in flight when there are few windows x tasks); those
// partial counts are combined with integer atomics (exact, order independent).
__device__ __forceinline__ bool pair_decode(const PgTask2 *__restrict__ tasks, int n_tasks, int kso, int n_win, PairCtx &c) {
    const int xcd = blockIdx.x & 7;
    const int v = blockIdx.x >> 3;
    const int per_win = n_tasks * kso;
    const int full = n_win >> 3;                       // rows of 8 windows: window 8*row + xcd runs on XCD xcd
    int rem;
    if (v < full * per_win) {
        c.win = (v / per_win) * 8 + xcd;
        rem = v % per_win;
    } else {
        // the last n_win % 8 windows (all of them when a job has fewer than 8, e.g. a whole-genome distMat): their blocks are
        // dealt to the 8 XCDs in equal contiguous runs, so that no XCD idles and neighbouring tasks still share an L2
        const int total = (n_win & 7) * per_win, q = (total + 7) >> 3;
        const int vt = v - full * per_win, lin = xcd * q + vt;
        if (vt >= q || lin >= total) return false;     // block-uniform
        c.win = full * 8 + lin / per_win;
        rem = lin % per_win;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.ks = (rem / n_tasks) * 4 + wave;                 // this wave's part of the 4*kso parts of the word range
    c.lane = threadIdx.x & 63;
    const PgTask2 tk = tasks[rem % n_tasks];
    c.row0 = __builtin_amdgcn_readfirstlane(tk.row0);
    c.nsub = __builtin_amdgcn_readfirstlane(tk.nsub);
    c.col0 = __builtin_amdgcn_readfirstlane(tk.col0);
    c.lower = __builtin_amdgcn_readfirstlane(tk.lower);
    return true;
}

// sum of the four waves' acc[] into wave 0 (returns true there)
template <int R>
__device__ __forceinline__ bool block_reduce(uint32_t (&acc)[R], uint32_t (*red)[64], int lane) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave) {
#pragma unroll
        for (int r = 0; r < R; ++r) red[(wave - 1) * R + r][lane] = acc[r];
    }
    __syncthreads();
    if (wave) return false;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] += red[r][lane] + red[R + r][lane] + red[2 * R + r][lane];
    return true;
}

// Circulant tasks: row i owns the unordered pairs {i, i+d}, d = 1 .. floor(n/2) (indices mod n; for even n the pairs
// at distance n/2 belong to the smaller index), plus the diagonal when asked for.  The columns of a task are
// col0, col0+1, ... (mod n), `nvalid` of them; a row block of 8 therefore needs 8 + floor(n/2) consecutive columns, i.e. one
// wave up to 112 units, instead of the rectangles of a triangular tiling that leave half of the diagonal blocks' lanes idle.
template <int R>
__device__ __forceinline__ void pair_store_circ(const uint32_t (&acc)[R], int row0, int col0, int lane, int nvalid, int n, int diag,
                                                int atomic, int32_t *__restrict__ M) {
    if (lane >= nvalid) return;
    int j = col0 + lane;
    if (j >= n) j -= n;
    const int h = n >> 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = row0 + r;
        if (i >= n) continue;
        int d = j - i;
        if (d < 0) d += n;
        const bool keep = d == 0 ? (diag != 0) : (d <= h && !(2 * d == n && i > j));
        if (!keep) continue;
        int32_t *dst = i <= j ? &M[(size_t)i * n + j] : &M[(size_t)j * n + i];
        if (atomic) { if (acc[r]) atomicAdd(dst, (int32_t)acc[r]); }
        else *dst = (int32_t)acc[r];
    }
}

// ------------------------------------------------------------------------------------------------------
// k_pairC: units x units "both called" counts.  Wave = 8*NSUB rows (SGPR operands) x 64 columns, 4 words per iteration.
// ------------------------------------------------------------------------------------------------------
// popcount with accumulate in ONE VALU op (v_bcnt_u32_b32 d = popcount(s0) + s1); the compiler otherwise reassociates a row's
// four popcounts into bcnt + v_add3 (2.5 ops per word instead of 2).
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// The inner loop is hand-scheduled assembly (pg_pairc_loop.inc, generated by gen_pairc_asm.py): 8 rows x 4 words of row
// operands live in SGPRs, ping-ponged between two sets so that the s_load_dwordx16 pair of word group g+1 completes under the
// 64 VALU ops (v_and + accumulating v_bcnt) of group g; the column's global_load_dwordx4 runs two groups ahead in three
// rotating VGPR sets.  The look-ahead reads two word groups past the wave's range: the called plane is allocated with padding.
#include "pg_pairc_loop.inc"

__device__ __forceinline__ void pairC_body(const uint32_t *__restrict__ Vp, int64_t vg0, int nwq, int NPv, const PairCtx &c,
                                           uint32_t (&acc)[8], int n_units) {
    int j = c.col0 + c.lane;
    if (j >= n_units) j -= n_units;                                  // circulant task: columns wrap around
    const uint32_t stride = (uint32_t)NPv * 16u;                       // bytes per word group
    // 32-bit byte offsets inside the asm loop: a call covers at most 2 GiB of the plane
    const int chunk = (int)(0x7fffffffu / stride) > 2 ? (int)(0x7fffffffu / stride) - 2 : 1;
    for (int q0 = 0; q0 < nwq; q0 += chunk) {
        const int nq = (nwq - q0 < chunk) ? nwq - q0 : chunk;
        const uint32_t *base = Vp + (size_t)(vg0 + q0) * NPv * 4u;
        const uint64_t b64 = (uint64_t)base;
        const uint64_t ubase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
        uint32_t soff = (uint32_t)c.row0 * 16u, voff = (uint32_t)j * 16u;
        uint32_t n6 = (uint32_t)__builtin_amdgcn_readfirstlane(nq / 6), rem = (uint32_t)__builtin_amdgcn_readfirstlane(nq % 6);
        asm volatile(PG_PAIRC_LOOP_ASM
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                       "+v"(acc[7]), "+s"(soff), "+v"(voff), "+s"(n6)
                     : "s"(ubase), "s"(stride), "s"(rem)
                     : PG_PAIRC_LOOP_CLOBBERS);
    }
}

__global__ __launch_bounds__(256) void k_pairC(const uint32_t *__restrict__ Vp, const int64_t *__restrict__ vgoff, int n_win,
                                               const PgTask2 *__restrict__ tasks, int n_tasks, int kso, int NPv, int n_units,
                                               int diag, int32_t *__restrict__ Cmat) {
    __shared__ uint32_t red[3 * 8][64];
    PairCtx c;
    if (!pair_decode(tasks, n_tasks, kso, n_win, c)) return;
    const int64_t vg_all = vgoff[c.win];
    const int nwq_all = (int)(vgoff[c.win + 1] - vg_all);
    const int parts = 4 * kso;
    const int q0 = (int)((long long)nwq_all * c.ks / parts), q1 = (int)((long long)nwq_all * (c.ks + 1) / parts);
    uint32_t acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0u;
    if (q1 > q0) pairC_body(Vp, vg_all + q0, q1 - q0, NPv, c, acc, n_units);
    if (block_reduce<8>(acc, red, c.lane)) {
        int32_t *Cw = Cmat + (size_t)c.win * n_units * n_units;
        pair_store_circ<8>(acc, c.row0, c.col0, c.lane, c.nsub, n_units, diag, kso > 1, Cw);
    }
}

// extra cut of the word range across blocks: wanted when windows x tasks x 4 waves cannot fill 256 CUs x 4 SIMDs x 8
static int pick_kso(int n_win, int n_tasks, int64_t steps_per_window, int min_steps) {
    int64_t waves = (int64_t)n_win * n_tasks * 4;
    int ks = 1;
    while (ks < 16 && waves * ks < 8192 && steps_per_window / (ks * 8) >= min_steps) ks *= 2;
    return ks;
}

void pg_launch_pairC(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, const PgTask2 *tasks, int n_tasks,
                     int NPv, int n_units, int diag, int64_t avg_wq, int32_t *Cmat) {
    if (n_win <= 0 || n_tasks <= 0) return;
    const int kso = pick_kso(n_win, n_tasks, avg_wq, 24);
    if (kso > 1) (void)hipMemsetAsync(Cmat, 0, (size_t)n_win * n_units * n_units * 4, st);
    const int64_t blocks = (int64_t)((n_win + 7) / 8) * n_tasks * kso * 8;
    hipLaunchKernelGGL(k_pairC, dim3((unsigned)blocks), dim3(256), 0, st, Vp, vgoff, n_win, tasks, n_tasks, kso, NPv, n_units,
                       diag, Cmat);
}

// ------------------------------------------------------------------------------------------------------
// k_pairD: haplotype x haplotype difference counts over the compacted polymorphic words of a window.  Planes per word:
// b0, b1 = the two bits of the allele index (A,C,G,T = 0..3), v = called.
//   differ & both called == ((b0_i ^ b0_j) | (b1_i ^ b1_j)) & v_i & v_j   -> v_xor, 2 x v_bitop3, accumulating v_bcnt
// ------------------------------------------------------------------------------------------------------
// The inner loop is hand-scheduled assembly (pg_paird_loop.inc, generated by gen_paird_asm.py): the same software pipeline as
// k_pairC's, 16 rows x {x, v} of one word in 32 SGPRs, ping-ponged between two sets.
#include "pg_paird_loop.inc"

__device__ __forceinline__ void pairD_body(const uint32_t *__restrict__ XVw, int w0, int w1, int NP, int N, const PairCtx &c,
                                           uint32_t (&acc)[16]) {
    int j = c.col0 + c.lane;
    if (j >= N) j -= N;                                    // circulant task: columns wrap around
    const uint32_t stride = (uint32_t)NP * 4u * PG_XV_PLANES;              // bytes per word
    // 32-bit byte offsets inside the asm loop: a call covers at most 2 GiB of the planes
    const int chunk = (int)(0x7fffffffu / stride) > 2 ? (int)(0x7fffffffu / stride) - 2 : 1;
    for (int q0 = w0; q0 < w1; q0 += chunk) {
        const int nq = (w1 - q0 < chunk) ? w1 - q0 : chunk;
        const uint32_t *base = XVw + (size_t)q0 * NP * PG_XV_PLANES;
        const uint64_t b64 = (uint64_t)base;
        const uint64_t ubase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
        uint32_t soff = (uint32_t)c.row0 * 8u, voff = (uint32_t)j * 8u;
        uint32_t n6 = (uint32_t)__builtin_amdgcn_readfirstlane(nq / 6), rem = (uint32_t)__builtin_amdgcn_readfirstlane(nq % 6);
        asm volatile(PG_PAIRD_LOOP_ASM
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                       "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]),
                       "+v"(acc[14]), "+v"(acc[15]), "+s"(soff), "+v"(voff), "+s"(n6)
                     : "s"(ubase), "s"(stride), "s"(rem)
                     : PG_PAIRD_LOOP_CLOBBERS);
    }
}

// Tasks are circulant (16 rows x up to 64 consecutive columns mod N, PgTask2.nsub = valid columns), see pair_store_circ.
__global__ __launch_bounds__(256) void k_pairD(const uint32_t *__restrict__ XV, const int32_t *__restrict__ nw,
                                               const int64_t *__restrict__ goff, int n_win, const PgTask2 *__restrict__ tasks,
                                               int n_tasks, int kso, int NP, int N, int32_t *__restrict__ Dmat, int capg) {
    __shared__ uint32_t red[3 * 16][64];
    PairCtx c;
    if (!pair_decode(tasks, n_tasks, kso, n_win, c)) return;
    const uint32_t *XVw = XV + (size_t)goff[c.win] * capg * PG_XV_PLANES * (size_t)NP;      // the window's words
    // (a window that overflowed its reservation is recomputed by the host; never read past the reservation)
    const int capw = (int)(goff[c.win + 1] - goff[c.win]) * capg;
    const int n_all = __builtin_amdgcn_readfirstlane(nw[c.win]);
    const int n_words = n_all < capw ? n_all : capw;
    const int parts = 4 * kso;
    const int a = (int)((long long)n_words * c.ks / parts), b = (int)((long long)n_words * (c.ks + 1) / parts);
    int32_t *Dw = Dmat + (size_t)c.win * N * N;
    uint32_t acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0u;
    if (b > a) pairD_body(XVw, a, b, NP, N, c, acc);
    if (block_reduce<16>(acc, red, c.lane)) pair_store_circ<16>(acc, c.row0, c.col0, c.lane, c.nsub, N, 0, kso > 1, Dw);
}

void pg_launch_pairD(hipStream_t st, const uint32_t *XV, const int32_t *nw, const int64_t *goff, int n_win,
                     const PgTask2 *tasks, int n_tasks, int NP, int N, int64_t avg_groups, int32_t *Dmat, int capg) {
    if (n_win <= 0 || n_tasks <= 0) return;
    const int kso = pick_kso(n_win, n_tasks, avg_groups, 1);
    if (kso > 1) (void)hipMemsetAsync(Dmat, 0, (size_t)n_win * N * N * 4, st);
    const int64_t blocks = (int64_t)((n_win + 7) / 8) * n_tasks * kso * 8;
    hipLaunchKernelGGL(k_pairD, dim3((unsigned)blocks), dim3(256), 0, st, XV, nw, goff, n_win, tasks, n_tasks, kso, NP, N, Dmat, capg);
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
