// Pairwise called counts on the matrix cores, every plane word fetched ONCE per window part: k_pairC_tile.
//
// Same arithmetic as k_pairC_fp4 / k_pairD_fp4 (pg_pair_mfma.hip): the counts are Gram matrices of 0/1 vectors
// (genomics.py:903-916, 1219-1221, 1042-1047), C = V V^T over the called plane, D = A B^T + B A^T over the
// virtual-site planes (a = x & v, b = ~x & v), multiplied as exact MX fp4 products (a set site is the e2m1 nibble 0b0001 = 0.5, an
// accumulator holds count / 4, exact in f32 below 2^24; the launcher cuts the word range into parts below 2^23 sites).
//
// What is different is who reads the planes.  The one-wave blocks of pg_pair_mfma.hip each read the row and column words of their
// 2 x 2 tiles; the ten tasks of a 200-unit window walk the words at different speeds (1 .. 4 products per step), so an XCD's L2
// (4 MB against ~50 windows in flight) serves little of the re-reading: 7.7 GB fetched per north-star launch for a 2.8 GB
// plane, the kernel at the fabric's limit with the matrix cores 45 % busy (profiles/r02/northstar_pmc_summary.json).  Here a
// BLOCK owns a window part: its waves stream the part's words into LDS with `global_load_lds_dwordx4` (the planes are already in
// the order the fragments want, so a stage is one linear copy: no registers, no ds_write), three stages deep, and every wave
// reads the fragments of its tiles from LDS.
//
//   tile      32 x 32 units, K = 64 sites per v_mfma_scale_f32_32x32x64_f8f6f4; lane (r = lane & 31, kb = lane >> 5) holds unit
//             32 t + r and the step's word kb: one ds_read_b128 = the four words of group 2 p + kb = four K steps.
//             Fragment dword m = (word >> m) & 0x11111111 (7 VALU per fragment).
//   strip     two tile rows (64 units) r0, r0+1 and the columns j >= r0; a SLOT is one column of a strip = two products per step
//             (the first slot of a strip, j == r0, has only the diagonal tile: `one` = 1).  The row fragments of a strip stay
//             in registers while its slots stream through: 7 VALU ops per two 32-cycle matrix instructions (the one-wave kernels:
//             per one).
//   program   the host deals the slots of the upper triangle, in strip order, to the W waves of `nblk` blocks in equal runs of at
//             most CS slots (accumulators: 32 registers per slot); a table in device memory tells every wave its run.
//   pipeline  iteration i: wait for the own copies of stage i (counted vmcnt: the copies of stage i+1 stay in flight), barrier,
//             queue the copies of stage i+2 into the ring slot that stage i-1 has just left, compute stage i from LDS.
//
// The lane halves of a step hold different words, both operands in the same (permuted) site order: all a dot product needs.
#include "pg_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int NSTG = 3;                // ring depth (stages)

// One LDS-DMA copy: 64 lanes x 16 bytes from each lane's `gsrc` to the wave-uniform LDS byte address `lds_dst` + 16 * lane.  Issued
// from inline assembly on purpose: hipcc orders every ds_read behind ALL outstanding LDS-DMA it knows of (s_waitcnt vmcnt(0) in
// front of the first fragment read), which would serialise the ring; these copies are invisible to its bookkeeping and are
// ordered by the counted waits + barriers of the stage loop instead.  (M0 = destination base, written in the same statement.)
__device__ __forceinline__ void glds16(const uint4 *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// at most n vector-memory operations of this wave still in flight (n is small and block-uniform)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define PG_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        PG_VM(0) PG_VM(1) PG_VM(2) PG_VM(3) PG_VM(4) PG_VM(5) PG_VM(6) PG_VM(7) PG_VM(8) PG_VM(9) PG_VM(10) PG_VM(11) PG_VM(12)
        PG_VM(13) PG_VM(14) PG_VM(15) PG_VM(16)
#undef PG_VM
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// XCD-aware block -> (window, rest): block b runs on XCD b % 8; all blocks of a window go to one XCD (the last n_win % 8 windows
// are dealt over all XCDs in contiguous runs).  Same dealing as pg_pair_mfma.hip.
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

// Fragments of one word (32 sites of one unit) as e2m1 nibbles, one site per nibble.  A nibble with only bit 0 / 1 / 2 set is 0.5 /
// 1.0 / 2.0 (bit 3 is the sign: -0), so the streamed operand (the COLUMNS of a slot) takes bits m = 0, 1, 2 of every nibble where
// they are -- three ANDs -- and only bit 3 needs a shift: 5 VALU ops instead of 7.  The cached operand (the ROWS of a strip)
// puts the same sites into the same (dword, nibble) places with the reciprocal values 2.0 / 1.0 / 0.5 / 2.0, so every product of
// two set sites is exactly 1.0 and an accumulator holds the count itself (exact in f32 below 2^24).
//     column dword m: sites 4n+m at value 0.5, 1.0, 2.0, 0.5 (m = 3 shifted down to bit 0)
//     row    dword m: the same sites at value 2.0, 1.0, 0.5, 2.0
// K1 / K2 / K4 = 0x11111111 / 0x22222222 / 0x44444444 in registers (an SGPR or literal source costs issue time); the ROW masks
// are zero in a lane whose word lies beyond the part (that lane then contributes nothing, whatever the column holds).
struct Masks { uint32_t k1, k2, k4; };

__device__ __forceinline__ v4i expand_col(uint32_t w, const Masks &K) {
    v4i f;
    f[0] = (int)(w & K.k1);
    f[1] = (int)(w & K.k2);
    f[2] = (int)(w & K.k4);
    f[3] = (int)((w >> 3) & K.k1);
    return f;
}

__device__ __forceinline__ v4i expand_row(uint32_t w, const Masks &K) {
    v4i f;
    f[0] = (int)((w << 2) & K.k4);
    f[1] = (int)(w & K.k2);
    f[2] = (int)((w >> 2) & K.k1);
    f[3] = (int)((w >> 1) & K.k4);
    return f;
}

// The slot bodies are hand-scheduled assembly: hipcc clusters the expansions in front of the matrix instructions (and hoists
// them out of the slot's block), which leaves a wave alternating between a VALU burst and an MFMA burst.  In the bodies the
// expansion of step k+1 sits between the matrix instructions of step k (5 VALU issue slots under each 32-cycle instruction), and
// the first expansion of a body runs under the previous body's last instruction.  Rules the strings obey (the compiler pads
// nothing inside an asm statement): a VALU-written fragment is read by a matrix instruction no sooner than two instructions
// later; an accumulator is only ever touched by matrix instructions between the zeroing and the epilogue (s_nop 11 in front of
// it: results of an 8-pass instruction).  Scratch registers: v[239:255] (clobbered; the kernels are built for 256 registers).
// `v_mfma_f32_32x32x64_f8f6f4` is the unscaled form (both scales 2^0), cbsz / blgp = 4: both operands fp4.
#define PG_F0 "v[240:243]"
#define PG_F1 "v[244:247]"
#define PG_F2 "v[248:251]"
#define PG_F3 "v[252:255]"
#define PG_EXP_A(d0, d1, w) "v_and_b32 " d0 ", " w ", %[k1]\n\tv_and_b32 " d1 ", " w ", %[k2]\n\t"
#define PG_EXP_B(d2, d3, w) "v_and_b32 " d2 ", " w ", %[k4]\n\tv_lshrrev_b32 v239, 3, " w "\n\tv_and_b32 " d3 ", v239, %[k1]\n\t"
#define PG_MFMA(acc, a, b) "v_mfma_f32_32x32x64_f8f6f4 " acc ", " a ", " b ", " acc " cbsz:4 blgp:4\n\t"
#define PG_SCRATCH "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

__device__ __forceinline__ uint32_t comp(const uint4 &v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

// accumulator tile (the counts) -> upper triangle of the window's matrix.  C/D layout of the 32 x 32 product: column = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
template <int SCALE = 1>
__device__ __forceinline__ void store_tile(const v16f &acc, int I, int J, int lane, int n, int diag, int atomic, int32_t *__restrict__ M) {
    const int col = 32 * J + (lane & 31);
    if (col >= n) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = 32 * I + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row >= n || row > col || (row == col && !diag)) continue;
        const int32_t v = (int32_t)(SCALE == 1 ? acc[reg] : acc[reg] * (float)SCALE);
        int32_t *dst = &M[(size_t)row * n + col];
        if (atomic) { if (v) atomicAdd(dst, v); }
        else *dst = v;
    }
}

struct Slot { int r0, j, one; };
__device__ __forceinline__ Slot slot_of(int32_t e) { return Slot{e & 0xff, (e >> 8) & 0xff, (e >> 16) & 1}; }

// Copies of one stage: `bytes` of plane words starting at gsrc -> lds_stage, 1 KiB per wave instruction, dealt round-robin to
// the W waves; every wave issues exactly nl instructions (the surplus ones repeat the last chunk), so that one counted vmcnt
// fits all of them.
template <int W>
__device__ __forceinline__ void stage_copy(const uint4 *__restrict__ gsrc, uint4 *lds_stage, int chunks, int nl, int wave, int lane) {
    for (int c = 0; c < nl; ++c) {
        int ch = wave + c * W;
        ch = ch < chunks ? ch : chunks - 1;
        glds16(gsrc + (size_t)ch * 64 + lane, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr(lds_stage) + (uint32_t)ch * 1024u)));
    }
}

// ---- C: called counts of unit pairs ------------------------------------------------------------------------------------------
// Vp[(vgoff[win] + q) * NPv + unit] = the four words of group q (128 sites) of one unit.  A stage = GP pairs of groups; the lane
// half kb works on group 2 p + kb of pair p; K step t = word t of the lane's own group.

// One slot of a pair of groups: four K steps, the column fragment of step t+1 is expanded under the products of step t.  A
// diagonal slot (`one`) has no second tile: its matrix instructions are skipped by scalar branches INSIDE the body, so that both
// kinds of slot are the same statement to the compiler (two statements in an if / else made it shuffle the accumulators
// between registers around them -- copies of a matrix result that nothing pads).
__device__ __forceinline__ void slotC(const uint4 &cw, const Masks &K, const v4i (&fr)[2][4], int one, v16f &a0, v16f &a1) {
#define PG_MFMA1(a, b) "s_cbranch_scc1 1f\n\t" PG_MFMA("%[a1]", a, b) "1:\n\t"
    asm volatile("s_cmp_lg_u32 %[one], 0\n\t"
                 PG_EXP_A("v240", "v241", "%[w0]") PG_EXP_B("v242", "v243", "%[w0]")
                 PG_EXP_A("v244", "v245", "%[w1]")
                 PG_MFMA("%[a0]", "%[r00]", PG_F0)
                 PG_EXP_B("v246", "v247", "%[w1]")
                 PG_MFMA1("%[r10]", PG_F0)
                 PG_EXP_A("v248", "v249", "%[w2]")
                 PG_MFMA("%[a0]", "%[r01]", PG_F1)
                 PG_EXP_B("v250", "v251", "%[w2]")
                 PG_MFMA1("%[r11]", PG_F1)
                 PG_EXP_A("v252", "v253", "%[w3]")
                 PG_MFMA("%[a0]", "%[r02]", PG_F2)
                 PG_EXP_B("v254", "v255", "%[w3]")
                 PG_MFMA1("%[r12]", PG_F2)
                 "s_nop 1\n\t"
                 PG_MFMA("%[a0]", "%[r03]", PG_F3)
                 PG_MFMA1("%[r13]", PG_F3)
                 : [a0] "+v"(a0), [a1] "+v"(a1)
                 : [w0] "v"(cw.x), [w1] "v"(cw.y), [w2] "v"(cw.z), [w3] "v"(cw.w), [k1] "v"(K.k1), [k2] "v"(K.k2), [k4] "v"(K.k4),
                   [r00] "v"(fr[0][0]), [r01] "v"(fr[0][1]), [r02] "v"(fr[0][2]), [r03] "v"(fr[0][3]),
                   [r10] "v"(fr[1][0]), [r11] "v"(fr[1][1]), [r12] "v"(fr[1][2]), [r13] "v"(fr[1][3]), [one] "s"(one)
                 : "scc", PG_SCRATCH);
}

template <int CS, int W, int GP, int NST>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_pairC_tile(const uint32_t *__restrict__ Vp, const int64_t *__restrict__ vgoff, int n_win, int T, int nblk, int kparts, int NPv,
                  int n_units, int diag, const int32_t *__restrict__ prog, int nl, int32_t *__restrict__ Cmat) {
    extern __shared__ uint4 lds[];
    int win, rem;
    if (!win_decode(nblk * kparts, n_win, win, rem)) return;
    const int bp = rem % nblk, kp = rem / nblk;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int32_t *my = prog + (size_t)(bp * W + wave) * (CS + 1);
    // the wave's program lives in scalar registers (the copies below clobber "memory": anything left in memory would be re-read)
    const int ns = __builtin_amdgcn_readfirstlane(my[0]);
    int sr0[CS], sj[CS], sone[CS];
#pragma unroll
    for (int s = 0; s < CS; ++s) {
        const Slot sl = slot_of(__builtin_amdgcn_readfirstlane(my[1 + s]));
        sr0[s] = sl.r0;
        sj[s] = sl.j;
        sone[s] = sl.one;
    }
    const int64_t vg = vgoff[win];
    const int nwq = (int)(vgoff[win + 1] - vg);
    const int q0 = (int)((long long)nwq * kp / kparts), q1 = (int)((long long)nwq * (kp + 1) / kparts);
    const int atomic = kparts > 1;
    int32_t *Cw = Cmat + (size_t)win * n_units * n_units;
    v16f acc[CS][2];
#pragma unroll
    for (int s = 0; s < CS; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][i][e] = 0.0f;
    const int stage_groups = 2 * GP;
    const int stage_u4 = stage_groups * NPv;                  // uint4 elements per stage
    const int chunks = stage_u4 / 64;                         // NPv is a multiple of 32
    const int nstage = (q1 - q0 + stage_groups - 1) / stage_groups;
    const uint4 *base = reinterpret_cast<const uint4 *>(Vp) + ((size_t)vg + q0) * NPv;
    Masks KC;                                                 // column masks: never zeroed (the row masks carry `live`)
    KC.k1 = 0x11111111u;
    KC.k2 = 0x22222222u;
    KC.k4 = 0x44444444u;
    asm volatile("" : "+v"(KC.k1), "+v"(KC.k2), "+v"(KC.k4));  // stay in vector registers
    if (nstage > 0) {
        // (copies past the last stage repeat it into a ring slot nobody reads: the counted waits stay uniform; a stage may reach
        // up to stage_groups - 1 groups past q1: those words exist (next part / window / padding) and their lanes are masked)
        for (int st = 0; st < NST - 1; ++st) {
            const int src = st < nstage ? st : nstage - 1;
            stage_copy<W>(base + (size_t)src * stage_u4, lds + (size_t)st * stage_u4, chunks, nl, wave, lane);
        }
        // (tools/audit_pair_tile_asm.py: no compiler code may touch an accumulator from here on; the operands pin their zeroing above)
#pragma unroll
        for (int s = 0; s < CS; ++s) asm volatile("; PG_AUDIT_BEGIN" : "+v"(acc[s][0]), "+v"(acc[s][1])::"memory");
        for (int st = 0; st < nstage; ++st) {
            wait_vm(nl * (NST - 2));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            {
                const int nx = st + NST - 1, src = nx < nstage ? nx : nstage - 1;
                stage_copy<W>(base + (size_t)src * stage_u4, lds + (size_t)(nx % NST) * stage_u4, chunks, nl, wave, lane);
            }
            const uint4 *sb = lds + (size_t)(st % NST) * stage_u4;
#pragma unroll 1
            for (int p = 0; p < GP; ++p) {
                const bool live = q0 + st * stage_groups + 2 * p + kb < q1;
                Masks KR;
                KR.k1 = live ? KC.k1 : 0u;
                KR.k2 = live ? KC.k2 : 0u;
                KR.k4 = live ? KC.k4 : 0u;
                const uint4 *pb = sb + (size_t)(2 * p + kb) * NPv + r;
                int cur = -1;
                v4i fr[2][4];
                uint4 craw = pb[32 * sj[0]];
#pragma unroll
                for (int s = 0; s < CS; ++s) {
                    if (s < ns) {
                        if (sr0[s] != cur) {
                            cur = sr0[s];
                            const int t1 = cur + 1 < T ? cur + 1 : cur;
                            const uint4 a = pb[32 * cur], b = pb[32 * t1];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                fr[0][t] = expand_row(comp(a, t), KR);
                                fr[1][t] = expand_row(comp(b, t), KR);
                            }
                        }
                        const uint4 cw = craw;
                        if (s + 1 < CS) craw = pb[32 * sj[s + 1]];                 // the next slot's words (a padding slot repeats a real one)
                        slotC(cw, KC, fr, sone[s], acc[s][0], acc[s][1]);
                    }
                }
            }
        }
        wait_vm(0);                                           // the surplus copies of the last iterations land before the block ends
        asm volatile("s_nop 11 ; PG_AUDIT_END" ::: "memory");  // the last matrix instruction's result is complete
    }
    const bool zero_fill = nstage <= 0 && !atomic;            // an empty window: the counts are zero and nobody else writes them
    if (nstage > 0 || zero_fill) {
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            if (s < ns) {
                store_tile(acc[s][0], sr0[s], sj[s], lane, n_units, diag, atomic, Cw);
                if (!sone[s]) store_tile(acc[s][1], sr0[s] + 1, sj[s], lane, n_units, diag, atomic, Cw);
            }
        }
    }
}

// ---- host: slot programs ---------------------------------------------------------------------------------------------------
// Slots of the upper triangle of T x T tiles in strip order (strip = tile rows r0, r0+1; slot = column j >= r0; the slot j == r0
// holds the diagonal tile only, as does every slot of a last strip of one row), dealt to nblk * W waves of at most CS slots each,
// balanced by the number of matrix instructions.  Entry = r0 | j << 8 | one << 16; a wave's record = [count, CS entries].
struct Program {
    std::vector<int32_t> tab;
    int nblk = 0;
};

Program make_program(int T, int CS, int W) {
    std::vector<int32_t> slots;
    for (int r0 = 0; r0 < T; r0 += 2)
        for (int j = r0; j < T; ++j) slots.push_back(r0 | (j << 8) | ((j == r0 || r0 + 1 >= T) ? 1 << 16 : 0));
    const int n = (int)slots.size();
    auto weight = [&](int k) { return (slots[(size_t)k] >> 16) & 1 ? 1 : 2; };          // matrix instructions per step
    Program p;
    p.nblk = (n + CS * W - 1) / (CS * W);
    const int waves = p.nblk * W;
    // contiguous runs of at most CS slots with the smallest possible heaviest run (dynamic programme over the cut points): the
    // waves of a block meet at a barrier every stage, so the heaviest wave sets the block's pace
    const int INF = 1 << 28;
    std::vector<std::vector<int>> best((size_t)waves + 1, std::vector<int>((size_t)n + 1, INF)), from(best);
    best[0][0] = 0;
    for (int w = 1; w <= waves; ++w)
        for (int e = 0; e <= n; ++e)
            for (int b = std::max(0, e - CS); b <= e; ++b) {
                if (best[(size_t)w - 1][(size_t)b] >= INF) continue;
                int wt = 0;
                for (int k = b; k < e; ++k) wt += weight(k);
                const int v = std::max(best[(size_t)w - 1][(size_t)b], wt);
                if (v < best[(size_t)w][(size_t)e]) { best[(size_t)w][(size_t)e] = v; from[(size_t)w][(size_t)e] = b; }
            }
    std::vector<std::vector<int>> run((size_t)waves);
    for (int w = waves, e = n; w >= 1; --w) {
        const int b = from[(size_t)w][(size_t)e];
        for (int k = b; k < e; ++k) run[(size_t)w - 1].push_back(k);
        e = b;
    }
    // then single slots move from the heaviest wave to a lighter one while that lowers the heavier of the two (a slot that leaves
    // its strip costs the receiving wave one more row expansion per pair of groups: only diagonal slots, weight 1, are moved)
    auto load = [&](const std::vector<int> &r) { int t = 0; for (int k : r) t += weight(k); return t; };
    for (int iter = 0; iter < 4 * waves; ++iter) {
        int hi = 0;
        for (int w = 1; w < waves; ++w) if (load(run[(size_t)w]) > load(run[(size_t)hi])) hi = w;
        bool moved = false;
        for (int w = 0; w < waves && !moved; ++w) {
            if (w == hi || (int)run[(size_t)w].size() >= CS || load(run[(size_t)w]) + 1 >= load(run[(size_t)hi])) continue;
            for (size_t q = 0; q < run[(size_t)hi].size(); ++q) {
                const int k = run[(size_t)hi][q];
                if (weight(k) != 1) continue;
                run[(size_t)hi].erase(run[(size_t)hi].begin() + (long)q);
                run[(size_t)w].push_back(k);
                moved = true;
                break;
            }
        }
        // or a two-tile slot of the heaviest wave changes places with a diagonal slot of a lighter one
        for (int w = 0; w < waves && !moved; ++w) {
            if (w == hi || load(run[(size_t)w]) + 1 >= load(run[(size_t)hi])) continue;
            for (size_t q = 0; q < run[(size_t)hi].size() && !moved; ++q)
                for (size_t u = 0; u < run[(size_t)w].size() && !moved; ++u)
                    if (weight(run[(size_t)hi][q]) == 2 && weight(run[(size_t)w][u]) == 1) {
                        std::swap(run[(size_t)hi][q], run[(size_t)w][u]);
                        moved = true;
                    }
        }
        if (!moved) break;
    }
    for (std::vector<int> &r : run) std::sort(r.begin(), r.end());          // strip order inside a wave: one row expansion per strip
    p.tab.assign((size_t)waves * (CS + 1), 0);
    for (int w = 0; w < waves; ++w) {
        int32_t *rec = &p.tab[(size_t)w * (CS + 1)];
        const std::vector<int> &r = run[(size_t)w];
        rec[0] = (int)r.size();
        for (int s = 0; s < CS; ++s) rec[1 + s] = slots[(size_t)(r.empty() ? 0 : r[std::min<size_t>((size_t)s, r.size() - 1)])];
    }
    return p;
}

// device copy of the program of (T, CS, W); a handful of shapes per process, kept for its lifetime
struct ProgCache {
    int T = -1, CS = 0, W = 0, nblk = 0, device = -1;
    int32_t *d = nullptr;
};
ProgCache g_prog[8];
std::mutex g_prog_lock;                                      // contexts of several devices may be driven from several host threads

int get_program(int T, int CS, int W, const int32_t **d_out, int *nblk_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> guard(g_prog_lock);
    ProgCache *slot = nullptr, *mine = nullptr;
    for (ProgCache &c : g_prog) {
        if (c.d && c.T == T && c.CS == CS && c.W == W && c.device == dev) {
            *d_out = c.d;
            *nblk_out = c.nblk;
            return 0;
        }
        if (!c.d && !slot) slot = &c;
        if (c.d && c.device == dev && !mine) mine = &c;
    }
    if (!slot) {
        // the cache is full: recycle an entry of THIS device (its table is freed on the device that owns it; kernels that were
        // launched with it are ordered before the free by the runtime).  No entry of this device to give up: no program.
        if (!mine) return -1;
        (void)hipFree(mine->d);
        mine->d = nullptr;
        slot = mine;
    }
    const Program p = make_program(T, CS, W);
    const std::vector<int32_t> &tab = p.tab;
    int32_t *d = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d), tab.size() * 4) != hipSuccess) return -1;
    if (hipMemcpy(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return -1;
    }
    slot->d = d;
    slot->nblk = p.nblk;
    slot->T = T;
    slot->CS = CS;
    slot->W = W;
    slot->device = dev;
    *d_out = slot->d;
    *nblk_out = slot->nblk;
    return 0;
}

// extra cut of the word range across blocks: wanted when windows x blocks cannot give every SIMD a few waves
int pick_parts(int n_win, int waves_per_win, int64_t steps_per_window, int min_steps) {
    const int64_t waves = (int64_t)n_win * waves_per_win;
    int kp = 1;
    while (kp < 64 && waves * kp < 4096 && steps_per_window / (kp * 2) >= min_steps) kp *= 2;
    return kp;
}
// an f32 accumulator holds a count exactly while it is < 2^24; no part of any window may see more sites than 2^23
int exact_parts(int64_t max_sites_per_window) { return (int)((max_sites_per_window + (1 << 23) - 1) >> 23); }

constexpr int CS_C = 4, W_C = 4, GP_C = 2;      // C: 4 waves x 4 slots, stage = 2 pairs of groups (512 sites)

}  // namespace

// The LDS-staged kernel takes planes of up to this many units per word (a stage must fit the ring).  PG_PAIR_TILE (A/B runs, tests)
// chooses who counts the called pairs: a 'b' lets k_pairC_big (pg_pair_big.hip) take planes of up to 224 units, a 'c' lets
// k_pairC_tile take what fits its ring, "none" leaves everything to the one-wave kernel of pg_pair_mfma.hip.  Default "bc".
// (The LDS-staged form of the DIFFERENCE counts lost to the one-wave kernel on every shape -- 1.3 vs 0.85 ms on the north-star
// shape -- and was removed in round 4; HISTORY.md.)
bool pg_pair_tile_fits(int NPv) {
    const char *sel = getenv("PG_PAIR_TILE");
    if (!sel) sel = "bc";
    if (!strchr(sel, 'c')) return false;
    const int64_t stage = (int64_t)2 * GP_C * NPv * 16;
    return NPv % 32 == 0 && stage * NSTG <= 64 * 1024;
}

int pg_launch_pairC_tile(hipStream_t st, const uint32_t *Vp, const int64_t *vgoff, int n_win, int NPv, int n_units, int diag,
                         int64_t avg_wq, int64_t max_sites, int32_t *Cmat) {
    if (n_win <= 0 || n_units <= 0) return 0;
    const int T = (n_units + 31) / 32;
    const int32_t *prog;
    int nblk;
    if (get_program(T, CS_C, W_C, &prog, &nblk) != 0) return -1;
    const int kparts = std::max(pick_parts(n_win, nblk * W_C, avg_wq / 2, 16), exact_parts(max_sites));
    if (kparts > 1) (void)hipMemsetAsync(Cmat, 0, (size_t)n_win * n_units * n_units * 4, st);
    // ring shape: 2 pairs of groups per stage, 2 stages (measured on the north-star shape against 2 x 3 and 4 x 2: 1.24 / 1.31 / 1.28 ms)
    constexpr int gp = 2, nst = 2;
    const int stage_u4 = 2 * gp * NPv, chunks = stage_u4 / 64, nl = (chunks + W_C - 1) / W_C;
    const size_t lds_bytes = (size_t)nst * stage_u4 * 16;
    const int64_t blocks = (int64_t)((n_win + 7) / 8) * nblk * kparts * 8;
    hipLaunchKernelGGL((k_pairC_tile<CS_C, W_C, gp, nst>), dim3((unsigned)blocks), dim3(64 * W_C), lds_bytes, st, Vp, vgoff, n_win, T, nblk,
                       kparts, NPv, n_units, diag, prog, nl, Cmat);
    return 0;
}
