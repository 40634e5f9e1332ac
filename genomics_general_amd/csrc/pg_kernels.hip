// Hand-written gfx950 (CDNA4, wave64) kernels of the per-window statistics path (the pack + pairwise kernels are in pg_pair2.hip,
// pg_pair_mfma.hip and pg_pair_tile.hip).
//
//   k_synth          counter-based synthetic genotype generator (spec: genomics_general_amd/synth.py)
//   k_popdist_fin, k_popstats   D,C -> per population-pair float64 sums of D/C + valid-pair counts -> pi / dxy / Fst
//                    (genomics.py:956-995, 88-90)
//   k_indpair_fin    D,C -> per individual-pair sums / counts or finished nanmeans (genomics.py:934-954)
//   k_abba_q         screening pass + per-population base counts -> ABBA/BABA/f4... window sums (genomics.py:1647-1695, 1585-1643)
//   k_popfreq_q      screening pass + per-population base counts -> l, S, sum of allele-pair products (genomics.py:1002-1028)
//   k_site_counts    raw per-site per-population base counts (genomics.py:1049-1052);  k_hap_called (genomics.py:1038-1040)
//
// Integer work is exact; float64 work is compiled with -ffp-contract=off so the per-site products are the
// same IEEE operations, in the same order, as the NumPy expressions of the reference.
#include "pg_internal.h"
#include <algorithm>
#ifdef PG_DIV_PROBE
#include <cstdio>
#endif

#define WAVE 64

// ------------------------------------------------------------------------------------------------------
// K_synth
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_synth(int8_t *__restrict__ gt, int S, int n_hap, int64_t site0,
                                               int64_t n_sites, const int32_t *__restrict__ slot_gen_hap,
                                               PgSynthParams p) {
    const int groups = S >> 2;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t site = idx / groups;
    int g = (int)(idx - site * groups);
    if (site >= n_sites) return;
    int64_t gi = p.first_site_index + site;
    uint64_t scaf = (uint64_t)(gi / p.scaf_len);
    uint64_t pos = (uint64_t)(gi % p.scaf_len) + 1ull;
    uint64_t ks = mix64(mix64(p.seed ^ (scaf * 0xD6E8FEB86659FD93ull)) ^ pos);
    int ref = (int)(ks & 3ull);
    bool variable = (int)((ks >> 2) & 0xFFFFull) < p.var_thr;
    int altoff = (int)((ks >> 18) & 0xFFull) % 3;
    int alt = (ref + 1 + altoff) & 3;
    bool has3 = (int)((ks >> 26) & 0xFFFFull) < 655;
    int third = (ref + 1 + (altoff + 1) % 3) & 3;
    int64_t p16 = (int64_t)((ks >> 42) & 0xFFFFull);
    uint32_t word = 0;
    for (int k = 0; k < 4; ++k) {
        int h = 4 * g + k;
        if (h >= n_hap) break;
        int gh = slot_gen_hap[h];
        int dip = gh >> 1;
        int pop = (int)(((int64_t)dip * p.n_pops) / p.n_dip);
        uint64_t kp = mix64(ks ^ (0xA24BAED4963EE407ull * (uint64_t)(pop + 1)));
        int64_t z = (int64_t)(kp & 0xFFFFull) + (int64_t)((kp >> 16) & 0xFFFFull) +
                    (int64_t)((kp >> 32) & 0xFFFFull) + (int64_t)((kp >> 48) & 0xFFFFull) - 131070;
        int64_t v = p16 + ((z * 17027) >> 16);
        v = v < 0 ? 0 : (v > 65535 ? 65535 : v);
        if (pop == p.n_pops - 1 && p.n_pops > 1) {
            if ((int)(mix64(kp) & 0xFFFFull) < 52429) v = 0;
        }
        uint64_t kh = mix64(ks ^ (0x9FB21C651E98DF25ull * (uint64_t)(gh + 1)));
        bool derived = (int64_t)(kh & 0xFFFFull) < v;
        bool use3 = has3 && ((int)((kh >> 16) & 0xFFull) < 26);
        uint64_t kd = mix64(ks ^ (0xC2B2AE3D27D4EB4Full * (uint64_t)(dip + 1)));
        bool missing = (int)((kd >> 8) & 0xFFFFull) < p.miss_thr;
        int allele = ref;
        if (variable && derived) allele = use3 ? third : alt;
        uint32_t code = missing ? 0u : (1u << allele);
        word |= code << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(gt + (site0 + site) * (int64_t)S + 4 * g) = word;
}

void pg_launch_synth(hipStream_t st, int8_t *gt, int S, int n_hap, int64_t site0, int64_t n_sites,
                     const int32_t *slot_gen_hap, PgSynthParams p) {
    // a launch holds at most 2^30 threads: a grid of 2^32 threads or more is silently truncated by the runtime (1e8 sites x 400
    // haplotypes = 1e10 threads filled 14 % of the rows in one launch)
    const int64_t groups = S >> 2;
    if (n_sites <= 0 || groups <= 0) return;
    const int64_t sites_per_launch = std::max<int64_t>(1, (1ll << 30) / groups);
    for (int64_t a = 0; a < n_sites; a += sites_per_launch) {
        const int64_t n = std::min(sites_per_launch, n_sites - a);
        const int64_t blocks = (n * groups + 255) / 256;
        PgSynthParams q = p;
        q.first_site_index = p.first_site_index + a;
        hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(256), 0, st, gt, S, n_hap, site0 + a, n, slot_gen_hap, q);
    }
}

// ------------------------------------------------------------------------------------------------------
// Deterministic block reductions (fixed thread partition + fixed tree).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_f64(double v, double *sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) sh[t] += sh[t + s];
        __syncthreads();
    }
    double r = sh[0];
    __syncthreads();
    return r;
}

// N sums at once: a fixed xor-butterfly inside each wave (shuffles, no barrier), then the waves' totals through LDS in wave
// order.  One barrier instead of ten per value; the result (valid in thread 0) depends only on the thread partition.
template <int N>
__device__ __forceinline__ void block_sum_multi(double (&v)[N], double *sh /* [waves][N] */) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m, 64);
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) sh[wave * N + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double t = sh[k];
            for (int w = 1; w < nw; ++w) t += sh[w * N + k];
            v[k] = t;
        }
    }
}

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) sh[t] += sh[t + s];
        __syncthreads();
    }
    unsigned long long r = sh[0];
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------------------------------------------
// K_popdist: one block per (population pair, window).
// ------------------------------------------------------------------------------------------------------
// D / C for counts (0 <= D <= C < 2^31): the quotient needs neither the range scaling nor the special-case fix-up of the
// compiler's IEEE division sequence, and the reciprocal of C serves every D that shares it.  rcp_counts: v_rcp_f64 + two
// Newton steps (error below one ulp); quot_counts: product, exact residual, one correction -- the step the IEEE sequence ends in.
__device__ __forceinline__ double rcp_counts(double c) {
    double r = __builtin_amdgcn_rcp(c);
    r = __builtin_fma(__builtin_fma(-c, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-c, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double quot_counts(double d, double c, double r) {
    const double q = d * r;
    return __builtin_fma(__builtin_fma(-c, q, d), r, q);
}

__device__ unsigned long long g_div_probe[2];           // PG_DIV_PROBE builds only: quotients compared / differing from d / c

// MODE 0: any layout, one haplotype pair at a time.  MODE 1 / 2: every sample diploid (haplotypes 2u, 2u + 1 of individual u, populations
// made of whole individuals): the 2 x 2 haplotype pairs of an individual pair together, in one order for both -- 1: called counts per
// individual pair (one reciprocal for the four), 2: per haplotype pair (PG_NO_DIP, half-missing genotypes) -- so that the two give
// the same float64 sums.
template <int MODE>
__global__ __launch_bounds__(256) void k_popdist_fin(const int32_t *__restrict__ Cmat, const int32_t *__restrict__ Dmat,
                                                     int N, int cN, int cshift, const int32_t *__restrict__ pop_start, int n_pops,
                                                     int min_pair_sites, double *__restrict__ sum_out,
                                                     int64_t *__restrict__ cnt_out, const int64_t *__restrict__ win_lo,
                                                     const int64_t *__restrict__ win_hi, long long skip_upto) {
    __shared__ double shd[8];
    if (win_lo && win_hi[blockIdx.y] - win_lo[blockIdx.y] <= skip_upto) return;      // k_popdist_np's window
    // decode pair index -> (x<=y)
    int pidx = blockIdx.x, x = 0;
    int rem = pidx;
    while (rem >= n_pops - x) { rem -= n_pops - x; ++x; }
    const int y = x + rem;
    const int win = blockIdx.y;
    const int32_t *Cw = Cmat + (size_t)win * cN * cN;      // unit-level called counts (units = haplotypes or diploid individuals)
    const int32_t *Dw = Dmat + (size_t)win * N * N;
    const int xs = pop_start[x], xe = pop_start[x + 1], ys = pop_start[y], ye = pop_start[y + 1];
    const int nx = xe - xs, ny = ye - ys;
    const int thr = min_pair_sites > 1 ? min_pair_sites : 1;
    double sum = 0.0;
    unsigned long long cnt = 0;
    if (MODE) {
        // the haplotype pairs of an individual pair (a, b) share their called count (MODE 1).  Thread t takes the
        // individual pairs t, t + 256, ... of the (nx / 2) x (ny / 2) rectangle: one reciprocal, two 8-byte loads of D, four
        // quotients added in the fixed order (2a,2b) (2a,2b+1) (2a+1,2b) (2a+1,2b+1); within an individual only (2a, 2a+1)
        const int ux = nx >> 1, uy = ny >> 1, u0 = xs >> 1, v0 = ys >> 1;
        const int total = ux * uy;
        const int qi = 256 / (uy > 0 ? uy : 1), qj = 256 - qi * (uy > 0 ? uy : 1);
        int a = u0 + (uy > 0 ? (int)threadIdx.x / uy : 0), b = v0 + (uy > 0 ? (int)threadIdx.x % uy : 0);
        for (int idx = threadIdx.x; idx < total; idx += 512) {
            int aa[2], bb[2];
            int2 c0[2], c1[2], d0[2], d1[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (b >= v0 + uy) { b -= uy; ++a; }
                const bool live = idx + 256 * u < total && !(x == y && a > b);
                aa[u] = a;
                bb[u] = b;
                if (MODE == 1) {
                    c0[u].x = live ? Cw[(size_t)a * cN + b] : 0;
                    c0[u].y = c1[u].x = c1[u].y = c0[u].x;
                } else {
                    c0[u] = live ? *reinterpret_cast<const int2 *>(Cw + (size_t)(2 * a) * cN + 2 * b) : make_int2(0, 0);
                    c1[u] = live ? *reinterpret_cast<const int2 *>(Cw + (size_t)(2 * a + 1) * cN + 2 * b) : make_int2(0, 0);
                }
                d0[u] = live ? *reinterpret_cast<const int2 *>(Dw + (size_t)(2 * a) * N + 2 * b) : make_int2(0, 0);
                d1[u] = live ? *reinterpret_cast<const int2 *>(Dw + (size_t)(2 * a + 1) * N + 2 * b) : make_int2(0, 0);
                a += qi;
                b += qj;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool same = x == y && aa[u] == bb[u];          // within an individual: only the pair (2a, 2a + 1)
                if (MODE == 1) {
                    if (c0[u].x >= thr) {
                        const double cd = (double)c0[u].x, r = rcp_counts(cd);
                        if (!same) sum += quot_counts((double)d0[u].x, cd, r);
                        sum += quot_counts((double)d0[u].y, cd, r);
                        if (!same) {
                            sum += quot_counts((double)d1[u].x, cd, r);
                            sum += quot_counts((double)d1[u].y, cd, r);
                        }
                        cnt += same ? 1 : 4;
#ifdef PG_DIV_PROBE
                        const int dd[4] = {d0[u].x, d0[u].y, d1[u].x, d1[u].y};
                        for (int k = 0; k < 4; ++k) {
                            atomicAdd(&g_div_probe[0], 1ull);
                            if (quot_counts((double)dd[k], cd, r) != (double)dd[k] / cd) atomicAdd(&g_div_probe[1], 1ull);
                        }
#endif
                    }
                } else {
                    const int cc[4] = {c0[u].x, c0[u].y, c1[u].x, c1[u].y}, dd[4] = {d0[u].x, d0[u].y, d1[u].x, d1[u].y};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((!same || k == 1) && cc[k] >= thr) {
                            const double cd = (double)cc[k];
                            sum += quot_counts((double)dd[k], cd, rcp_counts(cd));
                            ++cnt;
                        }
                }
            }
        }
    } else {
        // thread t takes the pairs t, t + 256, ... of the nx x ny rectangle (fixed partition: the float64 sum is reproducible);
        // row / column advance incrementally instead of dividing (nx, ny <= PG_MAX_HAP, so 256 / ny steps stay small)
        const int total = nx * ny;
        const int qi = 256 / (ny > 0 ? ny : 1), qj = 256 - qi * (ny > 0 ? ny : 1);
        int i = xs + (ny > 0 ? (int)threadIdx.x / ny : 0), j = ys + (ny > 0 ? (int)threadIdx.x % ny : 0);
        // four pairs per trip: their eight loads are issued before the first division (the loop was bound by the latency of one
        // dependent load -> divide chain per trip); the quotients are added in the order the one-pair loop added them
        for (int idx = threadIdx.x; idx < total; idx += 1024) {
            int c[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j >= ye) { j -= ny; ++i; }
                const bool live = idx + 256 * u < total && !(x == y && i >= j);
                c[u] = live ? Cw[(size_t)(i >> cshift) * cN + (j >> cshift)] : 0;
                d[u] = live ? Dw[(size_t)i * N + j] : 0;
                i += qi;
                j += qj;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c[u] >= thr) {
                    const double cd = (double)c[u];
                    sum += quot_counts((double)d[u], cd, rcp_counts(cd));
                    ++cnt;
                }
        }
    }
    // one barrier for both totals (the count travels as a double: exact); fixed butterfly + wave order: reproducible
    double tot[2] = {sum, (double)cnt};
    block_sum_multi<2>(tot, shd);
    if (threadIdx.x == 0) {
        const int npairs = n_pops * (n_pops + 1) / 2;
        sum_out[(size_t)win * npairs + pidx] = tot[0];
        cnt_out[(size_t)win * npairs + pidx] = (int64_t)tot[1];
    }
}

void pg_launch_popdist_fin(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                           const int32_t *pop_start, int n_pops, int min_pair_sites, double *sum_out,
                           int64_t *cnt_out, int all_diploid, const int64_t *win_lo, const int64_t *win_hi, long long skip_upto) {
    if (n_win <= 0 || n_pops <= 0) return;
    int npairs = n_pops * (n_pops + 1) / 2;
    if (all_diploid && cshift == 1)
        hipLaunchKernelGGL(k_popdist_fin<1>, dim3(npairs, n_win), dim3(256), 0, st, Cmat, Dmat, N, cN, cshift, pop_start, n_pops,
                           min_pair_sites, sum_out, cnt_out, win_lo, win_hi, skip_upto);
    else if (all_diploid && cshift == 0 && cN == N)
        hipLaunchKernelGGL(k_popdist_fin<2>, dim3(npairs, n_win), dim3(256), 0, st, Cmat, Dmat, N, cN, cshift, pop_start, n_pops,
                           min_pair_sites, sum_out, cnt_out, win_lo, win_hi, skip_upto);
    else
        hipLaunchKernelGGL(k_popdist_fin<0>, dim3(npairs, n_win), dim3(256), 0, st, Cmat, Dmat, N, cN, cshift, pop_start, n_pops,
                           min_pair_sites, sum_out, cnt_out, win_lo, win_hi, skip_upto);
#ifdef PG_DIV_PROBE
    {
        unsigned long long h[2];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_div_probe), sizeof h);
        fprintf(stderr, "k_popdist_fin: %llu quotients compared with d / c, %llu differ\n", h[0], h[1]);
    }
#endif
}

// ------------------------------------------------------------------------------------------------------
// K_popstats: pi / dxy / Fst from the per-pair sums and counts, one thread per (window, population pair x<=y), with the
// float64 operations of genomics.py:88-90 (nanmean_min) and 976-993 in the reference's order (-ffp-contract=off):
//   nanmean_min(block) = nan if 1 - (1.*n_nan/size) < minData (or nothing valid) else sum/count
//   pi_x   : block (x,x) holds every unordered pair twice plus a nan diagonal      -> (2*sum)/(2*cnt), size n_x^2
//   dxy    : block (x,y)                                                              -> sum/cnt, size n_x*n_y
//   Fst    : w = 1.*n_x/(n_x+n_y); pi_s = w*pi_x + (1-w)*pi_y; pi_t over block (x+y, x+y); Fst = 1 - pi_s/pi_t
// out[w][0..P) = pi, out[w][P + k] = dxy of the k-th pair (x<y, x-major), out[w][P + npo + k] = Fst.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double nanmean_min_dev(double total, long long n_valid, long long size, double min_data) {
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    if (size == 0) return nan;
    const bool frac_ok = (1 - (1. * (double)(size - n_valid) / (double)size)) >= min_data;
    if (!frac_ok || n_valid <= 0) return nan;
    return total / (double)n_valid;
}

__global__ __launch_bounds__(256) void k_popstats(const double *__restrict__ sums, const int64_t *__restrict__ cnts, int n_win,
                                                  const int32_t *__restrict__ pop_start, int n_pops, double min_data, int do_pairs,
                                                  double *__restrict__ out, const int64_t *__restrict__ win_lo,
                                                  const int64_t *__restrict__ win_hi, long long skip_upto) {
    const int npairs = n_pops * (n_pops + 1) / 2;
    const int npo = n_pops * (n_pops - 1) / 2;
    const int ncols = n_pops + (do_pairs ? 2 * npo : 0);
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n_win * npairs) return;
    const int win = (int)(idx / npairs), pidx = (int)(idx % npairs);
    if (win_lo && win_hi[win] - win_lo[win] <= skip_upto) return;
    int x = 0, rem = pidx;
    while (rem >= n_pops - x) { rem -= n_pops - x; ++x; }
    const int y = x + rem;
    const double *S = sums + (size_t)win * npairs;
    const int64_t *Cn = cnts + (size_t)win * npairs;
    const long long nx = pop_start[x + 1] - pop_start[x], ny = pop_start[y + 1] - pop_start[y];
    auto kdiag = [&](int p) { return p * n_pops - p * (p - 1) / 2; };
    auto pi_of = [&](int p, long long np_) { return nanmean_min_dev(2 * S[kdiag(p)], 2 * Cn[kdiag(p)], np_ * np_, min_data); };
    double *O = out + (size_t)win * ncols;
    if (x == y) {
        O[x] = pi_of(x, nx);
        return;
    }
    if (!do_pairs) return;
    // index of (x,y) among the x<y pairs, x-major
    const int k = x * n_pops - x * (x + 1) / 2 + (y - x - 1);
    O[n_pops + k] = nanmean_min_dev(S[pidx], Cn[pidx], nx * ny, min_data);
    const double w = 1. * (double)nx / (double)(nx + ny);
    const double pi_s = w * pi_of(x, nx) + (1 - w) * pi_of(y, ny);
    const double tot = 2 * S[kdiag(x)] + 2 * S[kdiag(y)] + 2 * S[pidx];
    const long long cnt = 2 * Cn[kdiag(x)] + 2 * Cn[kdiag(y)] + 2 * Cn[pidx];
    const double pi_t = nanmean_min_dev(tot, cnt, (nx + ny) * (nx + ny), min_data);
    O[n_pops + npo + k] = 1 - pi_s / pi_t;
}

// ------------------------------------------------------------------------------------------------------
// K_sample_het: Alignment.sampleHet (genomics.py:918-929) on the cached matrix: thread per (window, individual).
// The reference's `len(x)==2 & np.sum(mask & mask) >= _minSites` parses as `len(x) == (2 & C) >= 1`: a value is reported only
// for a diploid individual whose jointly called site count has bit 1 set; the value is the cached distance of its two
// haplotypes, i.e. D/C, nan where a preceding groupDistStats masked the pair (C < its minSites, genomics.py:959-961).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_het(const int32_t *__restrict__ Cmat, const int32_t *__restrict__ Dmat, int N,
                                                    int cN, int cshift, int n_win, const int32_t *__restrict__ samp_start,
                                                    int n_samp, int min_pair_sites, double *__restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n_win * n_samp) return;
    const int win = (int)(idx / n_samp), s = (int)(idx % n_samp);
    const int a = samp_start[s], pl = samp_start[s + 1] - a;
    double v = __longlong_as_double(0x7FF8000000000000ll);
    if (pl == 2) {
        const int c = Cmat[(size_t)win * cN * cN + (size_t)(a >> cshift) * cN + ((a + 1) >> cshift)];
        if ((c & 2) && c >= (min_pair_sites > 1 ? min_pair_sites : 1))
            v = (double)Dmat[(size_t)win * N * N + (size_t)a * N + a + 1] / (double)c;
    }
    out[idx] = v;
}

void pg_launch_sample_het(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                          const int32_t *samp_start, int n_samp, int min_pair_sites, double *out) {
    if (n_win <= 0 || n_samp <= 0) return;
    const long long total = (long long)n_win * n_samp;
    hipLaunchKernelGGL(k_sample_het, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Cmat, Dmat, N, cN, cshift, n_win,
                       samp_start, n_samp, min_pair_sites, out);
}

// ------------------------------------------------------------------------------------------------------
// K_hapstats: Alignment.H12stats (genomics.py:1079-1098) with the greedy clustering of distMat_to_cluster_sizes
// (genomics.py:1239-1261).  Block per (population, window):
//   1. match[i][j] = dist(i,j) <= maxDist as bit rows (global scratch, L2 resident); dist = D/C in float64 as the reference
//      forms it, nan (no match) where nothing is jointly called or a preceding groupDistStats masked the pair; the diagonal
//      is 0.0 (a match when maxDist >= 0) unless an earlier step of the reference's worker left it nan (diag_nan);
//   2. greedy extraction: the alive row with the most alive matches (ties: the first row in the reference's row order, i.e.
//      haplotype names sorted) founds a cluster of that size and its matches leave; a best row with <= 1 match ends the
//      loop, every remaining row is a singleton;
//   3. f = sizes / n;  H1 = sum f^2,  H12 = H1 + 2 f0 f1,  H2 = sum_{k>=1} f_k^2  (H12 = H1, H2 = 0 for a single cluster).
// `order[pop_start[p] .. pop_start[p+1])` = the population's slots in the reference's row order.
// ------------------------------------------------------------------------------------------------------
// float64 sum in the order NumPy's add.reduce visits a contiguous array (the reference's `(clusterFreq**2).sum()`,
// genomics.py:1088-1091): fewer than 8 elements left to right; up to 128 in eight strided partial sums combined as a tree, the
// tail added one by one; beyond that the range is halved (the first half rounded down to a multiple of 8).
template <int DEPTH>
__device__ double np_pairwise_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128 || DEPTH == 0) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum<(DEPTH > 0 ? DEPTH - 1 : 0)>(a, n2) + np_pairwise_sum<(DEPTH > 0 ? DEPTH - 1 : 0)>(a + n2, n - n2);
}

__global__ __launch_bounds__(256) void k_hapstats(const int32_t *__restrict__ Cmat, const int32_t *__restrict__ Dmat, int N,
                                                  int cN, int cshift, const int32_t *__restrict__ pop_start, int n_pops,
                                                  const int32_t *__restrict__ order, int min_pair_sites, int diag_nan,
                                                  double max_dist, uint32_t *__restrict__ bits, size_t bits_per_window,
                                                  double *__restrict__ out) {
    extern __shared__ double sq[];                         // f^2 of the clusters in extraction order (n values at most), then
    __shared__ int best_c[256], best_i[256];
    const int p = blockIdx.x, win = blockIdx.y, tid = threadIdx.x;
    const int s0 = pop_start[p], n = pop_start[p + 1] - s0, nw = (n + 31) >> 5;
    uint32_t *alive = reinterpret_cast<uint32_t *>(sq + n);                 // ceil(n/32) words
    size_t off = 0;
    for (int q = 0; q < p; ++q) {
        const size_t nq = pop_start[q + 1] - pop_start[q];
        off += nq * ((nq + 31) >> 5);
    }
    uint32_t *M = bits + (size_t)win * bits_per_window + off;
    const int32_t *Cw = Cmat + (size_t)win * cN * cN;
    const int32_t *Dw = Dmat + (size_t)win * N * N;
    const int thr = min_pair_sites > 1 ? min_pair_sites : 1;
    double *O = out + ((size_t)win * n_pops + p) * 3;
    if (n == 0) {
        if (tid == 0) O[0] = O[1] = O[2] = 0.0;
        return;
    }
    for (int idx = tid; idx < n * nw; idx += 256) {
        const int i = idx / nw, wd = idx - i * nw;
        const int hi = order[s0 + i];
        uint32_t word = 0u;
        for (int b = 0; b < 32; ++b) {
            const int j = wd * 32 + b;
            if (j >= n) break;
            bool m;
            if (j == i) {
                m = !diag_nan && 0.0 <= max_dist;
            } else {
                const int hj = order[s0 + j];
                const int lo = hi < hj ? hi : hj, up = hi < hj ? hj : hi;
                const int c = Cw[(size_t)(lo >> cshift) * cN + (up >> cshift)];
                m = c >= thr && (double)Dw[(size_t)lo * N + up] / (double)c <= max_dist;
            }
            word |= (uint32_t)m << b;
        }
        M[idx] = word;
    }
    for (int w = tid; w < nw; w += 256) alive[w] = (w == nw - 1 && (n & 31)) ? ((1u << (n & 31)) - 1u) : 0xFFFFFFFFu;
    __syncthreads();
    double f0 = 0.0, f1 = 0.0;                              // kept by every thread identically (block-uniform control flow)
    int k = 0;
    const double dn = (double)n;
    for (;;) {
        int bc = -1, bi = 0x7fffffff;
        for (int i = tid; i < n; i += 256) {
            if (!((alive[i >> 5] >> (i & 31)) & 1u)) continue;
            int c = 0;
            for (int w = 0; w < nw; ++w) c += __popc(M[(size_t)i * nw + w] & alive[w]);
            if (c > bc) { bc = c; bi = i; }                 // rows visited in increasing order: the first maximum stays
        }
        best_c[tid] = bc;
        best_i[tid] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const int c2 = best_c[tid + s], i2 = best_i[tid + s];
                if (c2 > best_c[tid] || (c2 == best_c[tid] && i2 < best_i[tid])) { best_c[tid] = c2; best_i[tid] = i2; }
            }
            __syncthreads();
        }
        const int matches = best_c[0], most = best_i[0];
        __syncthreads();
        if (matches < 0) break;                             // no row left
        if (matches > 1) {
            const double f = (double)matches / dn;
            if (k == 0) f0 = f; else if (k == 1) f1 = f;
            if (tid == 0) sq[k] = f * f;
            ++k;
            for (int w = tid; w < nw; w += 256) alive[w] &= ~M[(size_t)most * nw + w];
            __syncthreads();
        } else {
            int rem = 0;
            for (int w = 0; w < nw; ++w) rem += __popc(alive[w]);
            const double f = 1.0 / dn;
            for (int r = 0; r < rem; ++r) {
                if (k == 0) f0 = f; else if (k == 1) f1 = f;
                if (tid == 0) sq[k] = f * f;
                ++k;
            }
            break;
        }
    }
    if (tid == 0) {
        const double H1 = np_pairwise_sum<9>(sq, k);
        O[0] = H1;
        O[1] = k > 1 ? H1 + 2 * f0 * f1 : H1;
        O[2] = k > 1 ? np_pairwise_sum<9>(sq + 1, k - 1) : 0.0;
    }
}

void pg_launch_hapstats(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                        const int32_t *pop_start, int n_pops, int max_pop, const int32_t *order, int min_pair_sites, int diag_nan,
                        double max_dist, uint32_t *bits, size_t bits_per_window, double *out) {
    if (n_win <= 0 || n_pops <= 0) return;
    const size_t lds = (size_t)max_pop * 8 + (size_t)((max_pop + 31) / 32 + 1) * 4;
    hipLaunchKernelGGL(k_hapstats, dim3(n_pops, n_win), dim3(256), lds, st, Cmat, Dmat, N, cN, cshift, pop_start, n_pops, order,
                       min_pair_sites, diag_nan, max_dist, bits, bits_per_window, out);
}

// ------------------------------------------------------------------------------------------------------
// K_unpack: packed genotype cells -> resident rows, on the device (the host ships 1 byte per diploid
// genotype -- first allele's one-hot code in the low nibble, second allele's in the high nibble, the `.pgeno` cell -- instead of
// 1 byte per allele; replaces the host half of what splitSeq / seqArrayToNumArray do in the reference, genomics.py:390-396,
// 74-77).  Thread = 4 slots of one row; slot_src[slot] = 2 * cell column + allele (or -1: slot stays 0 = missing).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unpack(const uint8_t *__restrict__ cells, int n_cols, int64_t n_rows,
                                                const int32_t *__restrict__ slot_src, int n_hap, int8_t *__restrict__ gt, int S) {
    const int groups = S >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = idx / groups;
    if (row >= n_rows) return;
    const int g = (int)(idx - row * groups);
    const uint8_t *c = cells + row * n_cols;
    uint32_t word = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = 4 * g + k;
        if (h >= n_hap) break;
        const int src = slot_src[h];
        if (src < 0) continue;
        const uint32_t cell = c[src >> 1];
        word |= ((src & 1) ? (cell >> 4) : (cell & 15u)) << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(gt + row * (int64_t)S + 4 * g) = word;
}

void pg_launch_unpack(hipStream_t st, const uint8_t *cells, int n_cols, int64_t n_rows, const int32_t *slot_src, int n_hap,
                      int8_t *gt, int S) {
    const int64_t groups = S >> 2;
    if (n_rows <= 0 || groups <= 0) return;
    const int64_t rows_per_launch = std::max<int64_t>(1, (1ll << 30) / groups);     // a launch holds at most 2^30 threads
    for (int64_t a = 0; a < n_rows; a += rows_per_launch) {
        const int64_t n = std::min(rows_per_launch, n_rows - a);
        hipLaunchKernelGGL(k_unpack, dim3((unsigned)((n * groups + 255) / 256)), dim3(256), 0, st, cells + a * n_cols, n_cols, n,
                           slot_src, n_hap, gt + a * (int64_t)S, S);
    }
}

// the diploid-shortcut verdict as a double next to the result table; re-arms the flag
__global__ void k_flag_export(int32_t *__restrict__ flag, double *__restrict__ dst) {
    dst[0] = (double)flag[0];
    flag[0] = 0;
}

void pg_launch_flag_export(hipStream_t st, int32_t *flag, double *dst) {
    hipLaunchKernelGGL(k_flag_export, dim3(1), dim3(1), 0, st, flag, dst);
}

void pg_launch_popstats(hipStream_t st, const double *sums, const int64_t *cnts, int n_win, const int32_t *pop_start,
                        int n_pops, double min_data, int do_pairs, double *out, const int64_t *win_lo, const int64_t *win_hi,
                        long long skip_upto) {
    if (n_win <= 0 || n_pops <= 0) return;
    const long long total = (long long)n_win * (n_pops * (n_pops + 1) / 2);
    hipLaunchKernelGGL(k_popstats, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, sums, cnts, n_win, pop_start, n_pops,
                       min_data, do_pairs, out, win_lo, win_hi, skip_upto);
}

// ------------------------------------------------------------------------------------------------------
// k_popdist_np / k_popstats_np: pi / dxy / Fst with the float64 SUMS formed in NumPy's order.
// The reference takes np.nanmean over `distMat[np.ix_(rows, cols)]` (genomics.py:88-90, 976-992): the nan of the block become 0 and
// the flattened (row-major) block is added up by NumPy's pairwise summation -- halves (the left one a multiple of 8 long) down to
// runs of at most 128 values, a run as eight interleaved partial sums ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus its last n%8 values one
// by one.  Any other order differs in the last bits, which shows where a value sits on a rounding tie of the printed digit (small
// windows: quotients of small integers), where Fst is 1 - (pi_s / pi_t of a noise away from 1) = +-0.0, and it is what the goldens'
// tie tolerance was for.  Three kinds of blocks per window: (x, x) for pi, (x, y) for dxy and (x+y, x+y) for the pi_t of Fst --
// rows and columns in the reference's row order (haplotype names sorted, `ref_row`), the pair oriented by the sorted population
// names (np.unique, genomics.py:965: `pop_rank`).  The tree of a block length n is laid out by the host (pg_abi.cpp np_tree):
// leaf runs, and the inner nodes level by level.  One block of 256 threads per (task, window): 8 runs at a time are staged in LDS
// as quotients (coalesced over the flattened index), eight lanes add up a run, the levels of the tree follow.
// ------------------------------------------------------------------------------------------------------
#define PG_NP_STAGE_LEAVES 8

// The pair kernels store the upper triangles of C and D.  k_popdist_np walks whole rows of symmetric blocks: reading the lower half
// through the upper one (element (j, i) for i > j) turns every wave's load into 64 cache lines.  k_mirror_lower copies the upper
// triangle into the lower one once per window (32 x 32 tiles through LDS: coalesced rows in, coalesced rows out).
__global__ __launch_bounds__(256) void k_mirror_lower(int32_t *__restrict__ M, int n, const int64_t *__restrict__ win_lo,
                                                      const int64_t *__restrict__ win_hi, long long max_sites) {
    __shared__ int32_t tile[32][33];
    const int win = blockIdx.y;
    if (win_hi[win] - win_lo[win] > max_sites) return;
    const int nt = (n + 31) / 32;
    // blockIdx.x -> tile (ti <= tj) of the upper triangle
    int t = blockIdx.x, ti = 0;
    while (t >= nt - ti) { t -= nt - ti; ++ti; }
    const int tj = ti + t;
    int32_t *Mw = M + (size_t)win * n * n;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int i = ti * 32 + r, j = tj * 32 + lx;
        tile[r][lx] = (i < n && j < n) ? Mw[(size_t)i * n + j] : 0;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int i = tj * 32 + r, j = ti * 32 + lx;           // the mirrored tile: rows of tj, columns of ti
        if (i < n && j < n && i > j) Mw[(size_t)i * n + j] = tile[lx][r];
    }
}
__global__ __launch_bounds__(256) void k_popdist_np(const int32_t *__restrict__ Cmat, const int32_t *__restrict__ Dmat, int N, int cN,
                                                    int cshift, const int32_t *__restrict__ pop_start, int n_pops,
                                                    const int32_t *__restrict__ ref_row, const int32_t *__restrict__ pop_rank,
                                                    const int32_t *__restrict__ task_tree, const int32_t *__restrict__ trees,
                                                    int min_pair_sites, int max_leaves, double *__restrict__ sum_out,
                                                    int64_t *__restrict__ cnt_out, const int64_t *__restrict__ win_lo,
                                                    const int64_t *__restrict__ win_hi, long long max_sites) {
    extern __shared__ double np_lds[];
    if (win_hi[blockIdx.y] - win_lo[blockIdx.y] > max_sites) return;          // a long window: the fixed-tree finisher's
    double *slots = np_lds;                                  // [2 * max_leaves]: run sums, then the inner nodes
    double *q = np_lds + 2 * (size_t)max_leaves;             // [PG_NP_STAGE_LEAVES * 128]
    __shared__ unsigned long long shc[4];
    const int task = blockIdx.x, win = blockIdx.y, tid = threadIdx.x;
    const int n_tasks = gridDim.x;
    // task -> the two row segments A (then B for pi_t) and the column segments
    int pa, pb, kind;                                        // kind 0: pi of pa; 1: dxy rows pa, columns pb; 2: pi_t of pa + pb
    if (task < n_pops) { pa = pb = task; kind = 0; }
    else {
        const int u = task - n_pops, k = u >> 1;
        int x = 0, rem = k;
        while (rem >= n_pops - 1 - x) { rem -= n_pops - 1 - x; ++x; }
        const int y = x + 1 + rem;
        const bool xf = pop_rank[x] < pop_rank[y];
        pa = xf ? x : y;
        pb = xf ? y : x;
        kind = 1 + (u & 1);
    }
    const int as = pop_start[pa], na = pop_start[pa + 1] - as, bs = pop_start[pb], nb_ = pop_start[pb + 1] - bs;
    const int nr = kind == 2 ? na + nb_ : na, nc = kind == 0 ? na : kind == 1 ? nb_ : na + nb_;
    const int n = nr * nc;
    const int32_t *T = trees + task_tree[task];              // [L, n_inner, n_levels, leaf_off[L + 1], node_l[], node_r[], level_start[]]
    const int L = T[0], n_inner = T[1], n_levels = T[2];
    // the tree's tables are walked run by run and level by level: from LDS, not with a global-memory latency per level
    int32_t *tree_s = reinterpret_cast<int32_t *>(q + PG_NP_STAGE_LEAVES * 128);
    const int tree_len = (L + 1) + 2 * n_inner + (n_levels + 1);
    for (int t = tid; t < tree_len; t += 256) tree_s[t] = T[3 + t];
    const int32_t *leaf_off = tree_s, *node_l = leaf_off + L + 1, *node_r = node_l + n_inner, *level_start = node_r + n_inner;
    const int32_t *Cw = Cmat + (size_t)win * cN * cN, *Dw = Dmat + (size_t)win * N * N;
    const int thr = min_pair_sites > 1 ? min_pair_sites : 1;
    unsigned long long valid = 0ull;
    // the slots of the block's rows and columns (the reference's row order inside a population)
    int32_t *rowmap = tree_s + tree_len, *colmap = rowmap + nr;
    for (int t = tid; t < nr; t += 256) rowmap[t] = ref_row[t < na ? as + t : bs + (t - na)];
    for (int t = tid; t < nc; t += 256) colmap[t] = ref_row[kind == 1 ? bs + t : (t < na ? as + t : bs + (t - na))];
    __syncthreads();
    for (int l0 = 0; l0 < L; l0 += PG_NP_STAGE_LEAVES) {
        const int l1 = l0 + PG_NP_STAGE_LEAVES < L ? l0 + PG_NP_STAGE_LEAVES : L;
        const int k0 = leaf_off[l0], k1 = leaf_off[l1];
        // quotients of the flattened elements [k0, k1): element k = (row k / nc, column k % nc); four per trip, their loads first
        {
            int k = k0 + tid;
            int r = nc > 0 ? k / nc : 0, cc = nc > 0 ? k - r * nc : 0;
            const int qi = 256 / (nc > 0 ? nc : 1), qj = 256 - qi * (nc > 0 ? nc : 1);
            // every load of the round first (PG_NP_STAGE_LEAVES * 128 / 256 = 4 values a thread): the maps and the staged
            // quotients share LDS, so a store between two trips kept the next trip's loads from being issued early
            constexpr int U = PG_NP_STAGE_LEAVES * 128 / 256;
            int cv[U], dv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cv[u] = 0;
                dv[u] = 0;
                if (k + 256 * u < k1) {
                    const int i = rowmap[r], j = colmap[cc];
                    if (i != j) {                                    // (both triangles hold the counts: k_mirror_lower)
                        cv[u] = Cw[(i >> cshift) * cN + (j >> cshift)];
                        dv[u] = Dw[i * N + j];
                    }
                }
                r += qi;
                cc += qj;
                if (cc >= nc) { cc -= nc; ++r; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k + 256 * u < k1) {
                    double v = 0.0;
                    if (cv[u] >= thr) {
                        const double cd = (double)cv[u];
                        v = quot_counts((double)dv[u], cd, rcp_counts(cd));
                        ++valid;
                    }
                    q[k + 256 * u - k0] = v;
                }
            }
        }
        __syncthreads();
        {
            const int leaf = l0 + (tid >> 3), j = tid & 7;
            if (leaf < l1) {                                     // (the eight lanes of a run share the condition)
                const int off = leaf_off[leaf] - k0, len = leaf_off[leaf + 1] - leaf_off[leaf];
                double res;
                if (len < 8) {                                   // a block of fewer than 8 values: one after the other from 0.
                    res = 0.0;
                    for (int i = 0; i < len; ++i) res = res + q[off + i];
                } else {
                    const int m8 = len - (len & 7);
                    double rj = q[off + j];
                    for (int i = 8; i < m8; i += 8) rj = rj + q[off + i + j];
                    rj = rj + __shfl_xor(rj, 1, 64);
                    rj = rj + __shfl_xor(rj, 2, 64);
                    rj = rj + __shfl_xor(rj, 4, 64);
                    res = rj;
                    for (int i = m8; i < len; ++i) res = res + q[off + i];
                }
                if (j == 0) slots[leaf] = res;
            }
        }
        __syncthreads();
    }
    for (int lv = 0; lv < n_levels; ++lv) {
        for (int idx = level_start[lv] + tid; idx < level_start[lv + 1]; idx += 256) slots[L + idx] = slots[node_l[idx]] + slots[node_r[idx]];
        __syncthreads();
    }
    // the valid elements of the block (an integer, any order)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) valid += __shfl_xor(valid, m, 64);
    if ((tid & 63) == 0) shc[tid >> 6] = valid;
    __syncthreads();
    if (tid == 0) {
        const double root = L > 0 ? slots[n_inner > 0 ? L + n_inner - 1 : 0] : 0.0;
        sum_out[(size_t)win * n_tasks + task] = 0.0 + root;       // np.add.reduce starts from the identity
        cnt_out[(size_t)win * n_tasks + task] = (int64_t)(shc[0] + shc[1] + shc[2] + shc[3]);
    }
    (void)n;
}

// pi / dxy / Fst from k_popdist_np's sums and counts (task order: pi of every population, then dxy and pi_t of every pair x < y,
// x-major); the float64 operations of genomics.py:88-90 and 976-993 in the reference's order, the pair oriented by pop_rank
__global__ __launch_bounds__(256) void k_popstats_np(const double *__restrict__ sums, const int64_t *__restrict__ cnts, int n_win,
                                                     const int32_t *__restrict__ pop_start, const int32_t *__restrict__ pop_rank,
                                                     int n_pops, double min_data, int do_pairs, double *__restrict__ out,
                                                     const int64_t *__restrict__ win_lo, const int64_t *__restrict__ win_hi,
                                                     long long max_sites) {
    const int npairs = n_pops * (n_pops + 1) / 2;
    const int npo = n_pops * (n_pops - 1) / 2;
    const int n_tasks = n_pops + 2 * npo;
    const int ncols = n_pops + (do_pairs ? 2 * npo : 0);
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n_win * npairs) return;
    const int win = (int)(idx / npairs), pidx = (int)(idx % npairs);
    if (win_hi[win] - win_lo[win] > max_sites) return;
    int x = 0, rem = pidx;
    while (rem >= n_pops - x) { rem -= n_pops - x; ++x; }
    const int y = x + rem;
    const double *S = sums + (size_t)win * n_tasks;
    const int64_t *Cn = cnts + (size_t)win * n_tasks;
    auto size_of = [&](int p) { return (long long)(pop_start[p + 1] - pop_start[p]); };
    auto pi_of = [&](int p) { return nanmean_min_dev(S[p], Cn[p], size_of(p) * size_of(p), min_data); };
    double *O = out + (size_t)win * ncols;
    if (x == y) {
        O[x] = pi_of(x);
        return;
    }
    if (!do_pairs) return;
    const int k = x * n_pops - x * (x + 1) / 2 + (y - x - 1);
    const bool xf = pop_rank[x] < pop_rank[y];
    const int a = xf ? x : y, b = xf ? y : x;
    const long long na = size_of(a), nb = size_of(b);
    O[n_pops + k] = nanmean_min_dev(S[n_pops + 2 * k], Cn[n_pops + 2 * k], na * nb, min_data);
    const double w = 1. * (double)na / (double)(na + nb);
    const double pi_s = w * pi_of(a) + (1 - w) * pi_of(b);
    const double pi_t = nanmean_min_dev(S[n_pops + 2 * k + 1], Cn[n_pops + 2 * k + 1], (na + nb) * (na + nb), min_data);
    O[n_pops + npo + k] = 1 - pi_s / pi_t;
}

void pg_launch_popdist_np(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                          const int32_t *pop_start, int n_pops, const int32_t *ref_row, const int32_t *pop_rank,
                          const int32_t *task_tree, const int32_t *trees, int max_leaves, int max_side, int min_pair_sites,
                          double min_data, int do_pairs, double *sums, int64_t *cnts, double *out, const int64_t *win_lo,
                          const int64_t *win_hi, long long max_sites) {
    if (n_win <= 0 || n_pops <= 0) return;
    const int n_tasks = n_pops * n_pops;                     // P + 2 * P (P - 1) / 2
    // run sums and inner nodes, the staged quotients, the row and column maps
    const size_t lds = (2 * (size_t)max_leaves + PG_NP_STAGE_LEAVES * 128) * sizeof(double) +
                       (3 * (size_t)max_leaves + 64 + (size_t)max_side + 8) * sizeof(int32_t);
    {
        const int ntD = (N + 31) / 32, ntC = (cN + 31) / 32;
        hipLaunchKernelGGL(k_mirror_lower, dim3(ntD * (ntD + 1) / 2, n_win), dim3(256), 0, st, const_cast<int32_t *>(Dmat), N, win_lo, win_hi,
                           max_sites);
        hipLaunchKernelGGL(k_mirror_lower, dim3(ntC * (ntC + 1) / 2, n_win), dim3(256), 0, st, const_cast<int32_t *>(Cmat), cN, win_lo, win_hi,
                           max_sites);
    }
    hipLaunchKernelGGL(k_popdist_np, dim3(n_tasks, n_win), dim3(256), lds, st, Cmat, Dmat, N, cN, cshift, pop_start, n_pops, ref_row,
                       pop_rank, task_tree, trees, min_pair_sites, max_leaves, sums, cnts, win_lo, win_hi, max_sites);
    const long long total = (long long)n_win * (n_pops * (n_pops + 1) / 2);
    hipLaunchKernelGGL(k_popstats_np, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, sums, cnts, n_win, pop_start, pop_rank,
                       n_pops, min_data, do_pairs, out, win_lo, win_hi, max_sites);
}

// ------------------------------------------------------------------------------------------------------
// K_indpair: one thread per unordered individual pair (s<=t); haplotype slots of an individual are contiguous.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_indpair_fin(const int32_t *__restrict__ Cmat, const int32_t *__restrict__ Dmat,
                                                     int N, int cN, int cshift, const int32_t *__restrict__ samp_start,
                                                     const int32_t *__restrict__ samp_rank, int n_samp,
                                                     int min_pair_sites, double *__restrict__ sum_out,
                                                     int64_t *__restrict__ cnt_out, int mean_mode) {
    const int win = blockIdx.y;
    const long long npairs = (long long)n_samp * (n_samp + 1) / 2;
    const int32_t *Cw = Cmat + (size_t)win * cN * cN;
    const int32_t *Dw = Dmat + (size_t)win * N * N;
    const int thr = min_pair_sites > 1 ? min_pair_sites : 1;
    for (long long pidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs;
         pidx += (long long)gridDim.x * blockDim.x) {
        // row s of the upper triangle starts at s*n - s(s-1)/2
        long long lo = 0, hi = n_samp - 1;
        while (lo < hi) {
            long long mid = (lo + hi + 1) >> 1;
            long long start = mid * n_samp - mid * (mid - 1) / 2;
            if (start <= pidx) lo = mid; else hi = mid - 1;
        }
        const int s = (int)lo;
        const int t = s + (int)(pidx - ((long long)s * n_samp - (long long)s * (s - 1) / 2));
        double sum = 0.0;
        long long cnt = 0;
        // np.nanmean adds the haplotype block up row by row (fewer than 8 values: one after the other): the rows are the haplotypes of the
        // individual the caller names first -- `pairDistDict[i][j]` with i before j in sorted names (popgenWindows.py:55-57) or in the
        // order of the samples (distMat.py:44-45): samp_rank (pg_set_sample_rank; slot order until set)
        const int as = samp_start[s], ns = samp_start[s + 1] - as, bs = samp_start[t], nt = samp_start[t + 1] - bs;
        const bool s_rows = samp_rank[s] <= samp_rank[t];
        for (int u = 0; u < ns * nt; ++u) {
            const int a = as + (s_rows ? u / nt : u % ns), b = bs + (s_rows ? u % nt : u / ns);
            if (s == t && a >= b) continue;
            const int c = Cw[(size_t)(a >> cshift) * cN + (b >> cshift)];
            if (c >= thr) { sum += (double)Dw[(size_t)a * N + b] / (double)c; ++cnt; }
        }
        if (mean_mode == 0) {
            sum_out[(size_t)win * npairs + pidx] = sum;
            cnt_out[(size_t)win * npairs + pidx] = cnt;
        } else {
            // nanmean of the haplotype block (genomics.py:946-947): within an individual the symmetric block holds every pair
            // twice, plus, when the diagonal counts as data (includeSameWithSame, genomics.py:940), one zero per haplotype
            const double nan = __longlong_as_double(0x7ff8000000000000ll);
            double v;
            if (s != t) v = cnt > 0 ? sum / (double)cnt : nan;
            else if (mean_mode == 1) v = cnt > 0 ? (2 * sum) / (double)(2 * cnt) : nan;
            else v = (2 * sum) / (double)(2 * cnt + (samp_start[s + 1] - samp_start[s]));
            sum_out[(size_t)win * npairs + pidx] = v;
        }
    }
}

void pg_launch_indpair_fin(hipStream_t st, const int32_t *Cmat, const int32_t *Dmat, int N, int cN, int cshift, int n_win,
                           const int32_t *samp_start, const int32_t *samp_rank, int n_samp, int min_pair_sites, double *sum_out,
                           int64_t *cnt_out, int mean_mode) {
    if (n_win <= 0 || n_samp <= 0) return;
    long long npairs = (long long)n_samp * (n_samp + 1) / 2;
    int bx = (int)((npairs + 255) / 256);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(k_indpair_fin, dim3(bx, n_win), dim3(256), 0, st, Cmat, Dmat, N, cN, cshift, samp_start, samp_rank, n_samp,
                       min_pair_sites, sum_out, cnt_out, mean_mode);
}

// ------------------------------------------------------------------------------------------------------
// Per-site population base counts (SWAR on one-hot bytes): cnt[b] = #haplotypes of [s,e) with allele b.
// `row` = the site's S/4 dwords (global memory or an LDS tile row); only the first and last dword of a range need a mask.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void count_dword(uint32_t v, uint32_t cnt[4]) {
    cnt[0] += __popc(v & 0x01010101u);
    cnt[1] += __popc(v & 0x02020202u);
    cnt[2] += __popc(v & 0x04040404u);
    cnt[3] += __popc(v & 0x08080808u);
}

__device__ __forceinline__ void range_counts(const uint32_t *__restrict__ row, int s, int e, uint32_t cnt[4]) {
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0u;
    if (e <= s) return;
    const int d0 = s >> 2, d1 = (e - 1) >> 2;
    const uint32_t m_first = ~((1u << (8 * (s & 3))) - 1u);
    const int hi = ((e - 1) & 3) + 1;
    const uint32_t m_last = hi == 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u);
    if (d0 == d1) {
        count_dword(row[d0] & m_first & m_last, cnt);
        return;
    }
    count_dword(row[d0] & m_first, cnt);
    for (int d = d0 + 1; d < d1; ++d) count_dword(row[d], cnt);
    count_dword(row[d1] & m_last, cnt);
}

// ------------------------------------------------------------------------------------------------------
// K_abba: grid (chunk, window).  Per-site terms follow genomics.py:1409-1475 and 1565-1569 operation for operation;
// per-block partial sums are combined by k_abba_reduce in chunk order (deterministic).
// ------------------------------------------------------------------------------------------------------
// flag_site: one bit per site for the kernels that add a window's per-site values up in site order (k_popfreq_ordered: every slot
// called and polymorphic within some population; k_quartet_np: the used sites of ABBABABA / fourPop)
__device__ __forceinline__ void flag_site(uint32_t *__restrict__ flags, int64_t rel) {
    atomicOr(&flags[rel >> 5], 1u << (rel & 31));
}

__device__ __forceinline__ double f4_term(double a, double b, double c, double d) {
    return (1 - a) * b * c * (1 - d) - a * (1 - b) * c * (1 - d);
}

template <int NSUM>
struct QuartetAcc {
    double acc[NSUM];
    unsigned long long used;
    unsigned long long good;     // sites that are biallelic with enough data in every population (before the allele choice)
};

// Float64 phase for one usable site.  e = { c1, c2, c3, n1, n2, n3, n4, c4 }: allele counts / called counts of the chosen
// allele in P1, P2, P3, P4.  The first six sums are the ones genomics.ABBABABA needs (operation order of
// genomics.py:1409-1475, 1565-1569); with NSUM == PG_FOURPOP_NSUM the remaining terms of genomics.fourPop
// (genomics.py:1413-1563) follow, again operation for operation.
__device__ __forceinline__ double np_max(double a, double b) { return a != a ? a : (b != b ? b : (a > b ? a : b)); }

__device__ __forceinline__ double f4c_term(double a, double b, double c, double d) {
    return f4_term(a, b, c, d) + f4_term(1 - a, 1 - b, 1 - c, 1 - d);                   // :1413-1418
}

template <int NSUM>
__device__ __forceinline__ void quartet_values(const uint32_t e[8], double (&t)[NSUM]) {
    const double p1 = (double)e[0] / (double)e[3];
    const double p2 = (double)e[1] / (double)e[4];
    const double p3 = (double)e[2] / (double)e[5];
    const double p4 = (double)e[7] / (double)e[6];
    const double abba = (1 - p1) * p2 * p3 * (1 - p4);
    const double baba = p1 * (1 - p2) * p3 * (1 - p4);
    t[0] = f4_term(p1, p2, p3, p4);
    t[1] = abba + baba;
    const double pd = p2 * (double)(p2 > p3) + p3 * (double)(p3 >= p2);              // :1446
    t[2] = f4_term(p1, pd, pd, p4);
    const bool a = p3 > p1, bb = p3 > p2, x = p1 > p2, y = !x;                       // :1460-1468
    const double xa = (double)(x && a), nxa = (double)(!(x && a));
    const double yb = (double)(y && bb), nyb = (double)(!(y && bb));
    const double pdm1 = p3 * xa + p1 * nxa;
    const double pdm2 = p3 * yb + p2 * nyb;
    const double pdm3 = -p3 * xa + p3 * yb - p1 * (double)(x && !a) + p2 * (double)(y && !bb);
    t[3] = f4_term(pdm1, pdm2, pdm3, p4);
    t[4] = abba;
    t[5] = baba;
    if constexpr (NSUM > PG_ABBA_NSUM) {
        t[6] = f4c_term(p1, p2, p3, p4);                                              // fd', fdm', fdh, fdh2, fh numerators
        t[7] = f4c_term(p1, pd, pd, p4);                                              // :1455-1456
        t[8] = f4c_term(pdm1, pdm2, pdm3, p4);                                        // :1483-1486
        const double t11 = f4c_term(p1, p3, p3, p4), t12 = f4c_term(p4, p2, p3, p4);
        const double t21 = f4c_term(p3, p2, p3, p4), t22 = f4c_term(p1, p4, p3, p4);
        double m = np_max(np_max(np_max(t11, t12), t21), t22);                        // np.amax: NaN propagates
        t[9] = m;                                                                     // :1495-1502
        const double t31 = f4c_term(p1, p2, p2, p4), t32 = f4c_term(p1, p2, p3, p1);
        const double t41 = f4c_term(p1, p2, p1, p4), t42 = f4c_term(p1, p2, p3, p2);
        m = np_max(np_max(np_max(np_max(m, t31), t32), t41), t42);
        t[10] = m;                                                                    // :1511-1523
        const double t1 = fabs(p1 - p2), t2 = fabs(p3 - p4);
        const double den = t1 * (double)(t1 > t2) + t2 * (double)(t2 >= t1);
        t[11] = den * den;                                                            // :1551-1554
        t[12] = (1 - p1) * p2 * (1 - p3) * (1 - p4);                                  // ABAA :1557
        t[13] = p1 * (1 - p2) * (1 - p3) * (1 - p4);                                  // BAAA :1560
    }
}

template <int NSUM>
__device__ __forceinline__ void quartet_terms(const uint32_t e[8], QuartetAcc<NSUM> &A) {
    double t[NSUM];
    quartet_values<NSUM>(e, t);
#pragma unroll
    for (int k = 0; k < NSUM; ++k) A.acc[k] += t[k];
    ++A.used;
}

// ------------------------------------------------------------------------------------------------------
// k_abba_q: four lanes per site, one per population (P1,P2,P3,O).  A wave covers 16 consecutive site rows, i.e. a contiguous
// 16*S byte span that stays in L1 while the lanes walk their own population's byte range with 16-byte loads; no LDS row
// staging, so occupancy is not LDS-limited.  The four lanes of a site exchange counts with quad shuffles; usable sites are
// appended, in site order, to an LDS list and the float64 phase runs on full waves of list entries.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void range_counts_x4(const int8_t *__restrict__ rowb, int s, int e, uint32_t cnt[4]) {
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0u;
    if (e <= s) return;
    // bytes [s,e): 16-byte loads from the dword holding s (rows are 16-byte padded and the buffer has slack rows, so reading
    // a little past e is safe); only the first and the last dword of the range need a byte mask
    const int b0 = s & ~3;
    const int last = (e - 1) & ~3;                                         // byte offset of the last dword
    const uint32_t m_first = ~((1u << (8 * (s & 3))) - 1u);
    const int hi = ((e - 1) & 3) + 1;
    const uint32_t m_last = hi == 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u);
    for (int b = b0; b <= last; b += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(rowb + b);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (b == b0) w[0] &= m_first;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = b + 4 * k;
            if (off == last) w[k] &= m_last;
            if (off > last) w[k] = 0u;
            count_dword(w[k], cnt);
        }
    }
}

// OR of the bytes [s,e) of a row: the alleles present among those haplotypes (low 4 bits)
__device__ __forceinline__ uint32_t range_presence_x4(const int8_t *__restrict__ rowb, int s, int e) {
    if (e <= s) return 0u;
    const int b0 = s & ~3;
    const int last = (e - 1) & ~3;
    const uint32_t m_first = ~((1u << (8 * (s & 3))) - 1u);
    const int hi = ((e - 1) & 3) + 1;
    const uint32_t m_last = hi == 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u);
    uint32_t acc = 0u;
    for (int b = b0; b <= last; b += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(rowb + b);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (b == b0) w[0] &= m_first;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = b + 4 * k;
            if (off == last) w[k] &= m_last;
            if (off > last) w[k] = 0u;
        }
        acc |= w[0] | w[1] | w[2] | w[3];
    }
    acc |= acc >> 16;
    acc |= acc >> 8;
    return acc & 0xFu;
}

// Row loads of the screening passes (k_abba_q, k_popfreq_q): 16 lanes per row, NPASS passes of 256 bytes, 4 rows per
// instruction, GROUPS instructions per pass and step.
typedef uint32_t pg_u32x4 __attribute__((ext_vector_type(4)));

template <int GROUPS, int NPASS>
struct ScreenLoads {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[NPASS];                  // lane offset of pass p; beyond the descriptor for lanes past the end of a row
    int soff[GROUPS];                 // g * 4 rows, kept in SGPRs
    int S;
    int64_t c0;
    __device__ __forceinline__ ScreenLoads(const int8_t *gt, int S_, int64_t c0_, int64_t c1, int sub, int rsel) : S(S_), c0(c0_) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(gt + c0_ * (int64_t)S_), 0, (int)(c1 - c0_) * S_, 0x00020000);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int off = p * 256 + sub * 16;
            voff[p] = off < S_ ? rsel * S_ + off : 0x7ffffff0;
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            int t = g * 4 * S_;
            asm volatile("" : "+s"(t));
            soff[g] = t;
        }
    }
    __device__ __forceinline__ void issue(int64_t t0, uint4 (&v)[GROUPS][NPASS]) const {
        const int64_t rel = t0 - c0;
        const int base = rel < (1 << 20) ? (int)rel * S : 0x7ffffff0;          // far past the block: everything reads as zero
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const pg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[p] + base, soff[g], 0);
                v[g][p] = make_uint4(r.x, r.y, r.z, r.w);
            }
    }
};

#define PG_ABBA_RING 128          // per-wave ring of usable sites waiting for the float64 phase
#define PG_ABBA_CAND 128          // per-wave ring of candidate (biallelic) sites waiting for the counting pass

// Allele choice per site (sel):
//   PG_SEL_POLARIZE  the allele present in the four populations and absent from P4 (genomics.py:1672 / :1610)
//   PG_SEL_FIXED     same, and fixed (frequency 0 or 1) in each of P1, P2, P3 (genomics.py:1611-1614)
//   PG_SEL_MINOR     np.argsort(all4freqs)[:,2], i.e. the rarer of the two alleles (genomics.py:1615); a tie is resolved the
//                    way NumPy >= 2.0's x86 (AVX2 / AVX-512) argsort network resolves it for a 4-element row, which is what
//                    the reference produces on current hardware: {A,C}->C, {A,G}->A, {A,T}->A, {C,G}->C, {C,T}->C, {G,T}->G
// NPASS > 0: the screening pass reads whole rows with fully coalesced 16-byte loads (16 lanes per row, NPASS passes of 256
// bytes) and ORs them over the 16 lanes with DPP row shifts; NPASS == 0: rows longer than 1024 bytes, screened with the
// counting pass's quad layout.
template <int NSUM, int NPASS>
__global__ __launch_bounds__(256) void k_abba_q(const int8_t *__restrict__ gt, int S, const int64_t *__restrict__ win_lo,
                                                const int64_t *__restrict__ win_hi, int max_chunks,
                                                const int32_t *__restrict__ pop_start, int q1, int q2, int q3, int q4,
                                                double min_data, int sel, double *__restrict__ part_sums,
                                                int64_t *__restrict__ part_used, uint32_t *__restrict__ flags, int64_t base) {
    __shared__ double shd[4 * (NSUM + 2)];
    __shared__ uint32_t ring[4][PG_ABBA_RING][8];
    __shared__ int64_t cand[4][PG_ABBA_CAND];
    const int win = blockIdx.y, chunk = blockIdx.x;
    const int64_t lo = win_lo[win], hi = win_hi[win];
    const int64_t c0 = lo + (int64_t)chunk * PG_ABBA_SITES_PER_BLOCK;
    QuartetAcc<NSUM> A;
#pragma unroll
    for (int k = 0; k < NSUM; ++k) A.acc[k] = 0.0;
    A.used = 0;
    A.good = 0;
    const int qs[4] = {q1, q2, q3, q4};
    const int q = threadIdx.x & 3;
    const int my_s = pop_start[qs[q]], my_e = pop_start[qs[q] + 1];
    int my_nmin = 0;
    {
        const int Nq = my_e - my_s;
        while (my_nmin <= Nq && !((double)my_nmin * 1. / (double)Nq >= min_data)) ++my_nmin;   // genomics.py:1657-1660, as integers
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qbase = lane & ~3;
    uint32_t(*my_ring)[8] = ring[wave];
    int64_t *my_cand = cand[wave];
    int head = 0, tail = 0;                             // wave-uniform ring cursors (entries [head,tail) are pending)
    int chead = 0, ctail = 0;                           // same for the candidate ring
    // Counting pass for 16 sites (one per quad): per-population allele counts, the reference's site filters, and the
    // usable sites appended to the float64 ring.
    auto count_sites = [&](int64_t site, bool valid) {
            uint32_t cnt[4] = {0u, 0u, 0u, 0u};
            if (valid) range_counts_x4(gt + site * (int64_t)S, my_s, my_e, cnt);
            const uint32_t n = cnt[0] + cnt[1] + cnt[2] + cnt[3];
            int ok = valid && ((int)n >= my_nmin);
            ok &= __shfl_xor(ok, 1, 64);
            ok &= __shfl_xor(ok, 2, 64);
            uint32_t tot[4], c3[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                uint32_t t = cnt[b] + (uint32_t)__shfl_xor((int)cnt[b], 1, 64);
                tot[b] = t + (uint32_t)__shfl_xor((int)t, 2, 64);
                c3[b] = (uint32_t)__shfl((int)cnt[b], qbase + 3, 64);
            }
            const uint32_t n_o = (uint32_t)__shfl((int)n, qbase + 3, 64);
            const int nall = (tot[0] > 0) + (tot[1] > 0) + (tot[2] > 0) + (tot[3] > 0);
            int der = -1;
            bool good = ok && nall == 2;
            // the reference's goodSites (genomics.py:1655-1662): ABBABABA answers a window WITHOUT any with sitesUsed = nan (its
            // zip() of six names with seven values drops the 0, genomics.py:1693-1695), so the windows' counts of them are kept
            const unsigned long long gbal = __ballot(good && q == 0);
            if (lane == 0) A.good += (unsigned long long)__popcll(gbal);
            if (sel == PG_SEL_MINOR) {
                int lo_b = -1, hi_b = -1;                // the two alleles present, lo_b < hi_b
#pragma unroll
                for (int b = 3; b >= 0; --b)
                    if (tot[b] > 0) { if (hi_b < 0) hi_b = b; else lo_b = b; }
                if (good) {
                    uint32_t tl = 0u, th = 0u;
#pragma unroll
                    for (int b = 0; b < 4; ++b) { if (b == lo_b) tl = tot[b]; if (b == hi_b) th = tot[b]; }
                    if (tl < th) der = lo_b;
                    else if (th < tl) der = hi_b;
                    else der = (lo_b == 0 && hi_b == 1) ? 1 : lo_b;
                }
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (tot[b] > 0 && c3[b] == 0) der = b;
                good = good && n_o > 0 && der >= 0;
            }
            uint32_t cder = 0u;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b == der) cder = cnt[b];
            if (sel == PG_SEL_FIXED) {                   // P1,P2,P3 frequency exactly 0 or 1; 0/0 (nan) is neither
                int fx = (q == 3) || (n > 0 && (cder == 0u || cder == n));
                fx &= __shfl_xor(fx, 1, 64);
                fx &= __shfl_xor(fx, 2, 64);
                good = good && fx;
            }
            // gather the quad's chosen-allele counts and called counts into its lane 0
            const uint32_t c_p2 = (uint32_t)__shfl((int)cder, qbase + 1, 64), c_p3 = (uint32_t)__shfl((int)cder, qbase + 2, 64);
            const uint32_t c_p4 = (uint32_t)__shfl((int)cder, qbase + 3, 64);
            const uint32_t n_p2 = (uint32_t)__shfl((int)n, qbase + 1, 64), n_p3 = (uint32_t)__shfl((int)n, qbase + 2, 64);
            const bool writer = good && q == 0;
            const unsigned long long bal = __ballot(writer);
            if (writer) {
                if (flags) flag_site(flags, site - base);                 // a used site: k_quartet_np adds its terms up in site order
                const int before = __popcll(bal & ((1ull << lane) - 1ull));
                uint32_t *e = my_ring[(tail + before) & (PG_ABBA_RING - 1)];
                e[0] = cder; e[1] = c_p2; e[2] = c_p3; e[3] = n; e[4] = n_p2; e[5] = n_p3; e[6] = n_o; e[7] = c_p4;
            }
            tail += (int)__popcll(bal);
            if (tail - head >= 64) {                     // a full wave of usable sites: float64 phase
                uint32_t f[8];
                const uint32_t *e = my_ring[(head + lane) & (PG_ABBA_RING - 1)];
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = e[k];
                quartet_terms<NSUM>(f, A);
                head += 64;
            }
    };
    if (c0 < hi) {
        const int64_t c1 = (c0 + PG_ABBA_SITES_PER_BLOCK < hi) ? c0 + PG_ABBA_SITES_PER_BLOCK : hi;
        // Screening pass, every site: which alleles occur among the called haplotypes of the four populations (an OR of the
        // row bytes, a fraction of the counting work).  Only sites with exactly two alleles can pass genomics.py:1655 / :1593;
        // they are queued in site order and counted 16 at a time (their rows are re-read, from cache where possible).
        // A wave owns 16 consecutive sites per step; the whole loop is wave-synchronous (no block barrier).
        // bytes of this lane's 16-byte pieces that belong to one of the four populations
        uint32_t umask[NPASS > 0 ? NPASS : 1][4];
        const int sub = lane & 15, rsel = lane >> 4;
        if (NPASS > 0) {
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t m = 0u;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int x = p * 256 + sub * 16 + 4 * k + b;
                        bool in = false;
#pragma unroll
                        for (int z = 0; z < 4; ++z) in = in || (x >= pop_start[qs[z]] && x < pop_start[qs[z] + 1]);
                        if (in) m |= 0xFFu << (8 * b);
                    }
                    umask[p][k] = m;
                }
        }
        if (NPASS > 0) {
            // see ScreenLoads: zero-filling buffer loads, the loads of step k+1 issued before step k is processed
            constexpr int GROUPS = NPASS > 0 ? 8 / NPASS : 1, SPW = 4 * GROUPS, NP1 = NPASS > 0 ? NPASS : 1;
            const ScreenLoads<GROUPS, NP1> sl(gt, S, c0, c1, sub, rsel);
            auto process = [&](int64_t t0, const uint4 (&v)[GROUPS][NP1]) {
#pragma unroll
                for (int g = 0; g < GROUPS; ++g) {
                    const int64_t site = t0 + 4 * g + rsel;
                    uint32_t acc = 0u;
#pragma unroll
                    for (int p = 0; p < NP1; ++p)
                        acc |= (v[g][p].x & umask[p][0]) | (v[g][p].y & umask[p][1]) | (v[g][p].z & umask[p][2]) |
                               (v[g][p].w & umask[p][3]);
#define PG_DPP_ROW_OR(ctrl) acc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, ctrl, 0xf, 0xf, false)
                    PG_DPP_ROW_OR(0x111);        // row_shr:1
                    PG_DPP_ROW_OR(0x112);        // row_shr:2
                    PG_DPP_ROW_OR(0x114);        // row_shr:4
                    PG_DPP_ROW_OR(0x118);        // row_shr:8 -> lane 15 of each 16-lane row holds the site's OR
#undef PG_DPP_ROW_OR
                    acc |= acc >> 16;
                    acc |= acc >> 8;
                    const bool is_cand = sub == 15 && site < c1 && __popc(acc & 0xFu) == 2;
                    const unsigned long long bal = __ballot(is_cand);
                    if (is_cand) my_cand[(ctail + __popcll(bal & ((1ull << lane) - 1ull))) & (PG_ABBA_CAND - 1)] = site;
                    ctail += (int)__popcll(bal);
                }
                while (ctail - chead >= 16) {
                    count_sites(my_cand[(chead + (lane >> 2)) & (PG_ABBA_CAND - 1)], true);
                    chead += 16;
                }
            };
            uint4 va[GROUPS][NP1], vb[GROUPS][NP1];
            int64_t t0 = c0 + SPW * wave;
            sl.issue(t0, va);
            while (t0 < c1) {
                sl.issue(t0 + 4 * SPW, vb);
                process(t0, va);
                t0 += 4 * SPW;
                if (t0 >= c1) break;
                sl.issue(t0 + 4 * SPW, va);
                process(t0, vb);
                t0 += 4 * SPW;
            }
        } else {
            for (int64_t t0 = c0 + 16 * wave; t0 < c1; t0 += 64) {
                const int64_t site = t0 + (lane >> 2);
                uint32_t pres = site < c1 ? range_presence_x4(gt + site * (int64_t)S, my_s, my_e) : 0u;
                pres |= (uint32_t)__shfl_xor((int)pres, 1, 64);
                pres |= (uint32_t)__shfl_xor((int)pres, 2, 64);
                const bool is_cand = q == 0 && __popc(pres) == 2;
                const unsigned long long bal = __ballot(is_cand);
                if (is_cand) my_cand[(ctail + __popcll(bal & ((1ull << lane) - 1ull))) & (PG_ABBA_CAND - 1)] = site;
                ctail += (int)__popcll(bal);
                while (ctail - chead >= 16) {
                    count_sites(my_cand[(chead + (lane >> 2)) & (PG_ABBA_CAND - 1)], true);
                    chead += 16;
                }
            }
        }
        if (ctail > chead) {
            const bool valid = (lane >> 2) < ctail - chead;
            count_sites(valid ? my_cand[(chead + (lane >> 2)) & (PG_ABBA_CAND - 1)] : c0, valid);
        }
        if (lane < tail - head) {
            uint32_t f[8];
            const uint32_t *e = my_ring[(head + lane) & (PG_ABBA_RING - 1)];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = e[k];
            quartet_terms<NSUM>(f, A);
        }
    }
    const size_t o = (size_t)win * max_chunks + chunk;
    double tot[NSUM + 2];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) tot[k] = A.acc[k];
    tot[NSUM] = (double)A.used;                          // <= PG_ABBA_SITES_PER_BLOCK, exact in a double
    tot[NSUM + 1] = (double)A.good;
    block_sum_multi<NSUM + 2>(tot, shd);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NSUM; ++k) part_sums[o * NSUM + k] = tot[k];
        part_used[o] = (int64_t)tot[NSUM] | ((int64_t)tot[NSUM + 1] << 32);      // used sites | good sites of the chunk (both < 2^31)
    }
}

__global__ void k_abba_reduce(const double *__restrict__ part_sums, const int64_t *__restrict__ part_used, int n_win,
                              int max_chunks, int nsum, const int64_t *__restrict__ win_lo,
                              const int64_t *__restrict__ win_hi, double *__restrict__ sums_out,
                              int64_t *__restrict__ used_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int win = idx / (nsum + 1), k = idx % (nsum + 1);
    if (win >= n_win) return;
    const int64_t len = win_hi[win] - win_lo[win];
    const int nch = (int)((len + PG_ABBA_SITES_PER_BLOCK - 1) / PG_ABBA_SITES_PER_BLOCK);
    if (k < nsum) {
        double s = 0.0;
        for (int c = 0; c < nch; ++c) s += part_sums[((size_t)win * max_chunks + c) * nsum + k];
        sums_out[(size_t)win * nsum + k] = s;
    } else {
        int64_t u = 0, g = 0;
        for (int c = 0; c < nch; ++c) {
            const int64_t pu = part_used[(size_t)win * max_chunks + c];
            u += pu & 0xffffffffll;
            g += pu >> 32;
        }
        // ABBABABA (six sums): a window without a single good site has sitesUsed = nan in the reference, told here as -1
        used_out[win] = (nsum == PG_ABBA_NSUM && g == 0) ? -1 : u;
    }
}

// ------------------------------------------------------------------------------------------------------
// k_quartet_np: the sums of ABBABABA / fourPop in NumPy's order.  The reference takes `.sum()` of arrays over the window's used
// sites (genomics.py:1431-1475, 1565-1569; 1420-1563): pairwise summation over pieces of 8192 values (see k_popdist_np).  One wave per
// window walks the flag bits k_abba_q left (the used sites), a site per lane: the chosen allele's counts are taken again (the site
// filters were applied when the flag was raised), the terms of up to 128 sites are staged in LDS and added up as a run -- eight lanes
// per sum --, and the tree above the runs is followed with a small stack (its depth is at most 7 + the chain of pieces).
// ------------------------------------------------------------------------------------------------------
template <int NSUM>
__global__ __launch_bounds__(64) void k_quartet_np(const int8_t *__restrict__ gt, int S, const int64_t *__restrict__ win_lo,
                                                   const int64_t *__restrict__ win_hi, const int32_t *__restrict__ pop_start, int q1,
                                                   int q2, int q3, int q4, int sel, const uint32_t *__restrict__ flags, int64_t base,
                                                   double *__restrict__ sums_out, long long max_sites) {
    __shared__ uint16_t list[8192];
    __shared__ double vals[NSUM][128];
    __shared__ double runv[NSUM];
    __shared__ double leftv[12][NSUM];
    const int win = blockIdx.x, lane = threadIdx.x;
    const int64_t lo = win_lo[win], hi = win_hi[win];
    if (hi - lo > max_sites) return;                                     // a long window keeps k_abba_reduce's sums
    const int qs[4] = {q1, q2, q3, q4};
    int ps[4], pe[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ps[k] = pop_start[qs[k]]; pe[k] = pop_start[qs[k] + 1]; }
    const uint4 *f4 = reinterpret_cast<const uint4 *>(flags);
    const int64_t q_first = hi > lo ? (lo - base) >> 7 : 0, q_last = hi > lo ? (hi - 1 - base) >> 7 : -1;
    auto masked = [&](int64_t q, uint32_t (&bits)[4]) {                  // the flag words of sites [128 q, 128 q + 128) inside the window
        bits[0] = bits[1] = bits[2] = bits[3] = 0u;
        if (q > q_last) return;
        const uint4 v = f4[q];
        bits[0] = v.x; bits[1] = v.y; bits[2] = v.z; bits[3] = v.w;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t s0 = base + (q << 7) + 32 * k;
            if (s0 + 32 <= lo || s0 >= hi) bits[k] = 0u;
            else {
                if (s0 < lo) bits[k] &= ~0u << (int)(lo - s0);
                if (hi - s0 < 32) bits[k] &= (1u << (int)(hi - s0)) - 1u;
            }
        }
    };
    // n = the used sites of the window
    long long n = 0;
    for (int64_t q0 = q_first; q0 <= q_last; q0 += 64) {
        uint32_t bits[4];
        masked(q0 + lane, bits);
        n += __popc(bits[0]) + __popc(bits[1]) + __popc(bits[2]) + __popc(bits[3]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) n += __shfl_xor(n, m, 64);
    // the stream of used sites: list[list_pos .. list_len) of the step that starts at flag group q_next - 64
    int64_t q_next = q_first, step_site0 = 0;
    int list_len = 0, list_pos = 0;
    auto refill = [&]() {
        while (list_pos == list_len && q_next <= q_last) {
            uint32_t bits[4];
            masked(q_next + lane, bits);
            const int cnt = __popc(bits[0]) + __popc(bits[1]) + __popc(bits[2]) + __popc(bits[3]);
            int pre = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(pre, d, 64);
                if (lane >= d) pre += v;
            }
            const int total = __shfl(pre, 63, 64);
            int k = pre - cnt;
            __syncthreads();                                             // the previous step's list has been consumed
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                uint32_t b = bits[x];
                while (b) {
                    list[k++] = (uint16_t)(lane * 128 + 32 * x + (__ffs((int)b) - 1));
                    b &= b - 1u;
                }
            }
            __syncthreads();
            step_site0 = base + (q_next << 7);
            q_next += 64;
            list_len = total;
            list_pos = 0;
        }
    };
    // the terms of the next `len` used sites -> vals[s][0 .. len)
    auto fill_run = [&](int len) {
        int filled = 0;
        while (filled < len) {
            refill();
            int take = len - filled < list_len - list_pos ? len - filled : list_len - list_pos;
            if (take > 64) take = 64;
            if (lane < take) {
                const int8_t *rowb = gt + (step_site0 + list[list_pos + lane]) * (int64_t)S;
                uint32_t cnt[4][4], nk[4], tot[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    range_counts_x4(rowb, ps[k], pe[k], cnt[k]);
                    nk[k] = cnt[k][0] + cnt[k][1] + cnt[k][2] + cnt[k][3];
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) tot[b] = cnt[0][b] + cnt[1][b] + cnt[2][b] + cnt[3][b];
                int der = -1;                                            // the allele choice of k_abba_q (see there)
                if (sel == PG_SEL_MINOR) {
                    int lo_b = -1, hi_b = -1;
#pragma unroll
                    for (int b = 3; b >= 0; --b)
                        if (tot[b] > 0) { if (hi_b < 0) hi_b = b; else lo_b = b; }
                    uint32_t tl = 0u, th = 0u;
#pragma unroll
                    for (int b = 0; b < 4; ++b) { if (b == lo_b) tl = tot[b]; if (b == hi_b) th = tot[b]; }
                    if (tl < th) der = lo_b;
                    else if (th < tl) der = hi_b;
                    else der = (lo_b == 0 && hi_b == 1) ? 1 : lo_b;
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (tot[b] > 0 && cnt[3][b] == 0) der = b;
                }
                uint32_t e[8] = {0u, 0u, 0u, nk[0], nk[1], nk[2], nk[3], 0u};
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (b == der) { e[0] = cnt[0][b]; e[1] = cnt[1][b]; e[2] = cnt[2][b]; e[7] = cnt[3][b]; }
                double t[NSUM];
                quartet_values<NSUM>(e, t);
#pragma unroll
                for (int s = 0; s < NSUM; ++s) vals[s][filled + lane] = t[s];
            }
            filled += take;
            list_pos += take;
        }
        __syncthreads();
    };
    // a run of len <= 128 values per sum, as NumPy adds it up -> runv[s]
    auto run_sums = [&](int len) {
        for (int s0 = 0; s0 < NSUM; s0 += 8) {
            const int s = s0 + (lane >> 3), j = lane & 7;
            if (s < NSUM) {
                double res;
                if (len < 8) {
                    res = 0.0;
                    for (int i = 0; i < len; ++i) res = res + vals[s][i];
                } else {
                    const int m8 = len - (len & 7);
                    double rj = vals[s][j];
                    for (int i = 8; i < m8; i += 8) rj = rj + vals[s][i + j];
                    rj = rj + __shfl_xor(rj, 1, 64);
                    rj = rj + __shfl_xor(rj, 2, 64);
                    rj = rj + __shfl_xor(rj, 4, 64);
                    res = rj;
                    for (int i = m8; i < len; ++i) res = res + vals[s][i];
                }
                if (j == 0) runv[s] = res;
            }
        }
        __syncthreads();
    };
    double total = 0.0;                                                  // lane s: sum s
    for (long long done = 0; done < n; done += 8192) {
        const int piece = n - done < 8192 ? (int)(n - done) : 8192;
        // pairwise sum of `piece` values off the stream: the recursion with an explicit stack (wave-uniform control)
        int size[12], stage[12], sp = 0;
        size[0] = piece; stage[0] = 0; sp = 1;
        double val = 0.0;
        while (sp > 0) {
            const int m = size[sp - 1];
            bool have = false;
            if (m <= 128) {
                fill_run(m);
                run_sums(m);
                val = lane < NSUM ? runv[lane] : 0.0;
                __syncthreads();
                --sp;
                have = true;
            } else {
                int n2 = m / 2;
                n2 -= n2 % 8;
                stage[sp - 1] = 1;
                size[sp] = n2; stage[sp] = 0; ++sp;
            }
            while (have && sp > 0) {                                     // hand a finished value to its parent
                const int pm = size[sp - 1];
                int n2 = pm / 2;
                n2 -= n2 % 8;
                if (stage[sp - 1] == 1) {                                // the left half: keep it, go right
                    if (lane < NSUM) leftv[sp - 1][lane] = val;
                    stage[sp - 1] = 2;
                    size[sp] = pm - n2; stage[sp] = 0; ++sp;
                    have = false;
                } else {                                                 // the right half: the parent is done
                    if (lane < NSUM) val = leftv[sp - 1][lane] + val;
                    --sp;
                }
            }
        }
        total = total + val;
    }
    if (lane < NSUM) sums_out[(size_t)win * NSUM + lane] = total;
}

void pg_launch_abba(hipStream_t st, const int8_t *gt, int S, const int64_t *win_lo, const int64_t *win_hi,
                    int n_win, int max_chunks, const int32_t *pop_start, int p1, int p2, int p3, int p4,
                    double min_data, int sel, int nsum, double *part_sums, int64_t *part_used, double *sums_out,
                    int64_t *used_out, uint32_t *flags, int64_t base, long long max_sites) {
    if (n_win <= 0) return;
    if (max_chunks > 0) {
        const dim3 grid(max_chunks, n_win);
#define PG_ABBA_LAUNCH(NS, NP)                                                                                          \
    hipLaunchKernelGGL((k_abba_q<NS, NP>), grid, dim3(256), 0, st, gt, S, win_lo, win_hi, max_chunks, pop_start, p1, p2, \
                       p3, p4, min_data, sel, part_sums, part_used, flags, base)
        const int npass = (S + 255) / 256;
        if (nsum == PG_ABBA_NSUM) {
            if (npass == 1) PG_ABBA_LAUNCH(PG_ABBA_NSUM, 1);
            else if (npass == 2) PG_ABBA_LAUNCH(PG_ABBA_NSUM, 2);
            else if (npass <= 4) PG_ABBA_LAUNCH(PG_ABBA_NSUM, 4);
            else PG_ABBA_LAUNCH(PG_ABBA_NSUM, 0);
        } else {
            if (npass == 1) PG_ABBA_LAUNCH(PG_FOURPOP_NSUM, 1);
            else if (npass == 2) PG_ABBA_LAUNCH(PG_FOURPOP_NSUM, 2);
            else if (npass <= 4) PG_ABBA_LAUNCH(PG_FOURPOP_NSUM, 4);
            else PG_ABBA_LAUNCH(PG_FOURPOP_NSUM, 0);
        }
#undef PG_ABBA_LAUNCH
    }
    int total = n_win * (nsum + 1);
    hipLaunchKernelGGL(k_abba_reduce, dim3((total + 255) / 256), dim3(256), 0, st, part_sums, part_used, n_win,
                       max_chunks, nsum, win_lo, win_hi, sums_out, used_out);
    if (flags) {                                                         // the sums again, in NumPy's order (the counts stay)
        if (nsum == PG_ABBA_NSUM)
            hipLaunchKernelGGL((k_quartet_np<PG_ABBA_NSUM>), dim3(n_win), dim3(64), 0, st, gt, S, win_lo, win_hi, pop_start, p1, p2, p3,
                               p4, sel, flags, base, sums_out, max_sites);
        else
            hipLaunchKernelGGL((k_quartet_np<PG_FOURPOP_NSUM>), dim3(n_win), dim3(64), 0, st, gt, S, win_lo, win_hi, pop_start, p1, p2,
                               p3, p4, sel, flags, base, sums_out, max_sites);
    }
}

// ------------------------------------------------------------------------------------------------------
// K_popfreq: exact integers, so the cross-block combination uses integer atomics (order independent).
// ------------------------------------------------------------------------------------------------------
struct FreqAcc {
    unsigned long long l, Sx[PG_MAX_POPS], Px[PG_MAX_POPS];
};

__device__ __forceinline__ void popfreq_site(const uint32_t *__restrict__ row, int n_hap, const int32_t *__restrict__ pop_start,
                                             int n_pops, FreqAcc &F, uint32_t *__restrict__ flags, int64_t rel) {
    uint32_t call[4];
    range_counts(row, 0, n_hap, call);
    if ((int)(call[0] + call[1] + call[2] + call[3]) != n_hap) return;       // genomics.py:1010
    ++F.l;
    bool poly = false;
#pragma unroll
    for (int q = 0; q < PG_MAX_POPS; ++q) {
        if (q < n_pops) {
            uint32_t c[4];
            range_counts(row, pop_start[q], pop_start[q + 1], c);
            const unsigned long long pr = (unsigned long long)c[0] * c[1] + (unsigned long long)c[0] * c[2] +
                                          (unsigned long long)c[0] * c[3] + (unsigned long long)c[1] * c[2] +
                                          (unsigned long long)c[1] * c[3] + (unsigned long long)c[2] * c[3];
            F.Px[q] += pr;
            F.Sx[q] += (pr != 0ull);
            poly = poly || pr != 0ull;
        }
    }
    if (poly && flags) flag_site(flags, rel);
}

// Rows longer than 1024 bytes (more than 1024 haplotype slots): one thread per site straight from global memory.
__global__ __launch_bounds__(256) void k_popfreq(const int8_t *__restrict__ gt, int S, int n_hap,
                                                 const int64_t *__restrict__ win_lo, const int64_t *__restrict__ win_hi,
                                                 const int32_t *__restrict__ pop_start, int n_pops,
                                                 unsigned long long *__restrict__ l_out,
                                                 unsigned long long *__restrict__ S_out,
                                                 unsigned long long *__restrict__ pairsum_out, uint32_t *__restrict__ flags,
                                                 int64_t base) {
    __shared__ unsigned long long shu[256];
    const int win = blockIdx.y, chunk = blockIdx.x;
    const int64_t lo = win_lo[win], hi = win_hi[win];
    const int64_t c0 = lo + (int64_t)chunk * PG_SITES_PER_BLOCK;
    if (c0 >= hi) return;
    FreqAcc F;
    F.l = 0;
#pragma unroll
    for (int q = 0; q < PG_MAX_POPS; ++q) { F.Sx[q] = 0; F.Px[q] = 0; }
    const int64_t c1 = (c0 + PG_SITES_PER_BLOCK < hi) ? c0 + PG_SITES_PER_BLOCK : hi;
    for (int64_t t = c0 + threadIdx.x; t < c1; t += blockDim.x)
        popfreq_site(reinterpret_cast<const uint32_t *>(gt + t * (int64_t)S), n_hap, pop_start, n_pops, F, flags, t - base);
    __syncthreads();
    unsigned long long r = block_sum_u64(F.l, shu);
    if (threadIdx.x == 0 && r) atomicAdd(&l_out[win], r);
    for (int q = 0; q < n_pops; ++q) {
        r = block_sum_u64(F.Sx[q], shu);
        if (threadIdx.x == 0 && r) atomicAdd(&S_out[(size_t)win * n_pops + q], r);
        r = block_sum_u64(F.Px[q], shu);
        if (threadIdx.x == 0 && r) atomicAdd(&pairsum_out[(size_t)win * n_pops + q], r);
    }
}

// k_popfreq_q: same structure as k_abba_q.  Screening pass: a site takes part iff every haplotype slot is called
// (genomics.py:1010); called codes are one-hot and pad bytes zero, so that is popcount(row) == n_hap -- four accumulating
// v_bcnt per 16-byte piece on fully coalesced loads, summed over the 16 lanes of a row with DPP.  Counting pass, 16 queued
// sites at a time: lane q of a site's quad counts populations q, q+4, q+8, q+12.
template <int NPASS>
__global__ __launch_bounds__(256) void k_popfreq_q(const int8_t *__restrict__ gt, int S, int n_hap,
                                                   const int64_t *__restrict__ win_lo, const int64_t *__restrict__ win_hi,
                                                   const int32_t *__restrict__ pop_start, int n_pops,
                                                   unsigned long long *__restrict__ l_out,
                                                   unsigned long long *__restrict__ S_out,
                                                   unsigned long long *__restrict__ pairsum_out, uint32_t *__restrict__ flags,
                                                   int64_t base) {
    __shared__ int64_t cand[4][PG_ABBA_CAND];
    __shared__ unsigned long long shu[4][4][9];
    // per-lane accumulators live in LDS (slot k of lane l at [k][l]): the counting pass is the rare path on typical data, and
    // keeping its state out of the register file lets the screening loop hold more row loads in flight
    __shared__ unsigned long long lacc[4][9][64];
    const int win = blockIdx.y, chunk = blockIdx.x;
    const int64_t lo = win_lo[win], hi = win_hi[win];
    const int64_t c0 = lo + (int64_t)chunk * PG_ABBA_SITES_PER_BLOCK;
    if (c0 >= hi) return;
    const int64_t c1 = (c0 + PG_ABBA_SITES_PER_BLOCK < hi) ? c0 + PG_ABBA_SITES_PER_BLOCK : hi;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 3, sub = lane & 15, rsel = lane >> 4;
    int64_t *my_cand = cand[wave];
    int chead = 0, ctail = 0;
    unsigned long long (*my_acc)[64] = lacc[wave];
#pragma unroll
    for (int z = 0; z < 9; ++z) my_acc[z][lane] = 0ull;          // l, Sx[0..3], Px[0..3] for populations q, q+4, q+8, q+12
    auto count_sites = [&](int64_t site, bool valid) {
        if (!valid) return;
        const int8_t *rowb = gt + site * (int64_t)S;
        if (q == 0) my_acc[0][lane] += 1ull;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int p = q + 4 * j;
            if (p < n_pops) {
                uint32_t c[4];
                range_counts_x4(rowb, pop_start[p], pop_start[p + 1], c);
                const unsigned long long pr = (unsigned long long)c[0] * c[1] + (unsigned long long)c[0] * c[2] +
                                              (unsigned long long)c[0] * c[3] + (unsigned long long)c[1] * c[2] +
                                              (unsigned long long)c[1] * c[3] + (unsigned long long)c[2] * c[3];
                my_acc[1 + j][lane] += (pr != 0ull);
                my_acc[5 + j][lane] += pr;
                if (pr != 0ull && flags) flag_site(flags, site - base);
            }
        }
    };
    // Screening loads: raw buffer loads on a descriptor of the block's rows (rows past the window and the lanes past the end
    // of a row read as zero: no predication, no address arithmetic beyond one add per step), and the loads of step k+1 are
    // issued before step k is processed, so a wave always has GROUPS*NPASS .. 2*GROUPS*NPASS 16-byte loads in flight.
    constexpr int GROUPS = 8 / NPASS, SPW = 4 * GROUPS;
    const ScreenLoads<GROUPS, NPASS> sl(gt, S, c0, c1, sub, rsel);
    auto process = [&](int64_t t0, const uint4 (&v)[GROUPS][NPASS]) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int64_t site = t0 + 4 * g + rsel;
            uint32_t cnt = 0u;
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
                cnt += __popc(v[g][p].x) + __popc(v[g][p].y) + __popc(v[g][p].z) + __popc(v[g][p].w);
#define PG_DPP_ROW_ADD(ctrl) cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, ctrl, 0xf, 0xf, false)
            PG_DPP_ROW_ADD(0x111);        // row_shr:1
            PG_DPP_ROW_ADD(0x112);        // row_shr:2
            PG_DPP_ROW_ADD(0x114);        // row_shr:4
            PG_DPP_ROW_ADD(0x118);        // row_shr:8 -> lane 15 of each 16-lane row holds the row's popcount
#undef PG_DPP_ROW_ADD
            const bool is_cand = sub == 15 && site < c1 && (int)cnt == n_hap;
            const unsigned long long bal = __ballot(is_cand);
            if (is_cand) my_cand[(ctail + __popcll(bal & ((1ull << lane) - 1ull))) & (PG_ABBA_CAND - 1)] = site;
            ctail += (int)__popcll(bal);
        }
        while (ctail - chead >= 16) {
            count_sites(my_cand[(chead + (lane >> 2)) & (PG_ABBA_CAND - 1)], true);
            chead += 16;
        }
    };
    {
        uint4 va[GROUPS][NPASS], vb[GROUPS][NPASS];
        int64_t t0 = c0 + SPW * wave;
        sl.issue(t0, va);
        while (t0 < c1) {
            sl.issue(t0 + 4 * SPW, vb);
            process(t0, va);
            t0 += 4 * SPW;
            if (t0 >= c1) break;
            sl.issue(t0 + 4 * SPW, va);
            process(t0, vb);
            t0 += 4 * SPW;
        }
    }
    if (ctail > chead) {
        const bool valid = (lane >> 2) < ctail - chead;
        count_sites(valid ? my_cand[(chead + (lane >> 2)) & (PG_ABBA_CAND - 1)] : c0, valid);
    }
    // lanes with equal q: butterfly over lane bits 2..5, then the four waves through LDS; exact integers, any order
    unsigned long long acc[9];
#pragma unroll
    for (int z = 0; z < 9; ++z) acc[z] = my_acc[z][lane];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int m = 32; m >= 4; m >>= 1) acc[k] += __shfl_xor(acc[k], m, 64);
    }
    if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 9; ++k) shu[wave][lane][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long t[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) t[k] = shu[0][q][k] + shu[1][q][k] + shu[2][q][k] + shu[3][q][k];
        if (q == 0 && t[0]) atomicAdd(&l_out[win], t[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = q + 4 * j;
            if (p < n_pops) {
                if (t[1 + j]) atomicAdd(&S_out[(size_t)win * n_pops + p], t[1 + j]);
                if (t[5 + j]) atomicAdd(&pairsum_out[(size_t)win * n_pops + p], t[5 + j]);
            }
        }
    }
}

// k_popfreq_ordered: thetaPi as the reference forms it.  Alignment.groupFreqStats takes Python's sum() over the window's
// per-site values `pairs / (.5*N*(N-1))` (genomics.py:1016-1018, 609-616): a SEQUENTIAL float64 sum in site order, whose
// roundings no reduction tree reproduces -- and for a population of three haplotypes Tajima's D is that rounding noise over a
// variance of exactly zero (+-inf or nan in the reference's output).  One wave per window walks the flags the counting kernels
// left (1 bit per site: every slot called and some population polymorphic; all other sites add 0.0, which changes nothing),
// 2048 sites per step: the flagged sites are listed in site order, their per-population values are formed 64 sites at a time
// (one site per lane), and lane p adds the values of population p one after the other.
__global__ __launch_bounds__(64) void k_popfreq_ordered(const int8_t *__restrict__ gt, int S, const int64_t *__restrict__ win_lo,
                                                        const int64_t *__restrict__ win_hi, const int32_t *__restrict__ pop_start,
                                                        int n_pops, const uint32_t *__restrict__ flags, int64_t base,
                                                        double *__restrict__ theta_out) {
    __shared__ uint16_t list[8192];
    __shared__ double tile[PG_MAX_POPS][65];
    const int win = blockIdx.x, lane = threadIdx.x;
    const int64_t lo = win_lo[win], hi = win_hi[win];
    double acc = 0.0;
    if (hi > lo) {
        // four flag words (128 sites) per lane and step: 8192 sites per step (`base` is a multiple of 128 sites, the flag buffer
        // ends in slack words)
        const int64_t q_first = (lo - base) >> 7, q_last = (hi - 1 - base) >> 7;
        const uint4 *f4 = reinterpret_cast<const uint4 *>(flags);
        const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
        uint4 vn[4];                                                         // the flag words of the next four steps, in flight
#pragma unroll
        for (int a = 0; a < 4; ++a) vn[a] = q_first + 64 * a + lane <= q_last ? f4[q_first + 64 * a + lane] : zero4;
        for (int64_t q0 = q_first; q0 <= q_last; q0 += 64) {
            const int64_t q = q0 + lane;
            uint32_t bits[4] = {0u, 0u, 0u, 0u};
            const uint4 v = vn[0];
            vn[0] = vn[1]; vn[1] = vn[2]; vn[2] = vn[3];
            vn[3] = q + 256 <= q_last ? f4[q + 256] : zero4;
            if (q <= q_last) {
                bits[0] = v.x; bits[1] = v.y; bits[2] = v.z; bits[3] = v.w;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t s0 = base + (q << 7) + 32 * k;             // the site of bit 0 of word k
                    if (s0 + 32 <= lo || s0 >= hi) bits[k] = 0u;
                    else {
                        if (s0 < lo) bits[k] &= ~0u << (int)(lo - s0);
                        if (hi - s0 < 32) bits[k] &= (1u << (int)(hi - s0)) - 1u;
                    }
                }
            }
            const int cnt = __popc(bits[0]) + __popc(bits[1]) + __popc(bits[2]) + __popc(bits[3]);
            if (__ballot(cnt != 0) == 0ull) continue;                        // (wave-uniform) nothing flagged among these 8192 sites
            int pre = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(pre, d, 64);
                if (lane >= d) pre += v;
            }
            const int total = __shfl(pre, 63, 64);
            int k = pre - cnt;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                uint32_t b = bits[x];
                while (b) {
                    list[k++] = (uint16_t)(lane * 128 + 32 * x + (__ffs((int)b) - 1));
                    b &= b - 1u;
                }
            }
            __syncthreads();
            for (int b0 = 0; b0 < total; b0 += 64) {
                const int nb = total - b0 < 64 ? total - b0 : 64;
                if (lane < nb) {
                    const int8_t *rowb = gt + (base + (q0 << 7) + list[b0 + lane]) * (int64_t)S;
                    for (int p = 0; p < n_pops; ++p) {
                        const int ps = pop_start[p], pe = pop_start[p + 1];
                        uint32_t c[4];
                        range_counts_x4(rowb, ps, pe, c);
                        const unsigned long long pr = (unsigned long long)c[0] * c[1] + (unsigned long long)c[0] * c[2] +
                                                      (unsigned long long)c[0] * c[3] + (unsigned long long)c[1] * c[2] +
                                                      (unsigned long long)c[1] * c[3] + (unsigned long long)c[2] * c[3];
                        const double N = (double)(pe - ps);
                        tile[p][lane] = (double)pr / (.5 * N * (N - 1.0));       // baseCountPi, genomics.py:609-616
                    }
                }
                __syncthreads();
                if (lane < n_pops)
                    for (int j = 0; j < nb; ++j) acc = acc + tile[lane][j];
                __syncthreads();
            }
        }
    }
    if (lane < n_pops) theta_out[(size_t)win * n_pops + lane] = acc;
}

void pg_launch_popfreq(hipStream_t st, const int8_t *gt, int S, int n_hap, const int64_t *win_lo,
                       const int64_t *win_hi, int n_win, int max_chunks, const int32_t *pop_start, int n_pops,
                       unsigned long long *l_out, unsigned long long *S_out, unsigned long long *pairsum_out,
                       uint32_t *flags, int64_t base) {
    if (n_win <= 0 || max_chunks <= 0) return;
    const int npass = (S + 255) / 256;
    if (npass <= 4) {                 // rows up to 1024 bytes: screening + counting kernel, 4096-site blocks
        const int chunks_q = (int)(((int64_t)max_chunks * PG_SITES_PER_BLOCK + PG_ABBA_SITES_PER_BLOCK - 1) / PG_ABBA_SITES_PER_BLOCK);
        const dim3 grid(chunks_q, n_win);
#define PG_POPFREQ_LAUNCH(NP)                                                                                           \
    hipLaunchKernelGGL((k_popfreq_q<NP>), grid, dim3(256), 0, st, gt, S, n_hap, win_lo, win_hi, pop_start, n_pops, l_out, \
                       S_out, pairsum_out, flags, base)
        if (npass == 1) PG_POPFREQ_LAUNCH(1);
        else if (npass == 2) PG_POPFREQ_LAUNCH(2);
        else PG_POPFREQ_LAUNCH(4);
#undef PG_POPFREQ_LAUNCH
    } else {
        hipLaunchKernelGGL(k_popfreq, dim3(max_chunks, n_win), dim3(256), 0, st, gt, S, n_hap, win_lo, win_hi, pop_start, n_pops,
                           l_out, S_out, pairsum_out, flags, base);
    }
}

void pg_launch_popfreq_ordered(hipStream_t st, const int8_t *gt, int S, const int64_t *win_lo, const int64_t *win_hi, int n_win,
                               const int32_t *pop_start, int n_pops, const uint32_t *flags, int64_t base, double *theta_out) {
    if (n_win <= 0) return;
    hipLaunchKernelGGL(k_popfreq_ordered, dim3(n_win), dim3(64), 0, st, gt, S, win_lo, win_hi, pop_start, n_pops, flags, base,
                       theta_out);
}

// ------------------------------------------------------------------------------------------------------
// K_counts: cnt[site][pop][4]
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_site_counts(const int8_t *__restrict__ gt, int S, int64_t site_lo,
                                                     int64_t site_hi, const int32_t *__restrict__ pop_start, int n_pops,
                                                     int32_t *__restrict__ cnt_out) {
    const int64_t site = site_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (site >= site_hi) return;
    const uint32_t *row = reinterpret_cast<const uint32_t *>(gt + site * (int64_t)S);
    for (int q = 0; q < n_pops; ++q) {
        uint32_t c[4];
        range_counts(row, pop_start[q], pop_start[q + 1], c);
        int4 o = make_int4((int)c[0], (int)c[1], (int)c[2], (int)c[3]);
        *reinterpret_cast<int4 *>(cnt_out + ((size_t)(site - site_lo) * n_pops + q) * 4) = o;
    }
}

void pg_launch_site_counts(hipStream_t st, const int8_t *gt, int S, int64_t site_lo, int64_t site_hi,
                           const int32_t *pop_start, int n_pops, int32_t *cnt_out) {
    int64_t n = site_hi - site_lo;
    if (n <= 0 || n_pops <= 0) return;
    hipLaunchKernelGGL(k_site_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gt, S, site_lo, site_hi,
                       pop_start, n_pops, cnt_out);
}

// ------------------------------------------------------------------------------------------------------
// K_target: freq.py's per-site finaliser over cnt[site][pop][4] (freq.py:60-105; derivedAllele / minorAllele genomics.py:636-669):
// the target allele of the site (derived: the one ingroup allele the last population does not carry, when exactly one allele is
// seen there, two among the others, one shared; minor: the rarer of exactly two), then per population its count or frequency
// np.around(., 4) = rint(x * 1e4) / 1e4, the --threshold rule, and the row's keep flag.  A thread per site; float64 throughout.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_site_target(const int32_t *__restrict__ cnt, int64_t n_sites, int n_pops, int target,
                                                     double min_data, int as_counts, int has_threshold, double threshold,
                                                     double *__restrict__ f_out, long long *__restrict__ i_out,
                                                     uint8_t *__restrict__ keep_out) {
    const int64_t site = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (site >= n_sites) return;
    const int4 *c4 = reinterpret_cast<const int4 *>(cnt + (size_t)site * n_pops * 4);
    bool ok;
    int base = 0;
    if (target == 1) {
        const int4 o = c4[n_pops - 1];
        long long in[4] = {0, 0, 0, 0};
        for (int q = 0; q < n_pops - 1; ++q) {
            const int4 v = c4[q];
            in[0] += v.x; in[1] += v.y; in[2] += v.z; in[3] += v.w;
        }
        const bool outc[4] = {o.x > 0, o.y > 0, o.z > 0, o.w > 0};
        int n_out = 0, n_in = 0;
        bool shared = false, found = false;
        for (int b = 0; b < 4; ++b) {
            const bool inc = in[b] > 0;
            n_out += outc[b];
            n_in += inc;
            shared = shared || (outc[b] && inc);
            if (!found && inc && !outc[b]) { base = b; found = true; }       // np.argmax of booleans: the first True (0 without one)
        }
        ok = n_out == 1 && n_in == 2 && shared;
    } else {
        long long tot[4] = {0, 0, 0, 0};
        for (int q = 0; q < n_pops; ++q) {
            const int4 v = c4[q];
            tot[0] += v.x; tot[1] += v.y; tot[2] += v.z; tot[3] += v.w;
        }
        int seen = 0;
        long long best = 0x7fffffffffffffffll;
        for (int b = 0; b < 4; ++b) {
            seen += tot[b] > 0;
            const long long m = tot[b] > 0 ? tot[b] : 0x7fffffffffffffffll;
            if (m < best) { best = m; base = b; }                              // np.argmin: the first minimum
        }
        ok = seen == 2;
    }
    bool all_nan = true, all_zero = true;
    for (int q = 0; q < n_pops; ++q) {
        const int4 v = c4[q];
        const long long nq = (long long)v.x + v.y + v.z + v.w;
        const long long cb = base == 0 ? v.x : base == 1 ? v.y : base == 2 ? v.z : v.w;
        const bool good = ok && (double)nq >= min_data;                      // freq.py:80: the COUNT is compared with --minData
        if (as_counts) {
            const long long t = good ? cb : 0;
            i_out[(size_t)site * n_pops + q] = t;
            all_zero = all_zero && t == 0;
        } else {
            double t = __longlong_as_double(0x7ff8000000000000ll);
            if (good) t = rint((1.0 * (double)cb / (double)nq) * 10000.0) / 10000.0;
            if (has_threshold && t == t) t = t >= threshold ? 1.0 : 0.0;
            f_out[(size_t)site * n_pops + q] = t;
            all_nan = all_nan && t != t;
        }
    }
    keep_out[site] = as_counts ? !all_zero : !all_nan;
}

void pg_launch_site_target(hipStream_t st, const int32_t *cnt, int64_t n_sites, int n_pops, int target, double min_data, int as_counts,
                           int has_threshold, double threshold, double *f_out, long long *i_out, uint8_t *keep_out) {
    if (n_sites <= 0 || n_pops <= 0) return;
    hipLaunchKernelGGL(k_site_target, dim3((unsigned)((n_sites + 255) / 256)), dim3(256), 0, st, cnt, n_sites, n_pops, target, min_data,
                       as_counts, has_threshold, threshold, f_out, i_out, keep_out);
}

// ------------------------------------------------------------------------------------------------------
// K_called: per-window, per-haplotype number of called sites (Alignment.seqNonNan, genomics.py:1038-1040).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hap_called(const int8_t *__restrict__ gt, int S, int n_hap,
                                                    const int64_t *__restrict__ win_lo, const int64_t *__restrict__ win_hi,
                                                    unsigned long long *__restrict__ out) {
    const int win = blockIdx.y;
    const int64_t lo = win_lo[win], hi = win_hi[win];
    const int64_t c0 = lo + (int64_t)blockIdx.x * PG_SITES_PER_BLOCK;
    if (c0 >= hi) return;
    const int64_t c1 = (c0 + PG_SITES_PER_BLOCK < hi) ? c0 + PG_SITES_PER_BLOCK : hi;
    for (int g = threadIdx.x; 4 * g < n_hap; g += blockDim.x) {
        uint32_t tot[4] = {0u, 0u, 0u, 0u};
        uint32_t acc = 0u;
        int pending = 0;
        for (int64_t s = c0; s < c1; ++s) {
            const uint32_t d = *reinterpret_cast<const uint32_t *>(gt + s * (int64_t)S + 4 * g);
            acc += (d | (d >> 1) | (d >> 2) | (d >> 3)) & 0x01010101u;
            if (++pending == 255) {
#pragma unroll
                for (int k = 0; k < 4; ++k) tot[k] += (acc >> (8 * k)) & 0xFFu;
                acc = 0u;
                pending = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tot[k] += (acc >> (8 * k)) & 0xFFu;
            if (4 * g + k < n_hap && tot[k]) atomicAdd(&out[(size_t)win * n_hap + 4 * g + k], (unsigned long long)tot[k]);
        }
    }
}

void pg_launch_hap_called(hipStream_t st, const int8_t *gt, int S, int n_hap, const int64_t *win_lo,
                          const int64_t *win_hi, int n_win, int max_chunks, unsigned long long *out) {
    if (n_win <= 0 || max_chunks <= 0) return;
    hipLaunchKernelGGL(k_hap_called, dim3(max_chunks, n_win), dim3(256), 0, st, gt, S, n_hap, win_lo, win_hi, out);
}
