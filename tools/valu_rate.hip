// Micro-benchmark: sustained issue rate of the integer VALU ops the pair kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
// Every mode runs 16 independent accumulator chains per lane at 8 / 4 / 2 waves per SIMD (blocks of 256 threads on 256 CUs) and
// reports wave-instructions per second by wall clock (HIP events), and what that is in SIMD cycles per instruction if the
// clock were the nominal 2.4 GHz (the guide's figure is 2 cycles per wave64 instruction on a SIMD-32; the part clocks lower
// under a dense VALU stream, v_add_u32 / v_fma_f32 are the reference lines).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum {
    M_AND_BCNT_DEP,     // v_and(sgpr) -> v_bcnt(acc) back to back on the same temporary        (naive k_pairC body)
    M_AND_BCNT_ILV,     // 8 x v_and into 8 temporaries, then 8 x v_bcnt(acc)                     (k_pairC as scheduled)
    M_ANDV_BCNT_ILV,    // the same with the row operand in a VGPR instead of an SGPR
    M_AND_SGPR,         // v_and_b32 v, s, v
    M_AND_VGPR,         // v_and_b32 v, v, v
    M_BCNT_ACC,         // v_bcnt_u32_b32 acc, v, acc
    M_BCNT_ZERO,        // v_bcnt_u32_b32 d, v, 0 (no accumulate)
    M_XOR_BITOP_BCNT,   // v_xor(sgpr) + v_bitop3(and-and, sgpr) + v_bcnt(acc), interleaved by 8  (k_pairD body)
    M_BITOP3,           // v_bitop3_b32 alone
    M_AND_OR,           // v_and_or_b32
    M_ADD,              // v_add_u32 reference
    M_ADD3,             // v_add3_u32
    M_FMA,              // v_fma_f32 reference (the guide's 2-cycle instruction)
    N_MODES
};

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, unsigned long long *clk, int iters, uint32_t s0, uint32_t s1) {
    uint32_t a[16], t[8];
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9E3779B9u;
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = k;
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = k;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_AND_BCNT_ILV || MODE == M_ANDV_BCNT_ILV || MODE == M_XOR_BITOP_BCNT) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (MODE == M_AND_BCNT_ILV)
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t[k]) : "s"(s0 + k), "v"(x));
                    else if (MODE == M_ANDV_BCNT_ILV)
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t[k]) : "v"(y), "v"(x));
                    else
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[k]) : "s"(s0 + k), "v"(x));
                }
                if (MODE == M_XOR_BITOP_BCNT) {
#pragma unroll
                    for (int k = 0; k < 8; ++k)                     // t = t & s & y  (bitop3 0x80 = a & b & c)
                        asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(t[k]) : "s"(s1 + k), "v"(y));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[8 * h + k]) : "v"(t[k]));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (MODE == M_AND_BCNT_DEP) {
                    asm volatile("v_and_b32 %0, %1, %2" : "=v"(y) : "s"(s0 + k), "v"(x));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(y));
                } else if (MODE == M_AND_SGPR) {
                    asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[k]) : "s"(s0 + k), "v"(x));
                } else if (MODE == M_AND_VGPR) {
                    asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[k]) : "v"(y), "v"(x));
                } else if (MODE == M_BCNT_ACC) {
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
                } else if (MODE == M_BCNT_ZERO) {
                    asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(a[k]) : "v"(x));
                } else if (MODE == M_BITOP3) {
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(a[k]) : "s"(s1 + k), "v"(y));
                } else if (MODE == M_AND_OR) {
                    asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[k]) : "s"(s0 + k), "v"(x));
                } else if (MODE == M_ADD) {
                    asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
                } else if (MODE == M_ADD3) {
                    asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                } else {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += a[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) r += t[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + y;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int MODE>
void run(const char *name, int ops_per_iter, int waves_per_simd) {
    uint32_t *d;
    unsigned long long *clk, hclk[64];
    const int blocks = 256 * waves_per_simd, iters = 4000;
    hipMalloc(&d, blocks * 256 * 4);
    hipMalloc(&clk, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<MODE><<<blocks, 256>>>(d, clk, 10, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<MODE><<<blocks, 256>>>(d, clk, iters, 1, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, clk, sizeof(hclk), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < 64; ++i) cyc += (double)hclk[i] / 64;
    const double per_wave = (double)iters * ops_per_iter;
    const double winstr = (double)blocks * 4 * per_wave;
    // a wave's loop takes `cyc` shader cycles while waves_per_simd waves share its SIMD
    (void)cyc;
    printf("%-36s %d waves/SIMD %8.3f ms  %.3e wave-instr/s  = %.2f SIMD cycles/instr at 2.4 GHz\n", name, waves_per_simd, ms,
           winstr / (ms * 1e-3), 2.4e9 * 1024 / (winstr / (ms * 1e-3)));
    hipFree(d); hipFree(clk);
}

int main() {
    for (int wps : {8, 4, 2}) {
        if (wps == 8) {
            run<M_AND_BCNT_DEP>("v_and(s)->v_bcnt(acc) dependent", 32, 8);
            run<M_AND_BCNT_ILV>("8 v_and(s); 8 v_bcnt(acc)", 32, 8);
            run<M_ANDV_BCNT_ILV>("8 v_and(v); 8 v_bcnt(acc)", 32, 8);
            run<M_AND_SGPR>("v_and_b32 v,s,v", 16, 8);
            run<M_AND_VGPR>("v_and_b32 v,v,v", 16, 8);
            run<M_BCNT_ACC>("v_bcnt_u32_b32 acc", 16, 8);
            run<M_BCNT_ZERO>("v_bcnt_u32_b32 d,v,0", 16, 8);
            run<M_XOR_BITOP_BCNT>("8 v_xor(s); 8 v_bitop3(s); 8 v_bcnt", 48, 8);
            run<M_BITOP3>("v_bitop3_b32 v,s,v", 16, 8);
            run<M_AND_OR>("v_and_or_b32 v,s,v,v", 16, 8);
            run<M_ADD>("v_add_u32", 16, 8);
            run<M_ADD3>("v_add3_u32", 16, 8);
            run<M_FMA>("v_fma_f32", 16, 8);
        } else if (wps == 4) {
            run<M_AND_BCNT_DEP>("v_and(s)->v_bcnt(acc) dependent", 32, 4);
            run<M_AND_BCNT_ILV>("8 v_and(s); 8 v_bcnt(acc)", 32, 4);
            run<M_ANDV_BCNT_ILV>("8 v_and(v); 8 v_bcnt(acc)", 32, 4);
            run<M_XOR_BITOP_BCNT>("8 v_xor(s); 8 v_bitop3(s); 8 v_bcnt", 48, 4);
            run<M_ADD>("v_add_u32", 16, 4);
        } else {
            run<M_AND_BCNT_ILV>("8 v_and(s); 8 v_bcnt(acc)", 32, 2);
            run<M_XOR_BITOP_BCNT>("8 v_xor(s); 8 v_bitop3(s); 8 v_bcnt", 48, 2);
            run<M_ADD>("v_add_u32", 16, 2);
        }
    }
    return 0;
}
