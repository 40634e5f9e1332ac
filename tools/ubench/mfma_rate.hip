// Micro-benchmark behind HISTORY.md's pricing of the pair kernels against the matrix pipe of gfx950: cycles per instruction
// (s_memtime, the shader clock) and the clock itself (against the 100 MHz s_memrealtime) of
//   * back-to-back independent v_mfma_f32_32x32x64_f8f6f4 / v_mfma_f32_16x16x128_f8f6f4 on fp4 operands,
//   * the same with VALU operations between them that rewrite the fragment of the product after next (what k_pairC_big does),
//   * plain VALU streams (independent destinations; one destination written over and over),
// with one and with two waves per SIMD.  Every loop body is a single asm statement: the compiler's own scheduling of builtin
// calls (register copies, s_nop padding) would be what gets measured otherwise.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define M32(ACC, B) "v_mfma_f32_32x32x64_f8f6f4 " ACC ", v[52:55], v[" B "], " ACC " cbsz:4 blgp:4\n\t"
#define M16(ACC, B) "v_mfma_f32_16x16x128_f8f6f4 " ACC ", v[52:55], v[" B "], " ACC " cbsz:4 blgp:4\n\t"
#define V4(D0, D1, D2, D3) "v_and_b32 v" D0 ", %8, %9\n\tv_lshrrev_b32 v" D1 ", 1, %8\n\tv_and_b32 v" D2 ", %8, %9\n\tv_lshrrev_b32 v" D3 ", 2, %9\n\t"
#define V2(D0, D1) "v_and_b32 v" D0 ", %8, %9\n\tv_lshrrev_b32 v" D1 ", 1, %8\n\t"
#define NONE ""
// product i reads fragment F[i % 3] (v[40:43], v[44:47], v[48:51]); the VALU operations behind it rewrite F[(i + 2) % 3]
#define BODY(M, VA, VB, VC)                                                                                                            \
    M("%0", "40:43") VA M("%1", "44:47") VB M("%2", "48:51") VC M("%3", "40:43") VA M("%4", "44:47") VB M("%5", "48:51") VC M("%6", "40:43") VA M("%7", "44:47") VB
#define FRAG_CLOBBERS "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"

template <int MODE>
__global__ __launch_bounds__(64) void k(int iters, float *out, long long *probe) {
    v16f a[8];
    v4f b[8];
    for (int i = 0; i < 8; ++i) {
        for (int e = 0; e < 16; ++e) a[i][e] = 0.f;
        for (int e = 0; e < 4; ++e) b[i][e] = 0.f;
    }
    const int x0 = threadIdx.x * 0x01010101 + blockIdx.x, x1 = 0x11111111;
    asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\t"
                 "v_mov_b32 v46, %0\n\tv_mov_b32 v47, %0\n\tv_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\tv_mov_b32 v50, %0\n\tv_mov_b32 v51, %0\n\t"
                 "v_mov_b32 v52, %0\n\tv_mov_b32 v53, %0\n\tv_mov_b32 v54, %0\n\tv_mov_b32 v55, %0" ::"v"(x1) : FRAG_CLOBBERS);
    const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define RUN32(ASM) asm volatile(ASM : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7]) : "v"(x0), "v"(x1) : FRAG_CLOBBERS)
#define RUN16(ASM) asm volatile(ASM : "+a"(b[0]), "+a"(b[1]), "+a"(b[2]), "+a"(b[3]), "+a"(b[4]), "+a"(b[5]), "+a"(b[6]), "+a"(b[7]) : "v"(x0), "v"(x1) : FRAG_CLOBBERS)
        if (MODE == 0) RUN32(BODY(M32, NONE, NONE, NONE));
        if (MODE == 1) RUN32(BODY(M32, V2("48", "49"), V2("40", "41"), V2("44", "45")));
        if (MODE == 2) RUN32(BODY(M32, V4("48", "49", "50", "51"), V4("40", "41", "42", "43"), V4("44", "45", "46", "47")));
        if (MODE == 3) RUN32(BODY(M32, V4("48", "49", "50", "51") V2("48", "49"), V4("40", "41", "42", "43") V2("40", "41"), V4("44", "45", "46", "47") V2("44", "45")));
        if (MODE == 4) RUN16(BODY(M16, NONE, NONE, NONE));
        if (MODE == 5) RUN16(BODY(M16, V2("48", "49"), V2("40", "41"), V2("44", "45")));
        if (MODE == 6) RUN16(BODY(M16, V4("48", "49", "50", "51"), V4("40", "41", "42", "43"), V4("44", "45", "46", "47")));
#define NOM(ACC, B) ""
        if (MODE == 7) RUN32(BODY(NOM, V4("48", "49", "50", "51"), V4("40", "41", "42", "43"), V4("44", "45", "46", "47")));      // 32 VALU, destinations apart
        if (MODE == 8) RUN32(BODY(NOM, V4("48", "48", "48", "48"), V4("48", "48", "48", "48"), V4("48", "48", "48", "48")));      // 32 VALU, one destination
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0] + b[i][0];
    if (s == 12345.f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 77) { probe[0] = c1 - c0; probe[1] = r1 - r0; }
}

template <int MODE>
void run(const char *what, int per_group, int waves_per_simd, float *out, long long *probe) {
    const int iters = 20000;
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, 100, out, probe);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, iters, out, probe);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, probe, 16, hipMemcpyDeviceToHost);
    printf("%-52s waves/SIMD %d : %6.1f cycles per group per wave at %.2f GHz\n", what, waves_per_simd, (double)h[0] / ((double)iters * 8 / per_group * per_group) ,
           h[0] / (h[1] * 10.0));
}

int main() {
    float *out; long long *probe;
    hipMalloc(&out, 64); hipMalloc(&probe, 64);
    for (int w = 1; w <= 2; ++w) {
        run<0>("32x32x64 fp4, back to back", 1, w, out, probe);
        run<1>("32x32x64 fp4 + 2 VALU each", 1, w, out, probe);
        run<2>("32x32x64 fp4 + 4 VALU each", 1, w, out, probe);
        run<3>("32x32x64 fp4 + 6 VALU each", 1, w, out, probe);
        run<4>("16x16x128 fp4, back to back", 1, w, out, probe);
        run<5>("16x16x128 fp4 + 2 VALU each", 1, w, out, probe);
        run<6>("16x16x128 fp4 + 4 VALU each", 1, w, out, probe);
        run<7>("4 VALU, destinations apart", 1, w, out, probe);
        run<8>("4 VALU, one destination", 1, w, out, probe);
    }
    return 0;
}
