// Micro-benchmark: sustained issue rate of the integer VALU ops the pair kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, int iters, uint32_t s0, uint32_t s1) {
    uint32_t a[16];
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9E3779B9u;
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MODE == 0) {                       // v_and + v_bcnt(acc)   (k_pairC body)
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(y) : "s"(s0 + k), "v"(x));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(y));
            } else if (MODE == 1) {                // v_and only
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[k]) : "s"(s0 + k), "v"(x));
            } else if (MODE == 2) {                // v_bcnt only
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
            } else if (MODE == 3) {                // v_and_or
                asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[k]) : "s"(s0 + k), "v"(x));
            } else {                               // v_add_u32 reference
                asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + y;
}

template <int MODE>
void run(const char *name, int ops_per_k) {
    uint32_t *d;
    const int blocks = 256 * 8, iters = 4000;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<MODE><<<blocks, 256>>>(d, 10, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<MODE><<<blocks, 256>>>(d, iters, 1, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 16 * ops_per_k;
    printf("%-28s %8.3f ms  %.3e wave-instr/s  = %.2f cycles/instr/SIMD at 2.4GHz x 1024 SIMDs\n", name, ms, winstr / (ms * 1e-3),
           2.4e9 * 1024 / (winstr / (ms * 1e-3)));
    hipFree(d);
}

int main() {
    run<0>("v_and(sgpr) + v_bcnt(acc)", 2);
    run<1>("v_and(sgpr)", 1);
    run<2>("v_bcnt_u32_b32 (acc)", 1);
    run<3>("v_and_or_b32 (sgpr)", 1);
    run<4>("v_add_u32", 1);
    return 0;
}
