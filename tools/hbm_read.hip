// HBM read ceiling on this part: (a) plain streaming read with 16-byte loads, (b) the access pattern of k_pack2 (one wave per
// 2048-row group, one dword per lane per row, rows of S bytes).   hipcc -O3 --offload-arch=gfx950 tools/hbm_read.hip -o /tmp/hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stream(const uint4 *__restrict__ p, size_t n16, uint32_t *__restrict__ out) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = p[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// one wave per group of `rows_per_wave` consecutive rows; lane l reads dword l of each row (S/4 lanes active), 32 rows in flight
template <int TPB>
__global__ __launch_bounds__(TPB) void k_rows(const int8_t *__restrict__ gt, int S, int64_t n_rows, int rows_per_wave,
                                              uint32_t *__restrict__ out) {
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int64_t r0 = wave * rows_per_wave;
    if (r0 >= n_rows) return;
    const int h0 = 4 * lane;
    uint32_t acc = 0;
    if (h0 < S) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(gt + r0 * S), 0, rows_per_wave * S, 0x00020000);
        for (int r = 0; r < rows_per_wave; r += 32) {
            uint32_t d[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) d[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, h0, (r + s) * S, 0);
#pragma unroll
            for (int s = 0; s < 32; ++s) acc ^= d[s];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// same rows, 16-byte loads: 16 lanes per row, 4 rows per instruction, NLD instructions in flight per lane, and `valu` dependent
// VALU ops per loaded dword (0 = pure streaming) to see how arithmetic and the row stream share a SIMD
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NLD>
__global__ __launch_bounds__(64) void k_rows4(const int8_t *__restrict__ gt, int S, int64_t n_rows, int rows_per_wave, int valu,
                                              uint32_t *__restrict__ out) {
    const int64_t wave = blockIdx.x;
    const int lane = threadIdx.x & 63, sub = lane & 15, rsel = lane >> 4;
    const int64_t r0 = wave * rows_per_wave;
    if (r0 >= n_rows) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(gt + r0 * S), 0, rows_per_wave * S, 0x00020000);
    const int voff = sub * 16 < S ? rsel * S + sub * 16 : 0x7ffffff0;
    uint32_t acc = 0;
    for (int r = 0; r < rows_per_wave; r += 4 * NLD) {
        u32x4 d[NLD];
#pragma unroll
        for (int s = 0; s < NLD; ++s) d[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (r + 4 * s) * S, 0, 0);
#pragma unroll
        for (int s = 0; s < NLD; ++s) {
            uint32_t x = d[s].x ^ d[s].y ^ d[s].z ^ d[s].w;
            for (int v = 0; v < valu; ++v) x = x * 2654435761u + acc;
            acc ^= x;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int TPB>
__global__ __launch_bounds__(TPB) void k_rows_valu(const int8_t *__restrict__ gt, int S, int64_t n_rows, int rows_per_wave, int valu,
                                                   uint32_t *__restrict__ out) {
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int64_t r0 = wave * rows_per_wave;
    if (r0 >= n_rows) return;
    const int h0 = 4 * lane;
    uint32_t acc = 0;
    if (h0 < S) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(gt + r0 * S), 0, rows_per_wave * S, 0x00020000);
        for (int r = 0; r < rows_per_wave; r += 32) {
            uint32_t d[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) d[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, h0, (r + s) * S, 0);
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                uint32_t x = d[s];
                for (int v = 0; v < valu; ++v) x = x * 2654435761u + acc;
                acc ^= x;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 208;
    const int64_t n_rows = 10000000;
    const size_t bytes = (size_t)n_rows * S;
    int8_t *gt; uint32_t *out;
    CK(hipMalloc(&gt, bytes + 65536)); CK(hipMalloc(&out, 4));
    CK(hipMemset(gt, 1, bytes + 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (const uint4 *)gt, bytes / 16, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("stream  blocks=%6d  %.3f ms  %.0f GB/s\n", blocks, best, bytes / best / 1e6);
    }
    for (int rpw : {2048, 1024, 512, 256}) {
        const int64_t waves = (n_rows + rpw - 1) / rpw;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_rows<64>, dim3((unsigned)waves), dim3(64), 0, 0, gt, S, n_rows, rpw, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("rows    S=%d rows/wave=%5d waves=%6lld  %.3f ms  %.0f GB/s\n", S, rpw, (long long)waves, best, bytes / best / 1e6);
        best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_rows<256>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, gt, S, n_rows, rpw, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("rows256 S=%d rows/wave=%5d waves=%6lld  %.3f ms  %.0f GB/s\n", S, rpw, (long long)waves, best, bytes / best / 1e6);
    }
    {
        const int rpw = 2048;
        const int64_t waves = (n_rows + rpw - 1) / rpw;
        for (int valu : {0, 4, 8, 16, 32}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rows_valu<64>, dim3((unsigned)waves), dim3(64), 0, 0, gt, S, n_rows, rpw, valu, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("rows(dword, 32 in flight) + %d VALU/dword: %.3f ms  %.0f GB/s\n", valu, best, bytes / best / 1e6);
            best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rows4<8>, dim3((unsigned)waves), dim3(64), 0, 0, gt, S, n_rows, rpw, valu, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("rows(16-byte, 8 in flight)  + %d VALU/dword: %.3f ms  %.0f GB/s\n", valu, best, bytes / best / 1e6);
            best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rows4<16>, dim3((unsigned)waves), dim3(64), 0, 0, gt, S, n_rows, rpw, valu, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("rows(16-byte, 16 in flight) + %d VALU/dword: %.3f ms  %.0f GB/s\n", valu, best, bytes / best / 1e6);
        }
    }
    return 0;
}
